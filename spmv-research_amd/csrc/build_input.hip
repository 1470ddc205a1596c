// Input stage of spmv_mi355x_create(): from the arrays csr_to_format() receives (bench.cpp:600-603) to the local CSR a format is
// built from — symmetric-storage expansion (csr_sym.cpp:118-123), row block / column filter of a row-partitioned run
// (SURVEY §8e), and the validation that keeps an out-of-range index away from every kernel.
#include "handle.hpp"

namespace spmv {

// row_ptr must be non-decreasing BEFORE anything is sized from it: a non-monotone row_ptr (row lengths +10 then -10) would
// size a copy from the final prefix sum and overflow it while writing at the per-row offsets
static int
check_row_ptr(const int32_t * row_ptr, long r0, long r1)
{
	long bad = -1;
	#pragma omp parallel for num_threads(spmv::host_threads()) reduction(max : bad)
	for (long i = r0; i < r1; i++)
		if (row_ptr[i + 1] < row_ptr[i])
			bad = std::max(bad, i);
	if (bad >= 0)
	{
		set_error("row_ptr is not monotone at row %ld", bad);
		return 1;
	}
	return 0;
}

int
prepare_local_csr(const spmv_mi355x_opts & o, long m, long n, long nnz, const int32_t * row_ptr, const int32_t * col_idx,
		const double * values, LocalCsr & out)
{
	if (check_row_ptr(row_ptr, 0, m))
		return 1;
	if (row_ptr[0] < 0 || (long) row_ptr[m] - row_ptr[0] != nnz)
	{
		set_error("row_ptr[m]-row_ptr[0] = %ld does not match nnz = %ld", (long) row_ptr[m] - row_ptr[0], nnz);
		return 1;
	}
	// ---- symmetric storage in (KEEP_SYMMETRY builds of the harness; csr_sym.cpp:118-123 accepts exactly this): the arrays
	// hold ONE triangle; the product is y = (T + T^t - diag(T)) x, every stored off-diagonal (i, j, a) also acting as
	// (j, i, +a) — csr_sym.cpp:204-232, bench_spmv.cpp:135-148. The engine expands it and runs its general kernels:
	// scattering a*x[i] into y[j] with fp64 atomics runs at 24 G updates/s on MI355X for scattered j (175 G/s perfectly
	// coalesced; tools/atomic_bench.hip), an order of magnitude short of what halving the matrix stream would need.
	if (o.symmetric_input)
	{
		if (m != n)
		{
			set_error("symmetric_input needs a square matrix (m=%ld n=%ld)", m, n);
			return 1;
		}
		if (row_ptr[0] != 0)
		{
			set_error("symmetric_input: row_ptr must start at 0");
			return 1;
		}
		std::vector<int> cnt((size_t) m + 1, 0);
		long bad = -1;
		#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 4096)
		for (long i = 0; i < m; i++)
			for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++)
			{
				const int c = col_idx[j];
				if (c < 0 || c >= n)
				{
					#pragma omp atomic write
					bad = j;
					continue;
				}
				#pragma omp atomic
				cnt[i + 1]++;
				if (c != i)
				{
					#pragma omp atomic
					cnt[c + 1]++;
				}
			}
		if (bad >= 0)
		{
			set_error("column index %d out of range [0,%ld) at entry %ld", col_idx[bad], n, bad);
			return 1;
		}
		long total = 0;
		for (long i = 0; i < m; i++)
			total += cnt[i + 1];
		if (total >= 0x7fffffffL)
		{
			set_error("symmetric_input: the expanded matrix has %ld entries, beyond the int32 index range", total);
			return 1;
		}
		out.e_rp.assign((size_t) m + 1, 0);
		for (long i = 0; i < m; i++)
			out.e_rp[i + 1] = out.e_rp[i] + cnt[i + 1];
		out.e_ci.resize((size_t) std::max<long>(total, 1));
		out.e_va.resize((size_t) std::max<long>(total, 1));
		std::vector<int> pos(out.e_rp.begin(), out.e_rp.end() - 1);
		#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 4096)
		for (long i = 0; i < m; i++)
			for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++)
			{
				const int c = col_idx[j];
				int k;
				#pragma omp atomic capture
				k = pos[i]++;
				out.e_ci[k] = c;
				out.e_va[k] = values[j];
				if (c != i)
				{
					#pragma omp atomic capture
					k = pos[c]++;
					out.e_ci[k] = (int) i;
					out.e_va[k] = values[j];
				}
			}
		// rows ascending, columns ascending (what coo_to_csr gives the general path, csr_gen.c:178-213); equal columns are
		// ordered by value so the result does not depend on the thread interleaving above
		#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 1024)
		for (long i = 0; i < m; i++)
		{
			const long s0 = out.e_rp[i], len = out.e_rp[i + 1] - s0;
			bool sorted = true;
			for (long k = 1; k < len && sorted; k++)
				sorted = out.e_ci[s0 + k - 1] < out.e_ci[s0 + k];
			if (sorted)
				continue;
			std::vector<std::pair<int, double>> tmp((size_t) len);
			for (long k = 0; k < len; k++)
				tmp[k] = {out.e_ci[s0 + k], out.e_va[s0 + k]};
			std::sort(tmp.begin(), tmp.end());
			for (long k = 0; k < len; k++)
			{
				out.e_ci[s0 + k] = tmp[k].first;
				out.e_va[s0 + k] = tmp[k].second;
			}
		}
		row_ptr = out.e_rp.data();
		col_idx = out.e_ci.data();
		values = out.e_va.data();
		nnz = total;
	}

	// ---- row block / column filter (row-partitioned multi-GPU) -> local CSR on the host
	long r0 = o.row_begin, r1 = o.row_end;
	if (r0 == 0 && r1 == 0)
		r1 = m;
	if (r0 < 0 || r1 > m || r0 > r1)
	{
		set_error("bad row block [%ld,%ld) for m=%ld", r0, r1, m);
		return 1;
	}
	const long lm = r1 - r0;
	long lnnz;
	const bool filter = o.col_filter_mode == 1 || o.col_filter_mode == 2;
	if (!filter && row_ptr[r0] == 0)
	{
		out.rp = row_ptr + r0;     // [0, r1) prefix: offsets are already local
		out.ci = col_idx;
		out.va = values;
		lnnz = row_ptr[r1];
	}
	else
	{
		out.l_rp.assign((size_t) lm + 1, 0);
		const long c0 = o.col_begin, c1 = o.col_end;
		const bool inside = o.col_filter_mode == 1;
		#pragma omp parallel for num_threads(spmv::host_threads())
		for (long i = 0; i < lm; i++)
		{
			int cnt = 0;
			if (!filter)
				cnt = row_ptr[r0 + i + 1] - row_ptr[r0 + i];
			else
				for (long j = row_ptr[r0 + i]; j < row_ptr[r0 + i + 1]; j++)
				{
					bool in = col_idx[j] >= c0 && col_idx[j] < c1;
					cnt += (in == inside);
				}
			out.l_rp[i + 1] = cnt;
		}
		for (long i = 0; i < lm; i++)
			out.l_rp[i + 1] += out.l_rp[i];
		lnnz = out.l_rp[lm];
		out.l_ci.resize((size_t) std::max<long>(lnnz, 1));
		out.l_va.resize((size_t) std::max<long>(lnnz, 1));
		#pragma omp parallel for num_threads(spmv::host_threads())
		for (long i = 0; i < lm; i++)
		{
			long k = out.l_rp[i];
			for (long j = row_ptr[r0 + i]; j < row_ptr[r0 + i + 1]; j++)
			{
				if (filter)
				{
					bool in = col_idx[j] >= c0 && col_idx[j] < c1;
					if (in != inside)
						continue;
				}
				out.l_ci[k] = col_idx[j];
				out.l_va[k] = values[j];
				k++;
			}
		}
		out.rp = out.l_rp.data();
		out.ci = out.l_ci.data();
		out.va = out.l_va.data();
	}
	out.m = lm;
	out.nnz = lnnz;
	const int * rp = out.rp;
	const int * ci = out.ci;
	// full validation before anything reaches a kernel: an out-of-range index would be an out-of-bounds device read
	{
		long bad_col = -1, bad_row = -1;
		#pragma omp parallel for num_threads(spmv::host_threads()) reduction(max : bad_col)
		for (long j = 0; j < lnnz; j++)
			if (ci[j] < 0 || ci[j] >= n)
				bad_col = std::max(bad_col, j);
		#pragma omp parallel for num_threads(spmv::host_threads()) reduction(max : bad_row)
		for (long i = 0; i < lm; i++)
			if (rp[i + 1] < rp[i])
				bad_row = std::max(bad_row, i);
		if (bad_col >= 0 || bad_row >= 0)
		{
			if (bad_col >= 0)
				set_error("column index %d out of range [0,%ld) at entry %ld", ci[bad_col], n, bad_col);
			else
				set_error("row_ptr is not monotone at row %ld", bad_row);
			return 1;
		}
	}
	return 0;
}

}  // namespace spmv
