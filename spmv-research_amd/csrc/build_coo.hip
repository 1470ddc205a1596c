// csr_to_format() for COO: the reference's row-sorted expansion (mkl_coo.cpp:79-90: rowind[j] = i for j in row i) done on the
// GPU, and the column-blocked layout for graph matrices (kernels_coo.hip).
#include "handle.hpp"

namespace spmv {

// Column-blocked layout (opts.col_blocks; described in kernels_coo.hip). Built on the host: row ranges balanced by non-zeros
// (COO: equal shares of the entries, as coo_kernel's waves) or by rows + non-zeros (merge path: equal shares of the merge
// diagonal, merge.cpp:277-287), cut at chunk boundaries — by rows alone when a share would overflow a workgroup's LDS;
// chunks of 16 rows dealt round-robin to the 32 workgroups of a range; rows longer than a workgroup's fair share / 8 are
// split over the 32 workgroups; per workgroup a counting sort by column block and inside a block a sort by (column, row).
int
build_blocked_layout(spmv_mi355x_matrix * A, const int * rp, const int * ci, const double * va, int col_blocks, bool merge_balance)
{
	const long lm = A->m, lnnz = A->nnz, n = A->n;
	const char * pf = A->f32 ? "f" : "d";
	const int WGS = coo_blocked_wgs_per_range(), CH = coo_blocked_chunk_rows(), KMAX = coo_blocked_max_long_rows();
	const long cap = coo_blocked_rows_cap(A->f32);                 // chunk rows per workgroup (KMAX more slots are set aside for split rows)
	const long range_cap = cap * WGS;                              // rows per range
	const long P = std::max<long>(1, (lm + NUM_XCD * range_cap - 1) / (NUM_XCD * range_cap));
	const long NR = NUM_XCD * P, NT = NR * WGS;
	// ---- row ranges
	std::vector<int> range_row((size_t) NR + 1, 0);
	bool fits = true;
	for (long r = 1; r < NR; r++)
	{
		long row;
		if (merge_balance)
		{
			// first row whose merge-path coordinate (rows before it + entries before it) reaches the r-th share of the diagonal
			const long tgt = (long) ((double) (lm + lnnz) * r / NR);
			long lo = 0, hi = lm;
			while (lo < hi)
			{
				const long mid = (lo + hi) / 2;
				if ((long) rp[mid] + mid >= tgt)
					hi = mid;
				else
					lo = mid + 1;
			}
			row = lo;
		}
		else
		{
			const long tgt = (long) ((double) lnnz * r / NR);
			row = std::lower_bound(rp, rp + lm + 1, (int) std::min<long>(tgt, 0x7fffffffL)) - rp;
		}
		row = std::min<long>(lm, (row + CH / 2) / CH * CH);
		range_row[r] = (int) std::max<long>(row, range_row[r - 1]);
	}
	range_row[NR] = (int) lm;
	for (long r = 0; r < NR; r++)
		fits = fits && range_row[r + 1] - range_row[r] <= range_cap;
	if (!fits)
	{
		const long per = ((lm + NR - 1) / NR + CH - 1) / CH * CH;
		for (long r = 0; r <= NR; r++)
			range_row[r] = (int) std::min<long>(lm, r * per);
	}
	// ---- column blocks
	long W;
	if (col_blocks > 0)
		W = std::max<long>(1, (n + col_blocks - 1) / col_blocks);
	else
		W = (384L << 10) / A->vbytes;              // ~384 KiB of x per block
	W = std::max<long>(1, std::min<long>(W, 65536));   // 16-bit column offsets
	const long B = std::max<long>(1, (n + W - 1) / W);
	if (B > 16384)
	{
		set_error("col_blocks: %ld column blocks of %ld columns (limit 16384)", B, W);
		return 1;
	}
	double v0 = 0;
	const bool uniform = values_uniform(A, va, lnnz, &v0);
	// ---- rows to split: longer than 1/8 of a workgroup's fair share of the entries (at most KMAX per range, in row order)
	long long_min = std::max<long>(256, lnnz / NT / 8);
	if (const char * e = getenv("SPMV_MI355X_COOB_LONG_MIN"))      // tests: split ordinary rows of small matrices too
		if (atol(e) >= 1)
			long_min = atol(e);
	std::vector<int> range_long((size_t) NR + 1, 0), long_row;
	for (long r = 0; r < NR; r++)
	{
		int k = 0;
		for (long row = range_row[r]; row < range_row[r + 1] && k < KMAX; row++)
			if (rp[row + 1] - rp[row] >= long_min)
			{
				long_row.push_back((int) row);
				k++;
			}
		range_long[r + 1] = (int) long_row.size();
	}
	const long NL = (long) long_row.size();
	auto is_long = [&](long r, long row) {
		return std::binary_search(long_row.begin() + range_long[r], long_row.begin() + range_long[r + 1], (int) row);
	};
	// piece of a split row that workgroup j sums: entries [len*j/32, len*(j+1)/32)
	auto piece = [&](long row, long j, long & a, long & b) {
		const long len = rp[row + 1] - rp[row];
		a = rp[row] + len * j / WGS;
		b = rp[row] + len * (j + 1) / WGS;
	};
	// ---- entries of every workgroup: offsets first (workgroups in order), then fill + sort
	std::vector<int> wg_rows((size_t) NT, 0);
	std::vector<long> wg_nnz((size_t) NT + 1, 0);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 8)
	for (long t = 0; t < NT; t++)
	{
		const long r = t / WGS, j = t % WGS;
		const long r0 = range_row[r], r1 = range_row[r + 1];
		const long chunks = (r1 - r0 + CH - 1) / CH;
		const long mine = chunks > j ? (chunks - j + WGS - 1) / WGS : 0;
		const long nlong = range_long[r + 1] - range_long[r];
		wg_rows[t] = (int) (mine * CH + nlong);
		long cnt = 0;
		for (long c = j; c < chunks; c += WGS)
		{
			const long a = r0 + c * CH, b = std::min(r1, a + CH);
			cnt += rp[b] - rp[a];
			if (nlong)
				for (long row = a; row < b; row++)
					if (is_long(r, row))
						cnt -= rp[row + 1] - rp[row];
		}
		for (long k = range_long[r]; k < range_long[r + 1]; k++)
		{
			long a, b;
			piece(long_row[(size_t) k], j, a, b);
			cnt += b - a;
		}
		wg_nnz[t + 1] = cnt;
	}
	for (long t = 0; t < NT; t++)
		wg_nnz[t + 1] += wg_nnz[t];
	std::vector<int> seg_blk((size_t) NT * (B + 1), 0);
	std::vector<unsigned> ent((size_t) std::max<long>(lnnz, 1));
	std::vector<double> pval(uniform ? 0 : (size_t) std::max<long>(lnnz, 1));
	#pragma omp parallel num_threads(spmv::host_threads())
	{
		std::vector<int> pos((size_t) B + 1);
		std::vector<std::pair<unsigned, double>> tmp;
		#pragma omp for schedule(dynamic, 4)
		for (long t = 0; t < NT; t++)
		{
			const long r = t / WGS, j = t % WGS;
			const long r0 = range_row[r], r1 = range_row[r + 1];
			const long chunks = (r1 - r0 + CH - 1) / CH;
			const long mine = chunks > j ? (chunks - j + WGS - 1) / WGS : 0;
			const long nlong = range_long[r + 1] - range_long[r];
			int * sb = seg_blk.data() + (size_t) t * (B + 1);
			// the workgroup's entries as (first, last, LDS slot) spans: its chunk rows, then its pieces of the split rows
			auto for_each_span = [&](auto && fn) {
				for (long c = j; c < chunks; c += WGS)
				{
					const long a = r0 + c * CH, bnd = std::min(r1, a + CH);
					for (long row = a; row < bnd; row++)
						if (!(nlong && is_long(r, row)))
							fn((long) rp[row], (long) rp[row + 1], (unsigned) ((c / WGS) * CH + (row - a)));
				}
				for (long k = 0; k < nlong; k++)
				{
					long a, b;
					piece(long_row[(size_t) (range_long[r] + k)], j, a, b);
					fn(a, b, (unsigned) (mine * CH + k));
				}
			};
			std::fill(pos.begin(), pos.end(), 0);
			for_each_span([&](long a, long b, unsigned) {
				for (long e = a; e < b; e++)
					pos[(size_t) (ci[e] / W) + 1]++;
			});
			sb[0] = (int) wg_nnz[t];
			for (long b = 0; b < B; b++)
				sb[b + 1] = sb[b] + pos[(size_t) b + 1];
			for (long b = 0; b <= B; b++)
				pos[(size_t) b] = sb[b];
			for_each_span([&](long a, long b, unsigned l) {
				for (long e = a; e < b; e++)
				{
					const long blk = ci[e] / W;
					const int at = pos[(size_t) blk]++;
					ent[(size_t) at] = ((unsigned) (ci[e] - blk * W) << 16) | l;
					if (!uniform)
						pval[(size_t) at] = va[e];
				}
			});
			// inside a block: by column, then LDS slot (the packed dword is that key)
			for (long b = 0; b < B; b++)
			{
				const int s0 = sb[b], s1 = sb[b + 1];
				if (s1 - s0 < 2)
					continue;
				if (uniform)
					std::sort(ent.begin() + s0, ent.begin() + s1);
				else
				{
					tmp.resize((size_t) (s1 - s0));
					for (int k = s0; k < s1; k++)
						tmp[(size_t) (k - s0)] = std::make_pair(ent[(size_t) k], pval[(size_t) k]);
					std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<unsigned, double> & p, const std::pair<unsigned, double> & q) { return p.first < q.first; });
					for (int k = s0; k < s1; k++)
					{
						ent[(size_t) k] = tmp[(size_t) (k - s0)].first;
						pval[(size_t) k] = tmp[(size_t) (k - s0)].second;
					}
				}
			}
		}
	}
	if (wg_nnz[NT] != lnnz)
	{
		set_error("column-blocked layout: %ld entries placed, %ld expected", wg_nnz[NT], lnnz);
		return 1;
	}
	// ---- column blocks a range's workgroups walk together: first to last block any of them has entries in
	std::vector<int> range_blk((size_t) 2 * NR, 0);
	int max_rows = 0;
	for (long r = 0; r < NR; r++)
	{
		long lo = B, hi = 0;
		for (long t = r * WGS; t < (r + 1) * WGS; t++)
		{
			const int * sb = seg_blk.data() + (size_t) t * (B + 1);
			for (long b = 0; b < B; b++)
				if (sb[b + 1] > sb[b])
				{
					lo = std::min(lo, b);
					hi = std::max(hi, b + 1);
				}
			max_rows = std::max(max_rows, wg_rows[(size_t) t]);
		}
		range_blk[(size_t) 2 * r] = (int) (hi > lo ? lo : 0);
		range_blk[(size_t) 2 * r + 1] = (int) (hi > lo ? hi : 0);
	}
	if (upload_ints(wg_rows.data(), wg_rows.size(), &A->d_coob_wg_rows) || upload_ints(range_row.data(), range_row.size(), &A->d_coob_range_row) ||
	    upload_ints(seg_blk.data(), seg_blk.size(), &A->d_coob_seg_blk) || upload_ints(range_blk.data(), range_blk.size(), &A->d_coob_range_blk) ||
	    upload_ints(range_long.data(), range_long.size(), &A->d_coob_range_long) || upload_ints(long_row.data(), long_row.size(), &A->d_coob_long_row) ||
	    dev_alloc_bytes(&A->d_coob_carry, (size_t) std::max<long>(NL, 1) * WGS * A->vbytes) ||
	    upload_bytes(ent.data(), (size_t) lnnz * 4, STREAM_SLACK * 4, (void **) &A->d_coob_ent))
		return 1;
	if (!uniform && upload_values(A, pval.data(), (size_t) lnnz, &A->d_val))
		return 1;
	if (uniform)
	{
		A->cfg.unit = 1;
		A->cfg.unit_value = v0;
	}
	A->coob_ranges = (int) NR;
	A->coob_blocks = (int) B;
	A->coob_block_cols = (int) W;
	A->coob_num_long = (int) NL;
	A->coob_lds = (int) (((long) std::max(max_rows, 1) * A->vbytes + 15) / 16 * 16);
	A->cfg.map = xcd_map_uniform(1, 0);
	A->mem_footprint = (double) lnnz * (4 + (uniform ? 0 : A->vbytes)) + (double) NT * (B + 2) * 4 + (4.0 * NR + 2) * 4 + NL * (4.0 + WGS * A->vbytes);
	snprintf(A->format_name, sizeof(A->format_name), "MI355X_%s_r%ld_b%ld%s%s_%s", merge_balance ? "MERGEB" : "COOB", NR, B, NL ? "_split" : "",
			uniform ? "_unit" : "", pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), "coo_blocked_kernel");
	A->kernel_block = 1024;
	return 0;
}

int
build_coo_family(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va)
{
	const long lm = A->m, lnnz = A->nnz;
	const char * pf = A->f32 ? "f" : "d";
	if (o.col_blocks != 0)
		return build_blocked_layout(A, rp, ci, va, o.col_blocks, false);
	if (upload_ints(rp, (size_t) lm + 1, &A->d_row_ptr) || upload_ints(ci, (size_t) lnnz, &A->d_col) ||
	    upload_values(A, va, (size_t) lnnz, &A->d_val) || dev_alloc(&A->d_rowind, (size_t) lnnz))
		return 1;
	if (launch_expand_rows(A->d_row_ptr, (int) lm, A->d_rowind, nullptr))
		return 1;
	if (hipDeviceSynchronize() != hipSuccess)
	{
		set_error("COO row expansion failed");
		return 1;
	}
	(void) hipFree(A->d_row_ptr);          // COO keeps (rowind, colind, val) only: mkl_coo.cpp:65
	A->d_row_ptr = nullptr;
	const int per_wave = coo_wave_items(o.merge_items);
	A->coo_k = per_wave / WAVE;
	A->coo_num_waves = (int) ((lnnz + per_wave - 1) / per_wave);
	if (dev_alloc(&A->d_carry_row, (size_t) A->coo_num_waves) || dev_alloc_bytes(&A->d_carry_val, (size_t) A->coo_num_waves * A->vbytes))
		return 1;
	A->cfg.map = xcd_map_uniform((unsigned) ((A->coo_num_waves + coo_waves_per_tile() - 1) / coo_waves_per_tile()), resolve_remap(A->remap, 0));
	A->mem_footprint = (double) lnnz * (A->vbytes + 8);
	snprintf(A->format_name, sizeof(A->format_name), "MI355X_COO_k%d_%s", A->coo_k, pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), "coo_kernel");
	return 0;
}

}  // namespace spmv
