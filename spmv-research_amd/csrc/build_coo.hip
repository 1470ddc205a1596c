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
	const int WGS = coo_blocked_wgs_per_range(), RND = coo_blocked_chunk_rows(), KMAX = coo_blocked_max_long_rows();
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
		row = std::min<long>(lm, (row + RND / 2) / RND * RND);
		range_row[r] = (int) std::max<long>(row, range_row[r - 1]);
	}
	range_row[NR] = (int) lm;
	for (long r = 0; r < NR; r++)
		fits = fits && range_row[r + 1] - range_row[r] <= range_cap;
	if (!fits)
	{
		const long per = ((lm + NR - 1) / NR + RND - 1) / RND * RND;
		for (long r = 0; r <= NR; r++)
			range_row[r] = (int) std::min<long>(lm, r * per);
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
	// ---- how a range's rows are dealt to its 32 workgroups: chunks of CH rows, round-robin. CH = 16 (default): every workgroup sees
	// the whole range's column structure, entries per block are even. SPMV_MI355X_COOB_CHUNK=0: ONE contiguous chunk per workgroup
	// (fewer distinct x lines per workgroup on banded graphs, uneven block fill at the range's ends).
	long CH = RND;
	if (const char * e = getenv("SPMV_MI355X_COOB_CHUNK"))
	{
		if (atol(e) == 0)
		{
			long widest = 0;
			for (long r = 0; r < NR; r++)
				widest = std::max<long>(widest, range_row[r + 1] - range_row[r]);
			CH = std::max<long>(RND, ((widest + WGS - 1) / WGS + RND - 1) / RND * RND);
		}
		else if (atol(e) >= RND && atol(e) % RND == 0 && atol(e) <= cap)
			CH = atol(e);
	}
	// ---- column blocks, per range (the 32 workgroups of a range walk the same blocks in step): every block spans at most 65 536
	// columns (16-bit offsets). col_blocks > 0: uniform blocks of ceil(n / col_blocks) columns. col_blocks = -1: EQUI-DEPTH blocks —
	// as many columns as hold one full batch of entries per workgroup (85 % of it on average: the deal is statistical), so that
	// every step of the kernel is a full batch wherever the matrix is dense enough, and empty stretches of columns cost nothing.
	std::vector<std::vector<int>> bstart((size_t) NR);
	{
		const long W = col_blocks > 0 ? std::max<long>(1, std::min<long>((n + col_blocks - 1) / col_blocks, 65536)) : 0;
		const long target = (long) (0.85 * WGS * coo_blocked_batch_entries());
		#pragma omp parallel num_threads(std::min<int>(spmv::host_threads(), (int) NR))
		{
			std::vector<int> hist;
			#pragma omp for schedule(dynamic, 1)
			for (long r = 0; r < NR; r++)
			{
				const long e0 = rp[range_row[r]], e1 = rp[range_row[r + 1]];
				std::vector<int> & bs = bstart[(size_t) r];
				if (e1 == e0)
					continue;
				if (W > 0)
				{
					int lo = 0x7fffffff, hi = -1;
					for (long e = e0; e < e1; e++)
					{
						lo = std::min(lo, ci[e]);
						hi = std::max(hi, ci[e]);
					}
					for (long b = lo / W; b <= hi / W; b++)
						bs.push_back((int) (b * W));
					continue;
				}
				hist.assign((size_t) n, 0);
				for (long e = e0; e < e1; e++)
					hist[(size_t) ci[e]]++;
				long start = -1, acc = 0;
				for (long c = 0; c < n; c++)
				{
					const long h = hist[(size_t) c];
					if (h == 0)
						continue;
					if (start < 0 || c - start >= 65536 || (acc > 0 && acc + h > target))
					{
						bs.push_back((int) c);
						start = c;
						acc = 0;
					}
					acc += h;
				}
			}
		}
	}
	long B = 1;
	for (long r = 0; r < NR; r++)
		B = std::max<long>(B, (long) bstart[(size_t) r].size());
	if (B > 65536)
	{
		set_error("col_blocks: %ld column blocks in one row range (limit 65536)", B);
		return 1;
	}
	// block of column c in range r: the last block starting at or before c
	auto block_of = [&](long r, int c) {
		const std::vector<int> & bs = bstart[(size_t) r];
		return (long) (std::upper_bound(bs.begin(), bs.end(), c) - bs.begin()) - 1;
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
					pos[(size_t) block_of(r, ci[e]) + 1]++;
			});
			sb[0] = (int) wg_nnz[t];
			for (long b = 0; b < B; b++)
				sb[b + 1] = sb[b] + pos[(size_t) b + 1];
			for (long b = 0; b <= B; b++)
				pos[(size_t) b] = sb[b];
			for_each_span([&](long a, long b, unsigned l) {
				for (long e = a; e < b; e++)
				{
					const long blk = block_of(r, ci[e]);
					const int at = pos[(size_t) blk]++;
					ent[(size_t) at] = ((unsigned) (ci[e] - bstart[(size_t) r][(size_t) blk]) << 16) | l;
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
	// ---- per range: number of blocks, then the first column of every block
	std::vector<int> range_blk((size_t) NR * (B + 1), 0);
	int max_rows = 0;
	for (long r = 0; r < NR; r++)
	{
		const std::vector<int> & bs = bstart[(size_t) r];
		range_blk[(size_t) r * (B + 1)] = (int) bs.size();
		for (size_t b = 0; b < bs.size(); b++)
			range_blk[(size_t) r * (B + 1) + 1 + b] = bs[b];
		for (long t = r * WGS; t < (r + 1) * WGS; t++)
			max_rows = std::max(max_rows, wg_rows[(size_t) t]);
	}
	if (upload_ints(wg_rows.data(), wg_rows.size(), &A->d_coob_wg_rows) || upload_ints(range_row.data(), range_row.size(), &A->d_coob_range_row) ||
	    upload_ints(seg_blk.data(), seg_blk.size(), &A->d_coob_seg_blk) || upload_ints(range_blk.data(), range_blk.size(), &A->d_coob_range_blk) ||
	    upload_ints(range_long.data(), range_long.size(), &A->d_coob_range_long) || upload_ints(long_row.data(), long_row.size(), &A->d_coob_long_row) ||
	    dev_alloc_bytes(&A->d_coob_carry, (size_t) std::max<long>(NL, 1) * WGS * A->vbytes) ||
	    upload_bytes(ent.data(), (size_t) lnnz * 4, (size_t) coo_blocked_entry_slack() * 4, (void **) &A->d_coob_ent))
		return 1;
	if (!uniform)
	{
		// narrowed values with the same slack as the entries (the kernel's loads are unconditional)
		const size_t slack = (size_t) coo_blocked_entry_slack() * A->vbytes;
		if (A->f32)
		{
			std::vector<float> pf32((size_t) std::max<long>(lnnz, 1));
			#pragma omp parallel for num_threads(spmv::host_threads())
			for (long e = 0; e < lnnz; e++)
				pf32[(size_t) e] = (float) pval[(size_t) e];
			if (upload_bytes(pf32.data(), (size_t) lnnz * 4, slack, &A->d_val))
				return 1;
		}
		else if (upload_bytes(pval.data(), (size_t) lnnz * 8, slack, &A->d_val))
			return 1;
	}
	if (uniform)
	{
		A->cfg.unit = 1;
		A->cfg.unit_value = v0;
	}
	A->coob_ranges = (int) NR;
	A->coob_blocks = (int) B;
	A->coob_block_cols = (int) CH;                // rows per chunk of the deal (the field's second life)
	A->coob_num_long = (int) NL;
	A->coob_lds = (int) (((long) std::max(max_rows, 1) * 8 + 15) / 16 * 16);      // fp64 slots for both precisions
	A->cfg.map = xcd_map_uniform(1, 0);
	A->mem_footprint = (double) lnnz * (4 + (uniform ? 0 : A->vbytes)) + (double) NT * (B + 2) * 4 + (double) NR * (B + 4) * 4 + NL * (4.0 + WGS * A->vbytes);
	snprintf(A->format_name, sizeof(A->format_name), "MI355X_%s_r%ld_%s%ld%s%s_%s", merge_balance ? "MERGEB" : "COOB", NR, col_blocks > 0 ? "b" : "e", B, NL ? "_split" : "",
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
