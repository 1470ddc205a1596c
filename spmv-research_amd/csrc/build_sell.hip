// csr_to_format() for SELL-C-sigma (sell_sorted.cpp:112-298, sell_c_s.cpp:39-77 + sell-C-s/RISC-V/sellcs_format.c:137-200):
// the plain column-major layout (C = 16/32/64), the delta-compressed SELL-64 layout (host builder; the GPU builder is
// convert_sell.hip) and the LDS-window layout for banded matrices (kernels_sell_window.hip).
#include "handle.hpp"

namespace spmv {

// ---------------------------------------------------------------------------------------------------- SELL build
// Host-side CSR -> SELL-C-sigma (the reference converts on the host too: sell_sorted.cpp:112-298, sellcs_format.c:137-200).
// Window sort: stable, DESCENDING row length inside each window of sigma rows (radix_sort.c:103-122 semantics).
static int
build_sell(spmv_mi355x_matrix * A, const int * rp, const int * ci, const double * va)
{
	const long m = A->m;
	const int C = A->sell_c;
	const int TPR = C >= WAVE ? 1 : WAVE / C;            // lanes per row (C = 256: one workgroup per slice, one lane per row)
	const long slices_per_tile = C > WAVE ? 1 : sell_slices_per_tile();
	const long sigma = A->sell_sigma;
	const long num_slices = (m + C - 1) / C;
	if (A->convert_on_device)
	{
		// the GPU builder (convert_sell.hip): the same bytes
		std::vector<int64_t> slice_ptr;
		if (sell_plain_convert_device(A->f32, m, A->nnz, C, TPR, sigma, rp, ci, va, &A->d_slice_ptr, &A->d_col, &A->d_val, &A->d_row_of_sorted, slice_ptr))
			return 1;
		const int64_t nnz_ext = slice_ptr[(size_t) num_slices];
		A->sell_slices = num_slices;
		A->sell_nnz_ext = nnz_ext;
		A->cfg.map = xcd_map_balanced(slice_ptr.data(), num_slices, slices_per_tile, resolve_remap(A->remap, (num_slices + slices_per_tile - 1) / slices_per_tile));
		A->mem_footprint = (double) (num_slices + 1) * sizeof(int64_t) + (double) nnz_ext * (A->vbytes + 4) + (double) m * 4;
		return 0;
	}
	std::vector<int> row_of_sorted(std::max<long>(m, 1));
	const long num_windows = (m + sigma - 1) / sigma;
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4)
	for (long w = 0; w < num_windows; w++)
	{
		long s = w * sigma, e = std::min(m, s + sigma);
		// counting sort by length, descending, stable
		int maxlen = 0;
		for (long i = s; i < e; i++)
			maxlen = std::max(maxlen, rp[i + 1] - rp[i]);
		std::vector<long> cnt((size_t) maxlen + 2, 0);
		for (long i = s; i < e; i++)
			cnt[maxlen - (rp[i + 1] - rp[i]) + 1]++;
		for (int b = 0; b <= maxlen; b++)
			cnt[b + 1] += cnt[b];
		for (long i = s; i < e; i++)
			row_of_sorted[s + cnt[maxlen - (rp[i + 1] - rp[i])]++] = (int) i;
	}
	std::vector<int64_t> slice_ptr((size_t) num_slices + 1, 0);
	#pragma omp parallel for num_threads(spmv::host_threads())
	for (long sl = 0; sl < num_slices; sl++)
	{
		long width = 0;
		for (long i = sl * C; i < std::min(m, (sl + 1) * C); i++)
		{
			int o = row_of_sorted[i];
			width = std::max<long>(width, rp[o + 1] - rp[o]);
		}
		width = (width + TPR - 1) / TPR * TPR;
		slice_ptr[sl + 1] = width * C;
	}
	for (long sl = 0; sl < num_slices; sl++)
		slice_ptr[sl + 1] += slice_ptr[sl];
	const int64_t nnz_ext = slice_ptr[num_slices];
	std::vector<int> col((size_t) std::max<int64_t>(nnz_ext, 1));
	std::vector<double> val((size_t) std::max<int64_t>(nnz_ext, 1));
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 64)
	for (long sl = 0; sl < num_slices; sl++)
	{
		const int64_t base = slice_ptr[sl];
		const long width = (slice_ptr[sl + 1] - base) / C;
		for (int r = 0; r < C; r++)
		{
			long i = sl * C + r;
			long js = 0, len = 0;
			if (i < m)
			{
				int o = row_of_sorted[i];
				js = rp[o];
				len = rp[o + 1] - rp[o];
			}
			// padding: value 0 times a column this row already touches (keeps the gather in cache; the reference pads
			// with the last real column as well, sell_sorted.cpp:280-284)
			int pad_col = len > 0 ? ci[js + len - 1] : 0;
			for (long k = 0; k < width; k++)
			{
				int64_t p = base + k * C + r;
				if (k < len)
				{
					col[p] = ci[js + k];
					val[p] = va[js + k];
				}
				else
				{
					col[p] = pad_col;
					val[p] = 0.0;
				}
			}
		}
	}
	A->sell_slices = num_slices;
	A->sell_nnz_ext = nnz_ext;
	A->cfg.map = xcd_map_balanced(slice_ptr.data(), num_slices, slices_per_tile,
			resolve_remap(A->remap, (num_slices + slices_per_tile - 1) / slices_per_tile));
	if (dev_alloc(&A->d_slice_ptr, (size_t) num_slices + 1))
		return 1;
	HIP_TRY(hipMemcpy(A->d_slice_ptr, slice_ptr.data(), ((size_t) num_slices + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
	if (upload_ints(col.data(), (size_t) nnz_ext, &A->d_col))
		return 1;
	if (upload_values(A, val.data(), (size_t) nnz_ext, &A->d_val))
		return 1;
	if (upload_ints(row_of_sorted.data(), (size_t) m, &A->d_row_of_sorted))
		return 1;
	// (num_slices+1) offsets + padded entries + the row permutation (cf. sell_sorted.cpp:297)
	A->mem_footprint = (double) (num_slices + 1) * sizeof(int64_t) + (double) nnz_ext * (A->vbytes + 4) + (double) m * 4;
	return 0;
}

// mode 5 of the delta layout: the EXCEPTION lanes of a slice (64 equally long rows rows[0..63]) — those whose columns are not
// column_k[reference r] + (first column - reference's first column) at every step — and whether some lane's difference does not fit a
// signed byte (`hard`): the host twin of sell5_exceptions() in convert_sell.hip
static unsigned long long
sell5_exceptions(const int * rp, const int * ci, const int * rows, long maxlen, int r, bool & hard)
{
	unsigned long long mask = 0;
	hard = false;
	const int * cr = ci + rp[rows[r]];
	for (int l = 0; l < 64; l++)
	{
		const int * cl = ci + rp[rows[l]];
		const int off = cl[0] - cr[0];
		for (long k = 1; k < maxlen; k++)
		{
			const int d = cl[k] - cr[k] - off;
			if (d != 0)
				mask |= 1ull << l;
			if (d < -128 || d > 127)
				hard = true;
		}
	}
	return mask;
}

// SELL-64-sigma-delta build (layout: kernels_sell.hip). Same sigma-window sort and slice widths as build_sell with C = 64,
// widths padded to a multiple of 4 steps; per slice the narrowest index encoding that holds every (step, lane) delta.
static int
build_sell_delta(spmv_mi355x_matrix * A, const int * rp, const int * ci, const double * va)
{
	const long m = A->m;
	constexpr int C = 64;
	const long sigma = A->sell_sigma;
	const long num_slices = (m + C - 1) / C;
	if (A->convert_on_device)
	{
		std::vector<int64_t> val_ptr;
		int64_t nnz_ext = 0, idx_bytes = 0;
		void * d_val = nullptr;
		if (sell_delta_convert_device(A->f32, m, A->n, A->nnz, sigma, rp, ci, va, &A->d_row_of_sorted, &A->d_sell_desc, &A->d_sell_idx,
				&d_val, val_ptr, A->sell_mode_slices, &nnz_ext, &idx_bytes))
			return 1;
		A->d_val = d_val;
		A->sell_slices = num_slices;
		A->sell_nnz_ext = nnz_ext;
		A->sell_idx_bytes = idx_bytes;
		const long spt = sell_slices_per_tile() / A->sell_split;
		A->cfg.map = xcd_map_balanced(val_ptr.data(), num_slices, spt, resolve_remap(A->remap, (num_slices + spt - 1) / spt));
		A->mem_footprint = (double) (num_slices + 1) * 16 + (double) nnz_ext * A->vbytes + (double) idx_bytes + (double) m * 4;
		return 0;
	}
	std::vector<int> row_of_sorted(std::max<long>(m, 1));
	const long num_windows = (m + sigma - 1) / sigma;
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4)
	for (long w = 0; w < num_windows; w++)
	{
		long s = w * sigma, e = std::min(m, s + sigma);
		int maxlen = 0;
		for (long i = s; i < e; i++)
			maxlen = std::max(maxlen, rp[i + 1] - rp[i]);
		std::vector<long> cnt((size_t) maxlen + 2, 0);
		for (long i = s; i < e; i++)
			cnt[maxlen - (rp[i + 1] - rp[i]) + 1]++;
		for (int b = 0; b <= maxlen; b++)
			cnt[b + 1] += cnt[b];
		for (long i = s; i < e; i++)
			row_of_sorted[s + cnt[maxlen - (rp[i + 1] - rp[i])]++] = (int) i;
	}
	// pass 1: width and mode of every slice
	std::vector<int64_t> val_ptr((size_t) num_slices + 1, 0), idx_ptr((size_t) num_slices + 1, 0);
	std::vector<unsigned char> mode((size_t) std::max<long>(num_slices, 1), 4);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 64)
	for (long sl = 0; sl < num_slices; sl++)
	{
		long width = 0;
		const long i_e = std::min(m, (sl + 1) * C);
		for (long i = sl * C; i < i_e; i++)
		{
			int o = row_of_sorted[i];
			width = std::max<long>(width, rp[o + 1] - rp[o]);
		}
		const long maxlen = width;
		width = (width + 3) / 4 * 4;
		long maxdelta = 0;
		// step-invariant lane offsets: a full slice of equally long rows whose step-k columns are c_k[lane 0] + off[lane] with the
		// SAME off for every step (rows of one kind of a stencil: column = row + const_k). off = lane is the affine case.
		bool rowoff = (sl + 1) * C <= m && A->n >= C && maxlen > 0;
		for (long i = sl * C; i < i_e && rowoff; i++)
		{
			int o = row_of_sorted[i];
			rowoff = (rp[o + 1] - rp[o]) == maxlen;
		}
		bool affine = (sl + 1) * C <= m && A->n >= C && maxlen == 0;      // an all-empty slice stores nothing either
		if (rowoff)
		{
			const int o0 = row_of_sorted[sl * C];
			affine = true;
			for (long i = sl * C; i < i_e && rowoff; i++)
			{
				const int o = row_of_sorted[i];
				const int off = ci[rp[o]] - ci[rp[o0]];
				if (off != (int) (i - sl * C))
					affine = false;
				for (long k = 1; k < maxlen && rowoff; k++)
					rowoff = ci[rp[o] + k] - ci[rp[o0] + k] == off;
			}
			if (!rowoff)
				affine = false;
		}
		for (long k = 0; k < width; k++)
		{
			int lo = 0x7fffffff, hi = -1;
			for (long i = sl * C; i < i_e; i++)
			{
				int o = row_of_sorted[i];
				if (k < rp[o + 1] - rp[o])
				{
					int c = ci[rp[o] + k];
					lo = std::min(lo, c);
					hi = std::max(hi, c);
				}
			}
			if (hi >= 0)
				maxdelta = std::max<long>(maxdelta, (long) hi - lo);
		}
		// lane offsets with exceptions (mode 5, kernels_sell.hip): equally long rows, not all of one pattern, but at least 48 of the
		// 64 agree with one of the first four lanes taken as the reference (the same rule as convert_sell.hip: same bytes)
		int ref = -1, nex = 0;
		{
			bool equal_len = (sl + 1) * C <= m && A->n >= C && maxlen > 0;
			for (long i = sl * C; i < i_e && equal_len; i++)
				equal_len = (rp[row_of_sorted[i] + 1] - rp[row_of_sorted[i]]) == maxlen;
			if (equal_len && !rowoff && !(sell_modes_off() & 4))
				for (int r = 0; r < 4 && ref < 0; r++)
				{
					bool hard;
					const int e = __builtin_popcountll(sell5_exceptions(rp, ci, row_of_sorted.data() + sl * C, maxlen, r, hard));
					if (C - e >= 48 && e > 0 && !hard)
					{
						ref = r;
						nex = e;
					}
				}
		}
		const int md = (affine && !(sell_modes_off() & 1)) ? 0 : (rowoff && !(sell_modes_off() & 2)) ? 3 : ref >= 0 ? 5 : maxdelta < 256 ? 1 : maxdelta < 65536 ? 2 : 4;
		mode[sl] = (unsigned char) (md == 5 ? (5 | ref << 3) : md);
		val_ptr[sl + 1] = maxlen * C;                    // values: exact width; index groups: rounded up to 4 steps
		idx_ptr[sl + 1] = md == 5 ? (4 * C + 16) + (width / 4) * (16 + (nex + 3) / 4 * 16)
		                          : (md == 3 ? 4 * C : 0) + (width / 4) * ((md == 0 || md == 3) ? 16 : md == 1 ? 272 : md == 2 ? 528 : 1024);
	}
	for (long sl = 0; sl < num_slices; sl++)
	{
		val_ptr[sl + 1] += val_ptr[sl];
		idx_ptr[sl + 1] += idx_ptr[sl];
		A->sell_mode_slices[((mode[sl] & 7) == 0 || (mode[sl] & 7) == 3 || (mode[sl] & 7) == 5) ? 3 : (mode[sl] & 7) == 1 ? 0 : (mode[sl] & 7) == 2 ? 1 : 2]++;
	}
	const int64_t nnz_ext = val_ptr[num_slices];
	const int64_t idx_bytes = idx_ptr[num_slices];
	std::vector<double> val((size_t) std::max<int64_t>(nnz_ext, 1));
	std::vector<unsigned char> idx((size_t) std::max<int64_t>(idx_bytes, 16) + 1024, 0);
	std::vector<int64_t> desc(2 * ((size_t) num_slices + 1), 0);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 64)
	for (long sl = 0; sl < num_slices; sl++)
	{
		const int64_t vb = val_ptr[sl];
		const long maxlen = (val_ptr[sl + 1] - vb) / C;
		const long width = (maxlen + 3) / 4 * 4;
		const int md = mode[sl] & 7, ref = mode[sl] >> 3;
		unsigned char * ib = idx.data() + idx_ptr[sl];
		desc[2 * sl] = vb;
		desc[2 * sl + 1] = idx_ptr[sl] | md;
		const long i_e = std::min(m, (sl + 1) * C);
		int min_off = 0;
		unsigned long long exmask = 0;
		int nex = 0;
		int off5[64];
		if (md == 5)
		{
			// header: the 64 lane offsets relative to the reference lane's first column, the exception mask, padding
			bool hard;
			exmask = sell5_exceptions(rp, ci, row_of_sorted.data() + sl * C, maxlen, ref, hard);
			nex = __builtin_popcountll(exmask);
			const int oref = row_of_sorted[sl * C + ref];
			for (int r = 0; r < C; r++)
			{
				off5[r] = ci[rp[row_of_sorted[sl * C + r]]] - ci[rp[oref]];
				reinterpret_cast<int *>(ib)[r] = off5[r];
				min_off = std::min(min_off, off5[r]);
			}
			reinterpret_cast<unsigned long long *>(ib + 4 * C)[0] = exmask;
			reinterpret_cast<unsigned long long *>(ib + 4 * C)[1] = 0ull;
			ib += 4 * C + 16;
		}
		if (md == 3)
		{
			// header: the 64 lane offsets (relative to lane 0's column), then the groups of 4 bases
			const int o0 = row_of_sorted[sl * C];
			for (int r = 0; r < C; r++)
			{
				const int off = ci[rp[row_of_sorted[sl * C + r]]] - ci[rp[o0]];
				reinterpret_cast<int *>(ib)[r] = off;
				min_off = std::min(min_off, off);
			}
			ib += 4 * C;
		}
		for (long k = 0; k < width; k++)
		{
			int base = 0x7fffffff;
			for (long i = sl * C; i < i_e; i++)
			{
				int o = row_of_sorted[i];
				if (k < rp[o + 1] - rp[o])
					base = std::min(base, ci[rp[o] + k]);
			}
			if (base == 0x7fffffff)
				base = 0;                              // a step that is padding for every lane
			if (md == 3)                               // base + off[lane] must be lane 0's column (real step) / a valid column (padding)
				base = (k < rp[row_of_sorted[sl * C] + 1] - rp[row_of_sorted[sl * C]]) ? ci[rp[row_of_sorted[sl * C]] + k] : -min_off;
			if (md == 5)                               // ... the reference lane's
				base = k < maxlen ? ci[rp[row_of_sorted[sl * C + ref]] + k] : -min_off;
			const long g = k / 4, u = k % 4;
			const long gbytes = md == 5 ? 16 + (nex + 3) / 4 * 16 : (md == 0 || md == 3) ? 16 : md == 1 ? 272 : md == 2 ? 528 : 1024;
			unsigned char * gp = ib + g * gbytes;
			if (md != 4)
				reinterpret_cast<int *>(gp)[u] = base;
			for (int r = 0; r < C; r++)
			{
				const long i = sl * C + r;
				double v = 0.0;
				int c = base;                              // padding: value 0 times a column some lane really uses
				if (i < m)
				{
					int o = row_of_sorted[i];
					if (k < rp[o + 1] - rp[o])
					{
						c = ci[rp[o] + k];
						v = va[rp[o] + k];
					}
				}
				if (k < maxlen)
					val[vb + sell_pair_pos(k, maxlen, r)] = v;   // steps past the longest row exist in the index groups only
				const unsigned d = (unsigned) (c - base);
				if (md == 5)
				{
					// an exception lane's correction of this step: one signed byte at 4 * (its rank among the exception lanes) + u
					// (the idx array starts out zeroed: padding steps and the tail of the 16-byte-padded group stay 0)
					if (((exmask >> r) & 1ull) && k < maxlen)
						reinterpret_cast<signed char *>(gp + 16)[4 * __builtin_popcountll(exmask & ((1ull << r) - 1ull)) + u] = (signed char) (c - (base + off5[r]));
					continue;
				}
				if (md == 0 || md == 3)
					continue;                                  // column = base + lane offset, nothing stored per lane and step
				if (md == 1)
					gp[16 + r * 4 + u] = (unsigned char) d;
				else if (md == 2)
					reinterpret_cast<unsigned short *>(gp + 16)[r * 4 + u] = (unsigned short) d;
				else
					reinterpret_cast<int *>(gp)[u * C + r] = c;
			}
		}
	}
	desc[2 * num_slices] = nnz_ext;
	desc[2 * num_slices + 1] = idx_bytes | 4;
	A->sell_slices = num_slices;
	A->sell_nnz_ext = nnz_ext;
	A->sell_idx_bytes = idx_bytes;
	{
		const long spt = sell_slices_per_tile() / A->sell_split;       // slices per workgroup
		A->cfg.map = xcd_map_balanced(val_ptr.data(), num_slices, spt, resolve_remap(A->remap, (num_slices + spt - 1) / spt));
	}
	if (dev_alloc(&A->d_sell_desc, desc.size()))
		return 1;
	HIP_TRY(hipMemcpy(A->d_sell_desc, desc.data(), desc.size() * sizeof(int64_t), hipMemcpyHostToDevice));
	if (dev_alloc(&A->d_sell_idx, idx.size()))
		return 1;
	HIP_TRY(hipMemcpy(A->d_sell_idx, idx.data(), idx.size(), hipMemcpyHostToDevice));
	if (upload_values(A, val.data(), (size_t) nnz_ext, &A->d_val))
		return 1;
	if (upload_ints(row_of_sorted.data(), (size_t) m, &A->d_row_of_sorted))
		return 1;
	A->mem_footprint = (double) (num_slices + 1) * 16 + (double) nnz_ext * A->vbytes + (double) idx_bytes + (double) m * 4;
	return 0;
}

// SELL-64 with the slice group's x window in LDS (kernels_sell_window.hip). Groups of NS consecutive slices = contiguous row
// ranges of 64*NS rows, sorted by length inside the group (descending, stable: radix_sort.c:103-122 semantics with
// sigma = 64*NS). Returns 0 = built, 1 = error, 2 = not applicable (a group's window is wider than 65 536 columns or than
// the LDS budget): the caller falls back to the delta layout.
static int
build_sell_window(spmv_mi355x_matrix * A, const int * rp, const int * ci, const double * va, int NS, int S, long lds_budget_bytes, bool sym = false)
{
	const long m = A->m;
	constexpr int C = 64;
	const long num_slices = (m + C - 1) / C;
	const long num_groups = (num_slices + NS - 1) / NS;
	const long sigma = (long) NS * C;
	if (A->convert_on_device)
	{
		// the GPU builder (convert_sell.hip): the same bytes, without the host passes over the non-zeros
		std::vector<int64_t> sdesc;
		int max_w = 0;
		const int took = sell_window_convert_device(A->f32, m, (long) rp[m], NS, sym, lds_budget_bytes, rp, ci, va, &A->d_sellw_grp, &A->d_row_of_sorted, &A->d_sell_desc,
				(unsigned short **) &A->d_sell_idx, &A->d_val, sdesc, &max_w);
		if (took)
			return took;
		const int64_t nnz_ext = sdesc[2 * (size_t) num_slices], idx_count = sdesc[2 * (size_t) num_slices + 1];
		A->sell_slices = num_slices;
		A->sell_nnz_ext = nnz_ext;
		A->sell_idx_bytes = idx_count * 2;
		A->sellw_groups = (int) num_groups;
		A->sellw_ns = NS;
		A->sell_split = S;
		A->sellw_lds = (int) (((long) (max_w + 1) * A->vbytes + 15) / 16 * 16);
		std::vector<int64_t> gp((size_t) num_groups + 1, 0);
		for (long g = 0; g < num_groups; g++)
			gp[(size_t) g + 1] = sdesc[2 * (size_t) std::min<long>(num_slices, (g + 1) * NS)];
		A->cfg.map = xcd_map_balanced(gp.data(), num_groups, 1, resolve_remap(A->remap, num_groups));
		A->mem_footprint = (double) (num_slices + 1) * 16 + (double) num_groups * 16 + (double) nnz_ext * A->vbytes + (double) idx_count * 2 + (double) m * 4;
		return 0;
	}
	// ---- windows
	std::vector<int> grp((size_t) num_groups * 4, 0);
	long too_wide = 0;
	int max_w = 0;
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 16) reduction(+ : too_wide) reduction(max : max_w)
	for (long g = 0; g < num_groups; g++)
	{
		const long r0 = g * sigma, r1 = std::min(m, r0 + sigma);
		int lo = 0x7fffffff, hi = -1;
		for (long j = rp[r0]; j < rp[r1]; j++)
		{
			lo = std::min(lo, ci[j]);
			hi = std::max(hi, ci[j]);
		}
		if (sym)
		{
			// symmetric storage: the window also covers the group's own rows (x[i] of every row is read, y[i] written through it)
			lo = std::min<long>(lo, r0);
			hi = std::max<long>(hi, r1 - 1);
		}
		if (hi < 0)
			lo = 0;
		const long w = hi < 0 ? 1 : (long) hi - lo + 1;
		// one more LDS slot behind the window holds a zero; symmetric storage keeps an fp64 y window beside the x window
		if (w > 65534 || (w + 1) * (long) (A->vbytes + (sym ? 8 : 0)) > lds_budget_bytes)
			too_wide++;
		grp[(size_t) 4 * g] = lo;
		grp[(size_t) 4 * g + 1] = (int) std::min<long>(w, 0x7fffffffL);
		grp[(size_t) 4 * g + 2] = (int) (g * NS);
		grp[(size_t) 4 * g + 3] = (int) std::min<long>(NS, num_slices - g * NS);
		max_w = std::max(max_w, grp[(size_t) 4 * g + 1]);
	}
	if (too_wide)
		return 2;
	// ---- sort inside the groups, slice widths
	std::vector<int> row_of_sorted((size_t) m);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 16)
	for (long g = 0; g < num_groups; g++)
	{
		const long s0 = g * sigma, e0 = std::min(m, s0 + sigma);
		int maxlen = 0;
		for (long i = s0; i < e0; i++)
			maxlen = std::max(maxlen, rp[i + 1] - rp[i]);
		std::vector<long> cnt((size_t) maxlen + 2, 0);
		for (long i = s0; i < e0; i++)
			cnt[maxlen - (rp[i + 1] - rp[i]) + 1]++;
		for (int b = 0; b <= maxlen; b++)
			cnt[b + 1] += cnt[b];
		for (long i = s0; i < e0; i++)
			row_of_sorted[s0 + cnt[maxlen - (rp[i + 1] - rp[i])]++] = (int) i;
	}
	std::vector<int64_t> sdesc(2 * ((size_t) num_slices + 1), 0);
	for (long sl = 0; sl < num_slices; sl++)
	{
		const int o = row_of_sorted[sl * C];                       // the slice's longest row is its first
		const long width = rp[o + 1] - rp[o];
		sdesc[2 * (sl + 1)] = sdesc[2 * sl] + (width + 3) / 4 * 4 * C;           // values and indices: whole groups of 4 steps
		sdesc[2 * (sl + 1) + 1] = sdesc[2 * sl + 1] + (width + 3) / 4 * 4 * C;
	}
	const int64_t nnz_ext = sdesc[2 * num_slices], idx_count = sdesc[2 * num_slices + 1];
	std::vector<double> val((size_t) std::max<int64_t>(nnz_ext, 1));
	std::vector<unsigned short> idx((size_t) std::max<int64_t>(idx_count, 1), 0);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 64)
	for (long sl = 0; sl < num_slices; sl++)
	{
		const int lo = grp[(size_t) 4 * (sl / NS)], gw = grp[(size_t) 4 * (sl / NS) + 1];
		const int64_t vb = sdesc[2 * sl], ib = sdesc[2 * sl + 1];
		const long width = (sdesc[2 * sl + 2] - vb) / C;              // a multiple of 4
		for (int r = 0; r < C; r++)
		{
			const long i = sl * C + r;
			long js = 0, len = 0;
			if (i < m)
			{
				const int o = row_of_sorted[i];
				js = rp[o];
				len = rp[o + 1] - rp[o];
			}
			// padding: value 0 times a window entry the row already reads (its last column); an EMPTY row reads the zero the kernel
			// keeps behind the window (its y must be 0 whatever x holds: 0 * Inf from a neighbour's column would make it NaN)
			// (symmetric storage: every padding entry points at the spare slot — its mirrored addition must not land in a real row)
			const unsigned short pad = (len > 0 && !sym) ? (unsigned short) (ci[js + len - 1] - lo) : (unsigned short) gw;
			for (long k = 0; k < width; k++)
			{
				val[(size_t) (vb + sellw_val_pos(k, r, A->f32))] = k < len ? va[js + k] : 0.0;
				idx[(size_t) (ib + (k / 4) * 4 * C + r * 4 + k % 4)] = k < len ? (unsigned short) (ci[js + k] - lo) : pad;
			}
		}
	}
	A->sell_slices = num_slices;
	A->sell_nnz_ext = nnz_ext;
	A->sell_idx_bytes = idx_count * 2;
	A->sellw_groups = (int) num_groups;
	A->sellw_ns = NS;
	A->sell_split = S;
	A->sellw_lds = (int) (((long) (max_w + 1) * A->vbytes + 15) / 16 * 16);
	{
		// XCD map over the groups, balanced by their stored entries
		std::vector<int64_t> gp((size_t) num_groups + 1, 0);
		for (long g = 0; g < num_groups; g++)
			gp[(size_t) g + 1] = sdesc[2 * std::min<long>(num_slices, (g + 1) * NS)];
		A->cfg.map = xcd_map_balanced(gp.data(), num_groups, 1, resolve_remap(A->remap, num_groups));
	}
	if (upload_ints(grp.data(), grp.size(), &A->d_sellw_grp) || dev_alloc(&A->d_sell_desc, sdesc.size()))
		return 1;
	HIP_TRY(hipMemcpy(A->d_sell_desc, sdesc.data(), sdesc.size() * sizeof(int64_t), hipMemcpyHostToDevice));
	if (upload_bytes(idx.data(), (size_t) idx_count * 2, 1024, (void **) &A->d_sell_idx) || upload_values(A, val.data(), (size_t) nnz_ext, &A->d_val) ||
	    upload_ints(row_of_sorted.data(), (size_t) m, &A->d_row_of_sorted))
		return 1;
	A->mem_footprint = (double) (num_slices + 1) * 16 + (double) num_groups * 16 + (double) nnz_ext * A->vbytes + (double) idx_count * 2 + (double) m * 4;
	return 0;
}

// symmetric storage in (opts.symmetric_input), SELL-C-sigma asked for: the stored triangle in the LDS-window layout, multiplied without
// expanding it (sell_window_sym_kernel). A->m / n / nnz describe the EXPANDED matrix (what rows() / nnz() report), rp / ci / va the
// triangle. Returns 0 = built, 1 = error, 2 = not applicable (some slice group's window of rows + columns does not fit LDS or 16 bits):
// the caller then expands the triangle and takes the general path.
int
build_sell_symmetric(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va)
{
	const long lm = A->m;
	const long slices = (lm + 63) / 64;
	const char * pf = A->f32 ? "f" : "d";
	int S = o.sell_split ? o.sell_split : (slices >= 8192 ? 1 : slices >= 2048 ? 2 : 4);
	int NS = o.sell_group ? o.sell_group : 16 / S;
	if ((S != 1 && S != 2 && S != 4) || NS < 1 || NS * S > 16 || (NS & (NS - 1)))
		return 2;
	A->sell_c = 64;
	A->sell_delta = false;
	A->convert_on_device = o.convert_on != 2 && !getenv("SPMV_MI355X_HOST_CONVERT");
	const int took = build_sell_window(A, rp, ci, va, NS, S, 152 * 1024, true);
	if (took)
		return took;
	A->sell_window = true;
	A->sell_sym = true;
	A->sell_sigma = 64L * NS;
	if (o.nontemporal == 0)
		A->cfg.nt = (double) A->sell_nnz_ext * (A->vbytes + 2) > 32.0 * 1024 * 1024 ? 1 : 0;
	if (S > 1)
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELLWS_64_%d_w%d_%s", 64 * NS, S, pf);
	else
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELLWS_64_%d_%s", 64 * NS, pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), "sell_window_sym_kernel");
	A->kernel_block = 64 * NS * S;
	return 0;
}

int
build_sell_family(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va)
{
	const long lm = A->m;
	const char * pf = A->f32 ? "f" : "d";
	int rc = 0;
	// auto: one row per lane (C = 64, bit-exact) when there are enough slices to fill the chip several times over,
	// else 16-row slices with 4 lanes per row (4x the wavefronts, 1/4 of the dependent chain) — profiles/sweep_r01.md
	int C = o.sell_c ? o.sell_c : 64;
	if (C != 16 && C != 32 && C != 64 && C != 256)
	{
		set_error("sell_c must be 16, 32, 64 or 256 (got %d)", C);
		return 1;
	}
	long sigma = o.sell_sigma ? o.sell_sigma : 16384;
	if (sigma < C || sigma % C)
	{
		set_error("sell_sigma (%ld) must be a positive multiple of sell_c (%d)", sigma, C);
		return 1;
	}
	A->sell_c = C;
	A->sell_sigma = sigma;
	A->sell_delta = (C == 64) && (o.sell_delta != 2);      // 0 = auto (on for 64-row slices), 1 = on, 2 = off
	A->convert_on_device = o.convert_on != 2 && !getenv("SPMV_MI355X_HOST_CONVERT");
	{
		// waves per slice: enough wavefronts to occupy 256 CUs several times over
		const long slices = (lm + 63) / 64;
		int S = o.sell_split ? o.sell_split : (slices >= 16384 ? 1 : slices >= 8192 ? 2 : 4);
		if (S != 1 && S != 2 && S != 4 && !(S == 8 && o.sell_window == 1))
		{
			set_error("sell_split must be 1, 2 or 4 (8 with sell_window = 1) (got %d)", S);
			return 1;
		}
		A->sell_split = A->sell_delta ? S : 1;
	}
	if (o.sell_delta == 1 && C != 64)
	{
		set_error("sell_delta needs sell_c = 64 (one lane per row)");
		return 1;
	}
	// ---- x window in LDS + 16-bit indices (banded / FEM matrices): sell_window 0 = auto, 1 = on (error when not applicable), 2 = off
	// (auto only when sell_sigma, sell_delta and convert_on are all at their defaults: the window layout sorts inside a slice group, i.e. with
	// its own sigma = 64 * slices per group — a caller who names a sigma, e.g. for the a6'/a7 layout parity entry point, gets that sigma)
	if (C == 64 && o.sell_window != 2 && (o.sell_window == 1 || (o.sell_delta == 0 && o.convert_on == 0 && o.sell_sigma == 0)))
	{
		const long slices = (lm + 63) / 64;
		const double mean = lm > 0 ? (double) A->nnz / lm : 0;
		// slices per workgroup: from sell_sigma when given, else enough groups for two per CU; waves per slice: enough wavefronts
		// to occupy the chip when the matrix has few slices
		int NS = o.sell_group ? o.sell_group : o.sell_sigma ? (int) std::min<long>(16, std::max<long>(1, sigma / 64)) : 0;
		int S = o.sell_split ? o.sell_split : (slices >= 8192 ? 1 : slices >= 2048 ? 2 : 4);      // pwtk twin (3 405 slices): 2 x 8 16.5 us, 1 x 16 16.2-19.7 us
		if (NS == 0)
			NS = 16 / S;          // 16 wavefronts per workgroup: measured best on the cant (4 x 4) and pwtk (16 x 1) twins — the window is
			                      // copied once per workgroup, so fewer, larger groups copy less
		if (NS < 1 || NS * S > 16 || (NS & (NS - 1)))
		{
			if (o.sell_window == 1)
			{
				set_error("sell_window: %d slices x %d waves per workgroup (a power of two, at most 16 waves)", NS, S);
				return 1;
			}
		}
		else if (o.sell_window == 1 || (mean >= 8 && A->nnz >= (1L << 20) && A->nnz < (1L << 28)))
		{
			const int took = build_sell_window(A, rp, ci, va, NS, S, sell_window_lds_budget());
			if (took == 1)
				return 1;
			if (took == 0)
			{
				A->sell_window = true;
				A->sell_delta = false;
				// Matrix streams of this kernel are read once per launch and x lives in LDS: when an XCD's share of the stream
				// exceeds its 4 MiB L2 (stream > 32 MiB), letting it allocate there only evicts the window sources and descriptors —
				// nontemporal loads: cant twin fp64 (41 MB) 9.4 -> 8.5 us; below that size the stream itself stays L2-resident from
				// launch to launch and nt would throw that away (cant twin fp32, 21 MB: 3.9 -> 5.6 us)
				if (o.nontemporal == 0)
					A->cfg.nt = (double) A->sell_nnz_ext * (A->vbytes + 2) > 32.0 * 1024 * 1024 ? 1 : 0;
				A->sell_sigma = 64L * NS;
				if (S > 1)
					snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELLW_64_%d_w%d_%s", 64 * NS, S, pf);
				else
					snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELLW_64_%d_%s", 64 * NS, pf);
				snprintf(A->kernel_name, sizeof(A->kernel_name), "sell_window_kernel");
				A->kernel_block = 64 * NS * S;
				return 0;
			}
			if (o.sell_window == 1)
			{
				set_error("sell_window: a slice group's column window exceeds 65 536 columns or the LDS budget");
				return 1;
			}
		}
	}
	rc = A->sell_delta ? build_sell_delta(A, rp, ci, va) : build_sell(A, rp, ci, va);
	if (A->sell_delta && A->sell_split > 1)
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELLD_%d_%ld_w%d_%s", C, sigma, A->sell_split, pf);
	else
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELL%s_%d_%ld_%s", A->sell_delta ? "D" : "", C, sigma, pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), A->sell_delta ? "sell_delta_kernel" : "sell_kernel");
	return rc;
}


// SELL-64-sigma-delta from a CSR that is already in device memory (csr_stream.hip: rows appended piece by piece, so that the host never
// holds the whole matrix): the GPU conversion of build_sell_delta without the upload. Same layout, same names, same results as
// spmv_mi355x_create() gives for sell_c = 64 with the delta layout converted on the device.
int
build_sell_delta_resident(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * d_rp, const int * d_ci, const double * d_va)
{
	constexpr int C = 64;
	const long m = A->m;
	const char * pf = A->f32 ? "f" : "d";
	if ((o.sell_c && o.sell_c != C) || o.sell_delta == 2 || o.convert_on == 2 || o.sell_window == 1)
	{
		set_error("create_from_stream: only the SELL-64 delta layout converted on the device is built from a device-resident CSR");
		return 1;
	}
	const long sigma = o.sell_sigma ? o.sell_sigma : 16384;
	if (sigma < C || sigma % C)
	{
		set_error("sell_sigma (%ld) must be a positive multiple of sell_c (%d)", sigma, C);
		return 1;
	}
	const long num_slices = (m + C - 1) / C;
	int S = o.sell_split ? o.sell_split : (num_slices >= 16384 ? 1 : num_slices >= 8192 ? 2 : 4);
	if (S != 1 && S != 2 && S != 4)
	{
		set_error("sell_split must be 1, 2 or 4 (got %d)", S);
		return 1;
	}
	A->sell_c = C;
	A->sell_sigma = sigma;
	A->sell_delta = true;
	A->convert_on_device = true;
	A->sell_split = S;
	std::vector<int64_t> val_ptr;
	int64_t nnz_ext = 0, idx_bytes = 0;
	void * d_val = nullptr;
	if (sell_delta_convert_resident(A->f32, m, A->n, A->nnz, sigma, d_rp, d_ci, d_va, &A->d_row_of_sorted, &A->d_sell_desc, &A->d_sell_idx, &d_val,
			val_ptr, A->sell_mode_slices, &nnz_ext, &idx_bytes))
		return 1;
	A->d_val = d_val;
	A->sell_slices = num_slices;
	A->sell_nnz_ext = nnz_ext;
	A->sell_idx_bytes = idx_bytes;
	const long spt = sell_slices_per_tile() / A->sell_split;
	A->cfg.map = xcd_map_balanced(val_ptr.data(), num_slices, spt, resolve_remap(A->remap, (num_slices + spt - 1) / spt));
	A->mem_footprint = (double) (num_slices + 1) * 16 + (double) nnz_ext * A->vbytes + (double) idx_bytes + (double) m * 4;
	if (A->sell_split > 1)
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELLD_%d_%ld_w%d_%s", C, sigma, A->sell_split, pf);
	else
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELLD_%d_%ld_%s", C, sigma, pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), "sell_delta_kernel");
	return 0;
}

}  // namespace spmv
