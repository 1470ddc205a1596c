// C ABI of the MI355X SpMV engine (include/spmv_mi355x.h): handle management, format conversion
// (= the reference's csr_to_format constructors) and the spmv entry points. Host code only; the kernels live in
// kernels_*.hip. There is deliberately NO CPU compute path here: every y comes from a HIP kernel.

#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include <omp.h>

#include "../../include/spmv_mi355x.h"
#include "launch.hpp"

namespace spmv {

static thread_local char g_err[1024] = "";

void
set_error(const char * fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

}  // namespace spmv

using namespace spmv;

struct spmv_mi355x_matrix {
	int format = 0, precision = 0;
	long m = 0, n = 0, nnz = 0;            // local rows, columns, local non-zeros
	int device = 0;
	bool f32 = false;
	size_t vbytes = 8;
	LaunchCfg cfg{};
	int remap = 1;

	// CSR family
	int * d_row_ptr = nullptr;
	int * d_col = nullptr;
	void * d_val = nullptr;
	int lanes_per_row = 0;
	int rows_per_group = 1;                // CSR_VECTOR: rows a lane group keeps in flight (1, 2, 4)
	int * d_win_row = nullptr;             // CSR_STREAM mode 4: row block boundaries, window start, window length (0 = no LDS window)
	int * d_win_lo = nullptr;
	int * d_win_w = nullptr;
	unsigned short * d_col16 = nullptr;    // mode 4 with every window <= 65 536 columns: indices relative to the block's window
	int win_blocks = 0, win_lds_bytes = 0;
	int stream_mode = 0;                   // CSR_STREAM: 1 = products in LDS (row-major gather), 2 = (val,col) in LDS, lane-per-row walk
	// merge
	int merge_ipt = 0, merge_tile = 0, merge_num_tiles = 0;
	int * d_coords = nullptr;
	int * d_carry_row = nullptr;
	void * d_carry_val = nullptr;
	// column-blocked COO (opts.col_blocks): segments of rows, entries ordered by column block inside a segment
	int * d_coob_seg_row = nullptr;
	int * d_coob_seg_blk = nullptr;
	unsigned short * d_coob_lrow = nullptr;
	int coob_segs = 0, coob_blocks = 0, coob_lds = 0;
	// SELL
	int sell_c = 0;
	long sell_sigma = 0, sell_slices = 0, sell_nnz_ext = 0;
	int64_t * d_slice_ptr = nullptr;
	int * d_row_of_sorted = nullptr;
	bool sell_delta = false;               // delta-compressed column indices (C = 64 only)
	bool convert_on_device = true;         // build the delta layout on the GPU (convert_sell.hip) or on the host
	int sell_split = 1;                    // waves sharing one slice (delta format): 1, 2 or 4
	int64_t * d_sell_desc = nullptr;
	unsigned char * d_sell_idx = nullptr;
	long sell_idx_bytes = 0;
	long sell_mode_slices[4] = {0, 0, 0, 0};  // slices stored with 8-bit / 16-bit / 32-bit indices / none (affine)
	// COO
	int coo_k = 0, coo_num_waves = 0;
	int * d_rowind = nullptr;

	// host-buffer path
	void * d_x = nullptr;
	void * d_y = nullptr;
	const void * cached_x_host = nullptr;
	bool y_downloaded = false;
	bool always_copy = false;
	hipStream_t stream = nullptr;

	double mem_footprint = 0, csr_mem_footprint = 0;
	char format_name[96] = "";
	char kernel_name[64] = "";
	long last_grid = 0;
};

template <typename T>
static int
dev_alloc(T ** p, size_t count)
{
	*p = nullptr;
	if (count == 0)
		count = 1;
	HIP_TRY(hipMalloc((void **) p, count * sizeof(T)));
	return 0;
}

static int
dev_alloc_bytes(void ** p, size_t bytes)
{
	*p = nullptr;
	if (bytes == 0)
		bytes = 8;
	HIP_TRY(hipMalloc(p, bytes));
	return 0;
}

static void
free_all(spmv_mi355x_matrix * A)
{
	void * ptrs[] = {A->d_row_ptr, A->d_col, A->d_val, A->d_coords, A->d_carry_row, A->d_carry_val, A->d_slice_ptr,
	                 A->d_row_of_sorted, A->d_rowind, A->d_x, A->d_y, A->d_sell_desc, A->d_sell_idx, A->d_win_row, A->d_win_lo,
	                 A->d_win_w, A->d_col16, A->d_coob_seg_row, A->d_coob_seg_blk, A->d_coob_lrow};
	for (void * p : ptrs)
		if (p)
			(void) hipFree(p);
	if (A->stream)
		(void) hipStreamDestroy(A->stream);
}

// narrow fp64 reference values to the handle's precision (csr.cpp:72 `a[i] = values[i]`) and upload
static int
upload_values(spmv_mi355x_matrix * A, const double * v, size_t count, void ** d_out)
{
	// STREAM_SLACK spare entries: the LDS-DMA row-block copy reads whole 1 KiB chunks (kernels_csr_stream.hip)
	if (dev_alloc_bytes(d_out, (count + STREAM_SLACK) * A->vbytes))
		return 1;
	HIP_TRY(hipMemset((char *) *d_out + count * A->vbytes, 0, STREAM_SLACK * A->vbytes));
	if (count == 0)
		return 0;
	if (!A->f32)
	{
		HIP_TRY(hipMemcpy(*d_out, v, count * sizeof(double), hipMemcpyHostToDevice));
		return 0;
	}
	// chunked narrowing keeps the host staging buffer small for 10^9-entry matrices
	const size_t CH = (size_t) 1 << 26;
	std::vector<float> tmp(std::min(CH, count));
	for (size_t off = 0; off < count; off += CH)
	{
		size_t len = std::min(CH, count - off);
		#pragma omp parallel for num_threads(spmv::host_threads())
		for (long i = 0; i < (long) len; i++)
			tmp[i] = (float) v[off + i];
		HIP_TRY(hipMemcpy((char *) *d_out + off * sizeof(float), tmp.data(), len * sizeof(float), hipMemcpyHostToDevice));
	}
	return 0;
}

static int
upload_ints(const int * src, size_t count, int ** d_out)
{
	if (dev_alloc(d_out, count + STREAM_SLACK))
		return 1;
	HIP_TRY(hipMemset(*d_out + count, 0, STREAM_SLACK * sizeof(int)));
	if (count)
		HIP_TRY(hipMemcpy(*d_out, src, count * sizeof(int), hipMemcpyHostToDevice));
	return 0;
}

static int
pick_lanes_per_row(double mean)
{
	// measured on the five BASELINE.json twins (profiles/sweep_r01.md): 8..16 lanes win from 5.6 to 64 nnz/row — a
	// wider group only adds idle lanes and butterfly steps, a narrower one serialises the row
	if (mean <= 4) return 4;
	if (mean <= 12) return 8;
	if (mean <= 128) return 16;
	if (mean <= 512) return 32;
	return 64;
}

// auto tile order: contiguous work-balanced ranges keep each XCD's L2 on one window of x (best for small and skewed
// matrices); for many-tile matrices chunks of 64 tiles dealt round-robin balance row-count-bound kernels better
// (nlpkkt240 twin: csr_vector +26 %, csr_stream +12 %, SELL +2 %; pwtk/soc-LiveJournal1 twins prefer the ranges)
static int
resolve_remap(int requested, long ntiles)
{
	if (requested >= 0)
		return requested;
	return ntiles >= 8192 ? 2 : 1;
}

// ---------------------------------------------------------------------------------------------------- SELL build
// Host-side CSR -> SELL-C-sigma (the reference converts on the host too: sell_sorted.cpp:112-298, sellcs_format.c:137-200).
// Window sort: stable, DESCENDING row length inside each window of sigma rows (radix_sort.c:103-122 semantics).
static int
build_sell(spmv_mi355x_matrix * A, const int * rp, const int * ci, const double * va)
{
	const long m = A->m;
	const int C = A->sell_c;
	const int TPR = WAVE / C;
	const long sigma = A->sell_sigma;
	const long num_slices = (m + C - 1) / C;
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
	A->cfg.map = xcd_map_balanced(slice_ptr.data(), num_slices, sell_slices_per_tile(),
			resolve_remap(A->remap, (num_slices + sell_slices_per_tile() - 1) / sell_slices_per_tile()));
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

// SELL-64-sigma-delta build (layout: kernels_sell.hip). Same sigma-window sort and slice widths as build_sell with C = 64,
// widths padded to a multiple of 4 steps; per slice the narrowest index encoding that holds every (step, lane) delta.
// Column-blocked COO (opts.col_blocks, kernels_coo.hip): rows cut into segments of about equal non-zeros and at most
// coo_blocked_rows_cap() rows (their y lives in LDS); inside a segment the entries are ordered by column block (stable: rows
// ascending, columns ascending inside a block) and stored as (column, row inside the segment, value unless all equal).
static int
build_coo_blocked(spmv_mi355x_matrix * A, const int * rp, const int * ci, const double * va, int col_blocks, const char * pf)
{
	const long lm = A->m, lnnz = A->nnz, n = A->n;
	const int cap = coo_blocked_rows_cap(A->f32);
	const long per = coo_blocked_segments_per_launch();
	// launches: enough segments that the row cap is rarely what ends one
	const long launches = std::max<long>(1, (long) ((double) lm / (0.75 * cap * per) + 0.999));
	// equal shares of the entries (the workgroups must keep pace), then any share with more than `cap` rows is cut further
	std::vector<int> seg_row(1, 0);
	{
		const long S0 = per * launches;
		int prev = 0;
		for (long sgm = 1; sgm <= S0; sgm++)
		{
			const long tgt = (long) ((double) lnnz * sgm / S0);
			long r = std::lower_bound(rp, rp + lm + 1, (int) std::min<long>(tgt, 0x7fffffffL)) - rp;
			const int end = (int) (sgm == S0 ? lm : std::min<long>(std::max<long>(r, prev), lm));
			for (int a = prev; a < end; a += cap)
				seg_row.push_back(std::min(a + cap, end));
			prev = end;
		}
	}
	const long S = (long) seg_row.size() - 1;
	long W;
	int B;
	if (col_blocks > 0)
	{
		B = (int) std::min<long>(col_blocks, std::max<long>(n, 1));
		W = std::max<long>(1, (n + B - 1) / B);
	}
	else
	{
		W = std::max<long>(1024, (384L << 10) / A->vbytes);          // ~384 KiB of x per block: with the per-block barrier 96-128 blocks are best on the soc-LiveJournal1 twin (420-440 us; 37: 470, 192: 495, 256: 600)
		B = (int) std::max<long>(1, (n + W - 1) / W);
	}
	B = (int) std::max<long>(1, (n + W - 1) / W);
	if (B > 4096)
	{
		set_error("col_blocks: %d column blocks (limit 4096)", B);
		return 1;
	}
	bool uniform = lnnz > 0;
	double v0 = 0;
	{
		long differs = 0;
		v0 = lnnz > 0 ? (A->f32 ? (double) (float) va[0] : va[0]) : 0.0;
		#pragma omp parallel for num_threads(spmv::host_threads()) reduction(+ : differs)
		for (long j = 0; j < lnnz; j++)
			differs += (A->f32 ? (double) (float) va[j] : va[j]) != v0;
		uniform = uniform && differs == 0 && v0 == v0;
	}
	std::vector<int> seg_blk((size_t) std::max<long>(S, 1) * (B + 1), 0);
	std::vector<int> pcol((size_t) lnnz);
	std::vector<unsigned short> plrow((size_t) lnnz);
	std::vector<double> pval(uniform ? 0 : (size_t) lnnz);
	#pragma omp parallel num_threads(spmv::host_threads())
	{
		std::vector<int> pos((size_t) B + 1);
		#pragma omp for schedule(dynamic, 8)
		for (long sgm = 0; sgm < S; sgm++)
		{
			const int r0 = seg_row[sgm], r1 = seg_row[sgm + 1];
			std::fill(pos.begin(), pos.end(), 0);
			for (long j = rp[r0]; j < rp[r1]; j++)
				pos[(size_t) (ci[j] / W) + 1]++;
			int * sb = seg_blk.data() + (size_t) sgm * (B + 1);
			sb[0] = rp[r0];
			for (int b = 0; b < B; b++)
				sb[b + 1] = sb[b] + pos[(size_t) b + 1];
			for (int b = 0; b <= B; b++)
				pos[(size_t) b] = sb[b];
			for (int r = r0; r < r1; r++)
				for (long j = rp[r]; j < rp[r + 1]; j++)
				{
					const int at = pos[(size_t) (ci[j] / W)]++;
					pcol[(size_t) at] = ci[j];
					plrow[(size_t) at] = (unsigned short) (r - r0);
					if (!uniform)
						pval[(size_t) at] = va[j];
				}
		}
	}
	int rc = upload_ints(seg_row.data(), seg_row.size(), &A->d_coob_seg_row) || upload_ints(seg_blk.data(), seg_blk.size(), &A->d_coob_seg_blk) ||
	         upload_ints(pcol.data(), (size_t) lnnz, &A->d_col) || dev_alloc_bytes((void **) &A->d_coob_lrow, ((size_t) lnnz + STREAM_SLACK) * 2);
	if (rc)
		return 1;
	if (lnnz)
		HIP_TRY(hipMemcpy(A->d_coob_lrow, plrow.data(), (size_t) lnnz * 2, hipMemcpyHostToDevice));
	if (!uniform && upload_values(A, pval.data(), (size_t) lnnz, &A->d_val))
		return 1;
	if (uniform)
	{
		A->cfg.unit = 1;
		A->cfg.unit_value = v0;
	}
	int max_rows = 0;
	for (long sgm = 0; sgm < S; sgm++)
		max_rows = std::max(max_rows, seg_row[sgm + 1] - seg_row[sgm]);
	A->coob_segs = (int) S;
	A->coob_blocks = B;
	A->coob_lds = (int) (((long) std::max(max_rows, 1) * A->vbytes + 15) / 16 * 16);
	A->cfg.map = xcd_map_uniform(1, 0);
	A->mem_footprint = (double) lnnz * (6 + (uniform ? 0 : A->vbytes)) + (S + 1) * 4.0 + (double) S * (B + 1) * 4;
	snprintf(A->format_name, sizeof(A->format_name), "MI355X_COOB_s%ld_b%d%s_%s", S, B, uniform ? "_unit" : "", pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), "coo_blocked_kernel");
	return 0;
}

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
		const int md = affine ? 0 : rowoff ? 3 : maxdelta < 256 ? 1 : maxdelta < 65536 ? 2 : 4;
		mode[sl] = (unsigned char) md;
		val_ptr[sl + 1] = maxlen * C;                    // values: exact width; index groups: rounded up to 4 steps
		idx_ptr[sl + 1] = (md == 3 ? 4 * C : 0) + (width / 4) * ((md == 0 || md == 3) ? 16 : md == 1 ? 272 : md == 2 ? 528 : 1024);
	}
	for (long sl = 0; sl < num_slices; sl++)
	{
		val_ptr[sl + 1] += val_ptr[sl];
		idx_ptr[sl + 1] += idx_ptr[sl];
		A->sell_mode_slices[(mode[sl] == 0 || mode[sl] == 3) ? 3 : mode[sl] == 1 ? 0 : mode[sl] == 2 ? 1 : 2]++;
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
		const int md = mode[sl];
		unsigned char * ib = idx.data() + idx_ptr[sl];
		desc[2 * sl] = vb;
		desc[2 * sl + 1] = idx_ptr[sl] | md;
		const long i_e = std::min(m, (sl + 1) * C);
		int min_off = 0;
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
			const long g = k / 4, u = k % 4;
			const long gbytes = (md == 0 || md == 3) ? 16 : md == 1 ? 272 : md == 2 ? 528 : 1024;
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
					val[vb + k * C + r] = v;                   // steps past the longest row exist in the index groups only
				const unsigned d = (unsigned) (c - base);
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

// ---------------------------------------------------------------------------------------------------- C ABI

extern "C" {

const char *
spmv_mi355x_last_error(void)
{
	return g_err;
}

int
spmv_mi355x_device_count(int * count_out)
{
	int c = 0;
	hipError_t e = hipGetDeviceCount(&c);
	if (e != hipSuccess)
	{
		(void) hipGetLastError();
		c = 0;
	}
	*count_out = c;
	return 0;
}

int
spmv_mi355x_device_info(int device, char * name_out, long name_n, int * compute_units_out, long * hbm_bytes_out)
{
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (name_out && name_n > 0)
		snprintf(name_out, name_n, "%s (%s)", prop.name, prop.gcnArchName);
	if (compute_units_out)
		*compute_units_out = prop.multiProcessorCount;
	if (hbm_bytes_out)
		*hbm_bytes_out = (long) prop.totalGlobalMem;
	return 0;
}

int
spmv_mi355x_create(spmv_mi355x_matrix ** out, int format, int precision, long m, long n, long nnz,
		const int32_t * row_ptr, const int32_t * col_idx, const double * values, const spmv_mi355x_opts * opts_in)
{
	*out = nullptr;
	spmv_mi355x_opts o;
	memset(&o, 0, sizeof(o));
	o.device = -1;
	if (opts_in)
	{
		size_t sz = std::min<size_t>(sizeof(o), (size_t) std::max(opts_in->struct_size, 0));
		if (sz < 8)
		{
			set_error("opts->struct_size not set");
			return 1;
		}
		memcpy(&o, opts_in, sz);
	}
	if (format < 0 || format >= SPMV_MI355X_NUM_FORMATS)
	{
		set_error("unknown format %d", format);
		return 1;
	}
	if (precision != SPMV_MI355X_F64 && precision != SPMV_MI355X_F32)
	{
		set_error("unknown precision %d", precision);
		return 1;
	}
	if (m < 0 || n < 0 || nnz < 0 || m >= 0x7fffffffL || n >= 0x7fffffffL || nnz >= 0x7fffffffL || m + nnz >= 0x7fffffffL)
	{
		set_error("sizes out of the int32 index range (m=%ld n=%ld nnz=%ld)", m, n, nnz);
		return 1;
	}
	if (!row_ptr || (nnz > 0 && (!col_idx || !values)))
	{
		set_error("NULL input array");
		return 1;
	}
	if (row_ptr[m] - row_ptr[0] != nnz)
	{
		set_error("row_ptr[m]-row_ptr[0] = %ld does not match nnz = %ld", (long) (row_ptr[m] - row_ptr[0]), nnz);
		return 1;
	}
	int ndev = 0;
	spmv_mi355x_device_count(&ndev);
	if (ndev < 1)
	{
		set_error("no HIP device available: this engine has no CPU fallback");
		return 1;
	}
	int device = o.device;
	if (device < 0)
		HIP_TRY(hipGetDevice(&device));
	if (device >= ndev)
	{
		set_error("device %d out of range (%d devices)", device, ndev);
		return 1;
	}
	HIP_TRY(hipSetDevice(device));

	// ---- symmetric storage in (KEEP_SYMMETRY builds of the harness; csr_sym.cpp:118-123 accepts exactly this): the arrays
	// hold ONE triangle; the product is y = (T + T^t - diag(T)) x, every stored off-diagonal (i, j, a) also acting as
	// (j, i, +a) — csr_sym.cpp:204-232, bench_spmv.cpp:135-148. The engine expands it and runs its general kernels:
	// scattering a*x[i] into y[j] with fp64 atomics runs at 24 G updates/s on MI355X for scattered j (175 G/s perfectly
	// coalesced; tools/atomic_bench.hip), an order of magnitude short of what halving the matrix stream would need.
	std::vector<int> e_rp, e_ci;
	std::vector<double> e_va;
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
		for (long i = 0; i < m; i++)
			if (row_ptr[i + 1] < row_ptr[i])
			{
				set_error("row_ptr is not monotone at row %ld", i);
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
		e_rp.assign((size_t) m + 1, 0);
		for (long i = 0; i < m; i++)
			e_rp[i + 1] = e_rp[i] + cnt[i + 1];
		e_ci.resize((size_t) std::max<long>(total, 1));
		e_va.resize((size_t) std::max<long>(total, 1));
		std::vector<int> pos(e_rp.begin(), e_rp.end() - 1);
		#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 4096)
		for (long i = 0; i < m; i++)
			for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++)
			{
				const int c = col_idx[j];
				int k;
				#pragma omp atomic capture
				k = pos[i]++;
				e_ci[k] = c;
				e_va[k] = values[j];
				if (c != i)
				{
					#pragma omp atomic capture
					k = pos[c]++;
					e_ci[k] = (int) i;
					e_va[k] = values[j];
				}
			}
		// rows ascending, columns ascending (what coo_to_csr gives the general path, csr_gen.c:178-213); equal columns are
		// ordered by value so the result does not depend on the thread interleaving above
		#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 1024)
		for (long i = 0; i < m; i++)
		{
			const long s0 = e_rp[i], len = e_rp[i + 1] - s0;
			bool sorted = true;
			for (long k = 1; k < len && sorted; k++)
				sorted = e_ci[s0 + k - 1] < e_ci[s0 + k];
			if (sorted)
				continue;
			std::vector<std::pair<int, double>> tmp((size_t) len);
			for (long k = 0; k < len; k++)
				tmp[k] = {e_ci[s0 + k], e_va[s0 + k]};
			std::sort(tmp.begin(), tmp.end());
			for (long k = 0; k < len; k++)
			{
				e_ci[s0 + k] = tmp[k].first;
				e_va[s0 + k] = tmp[k].second;
			}
		}
		row_ptr = e_rp.data();
		col_idx = e_ci.data();
		values = e_va.data();
		nnz = total;
	}

	spmv_mi355x_matrix * A = new spmv_mi355x_matrix();
	A->format = format;
	A->precision = precision;
	A->f32 = (precision == SPMV_MI355X_F32);
	A->vbytes = A->f32 ? 4 : 8;
	A->device = device;
	A->n = n;

	// ---- row block / column filter (row-partitioned multi-GPU) -> local CSR on the host
	long r0 = o.row_begin, r1 = o.row_end;
	if (r0 == 0 && r1 == 0)
		r1 = m;
	if (r0 < 0 || r1 > m || r0 > r1)
	{
		set_error("bad row block [%ld,%ld) for m=%ld", r0, r1, m);
		delete A;
		return 1;
	}
	const long lm = r1 - r0;
	std::vector<int> l_rp;
	std::vector<int> l_ci;
	std::vector<double> l_va;
	const int * rp;
	const int * ci;
	const double * va;
	long lnnz;
	const bool filter = o.col_filter_mode == 1 || o.col_filter_mode == 2;
	if (!filter && row_ptr[r0] == 0)
	{
		rp = row_ptr + r0;     // [0, r1) prefix: offsets are already local
		ci = col_idx;
		va = values;
		lnnz = row_ptr[r1];
	}
	else
	{
		l_rp.assign((size_t) lm + 1, 0);
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
			l_rp[i + 1] = cnt;
		}
		for (long i = 0; i < lm; i++)
			l_rp[i + 1] += l_rp[i];
		lnnz = l_rp[lm];
		l_ci.resize((size_t) std::max<long>(lnnz, 1));
		l_va.resize((size_t) std::max<long>(lnnz, 1));
		#pragma omp parallel for num_threads(spmv::host_threads())
		for (long i = 0; i < lm; i++)
		{
			long k = l_rp[i];
			for (long j = row_ptr[r0 + i]; j < row_ptr[r0 + i + 1]; j++)
			{
				if (filter)
				{
					bool in = col_idx[j] >= c0 && col_idx[j] < c1;
					if (in != inside)
						continue;
				}
				l_ci[k] = col_idx[j];
				l_va[k] = values[j];
				k++;
			}
		}
		rp = l_rp.data();
		ci = l_ci.data();
		va = l_va.data();
	}
	A->m = lm;
	A->nnz = lnnz;
	A->csr_mem_footprint = (double) lnnz * (A->vbytes + 4) + (double) (lm + 1) * 4;
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
		if (bad_col >= 0 || bad_row >= 0 || (lm > 0 && rp[0] != 0 && !l_rp.empty()))
		{
			if (bad_col >= 0)
				set_error("column index %d out of range [0,%ld) at entry %ld", ci[bad_col], n, bad_col);
			else
				set_error("row_ptr is not monotone at row %ld", bad_row);
			delete A;
			return 1;
		}
	}

	// ---- launch policy
	A->remap = (o.xcd_remap == 2) ? 0 : (o.xcd_remap == 3) ? 2 : (o.xcd_remap == 1) ? 1 : -1;   // -1 = auto, resolved per kernel
	const double stream_bytes = (double) lnnz * (A->vbytes + 4);
	A->cfg.nt = (o.nontemporal == 1) ? 1 : (o.nontemporal == 2) ? 0 : (stream_bytes > 192.0 * 1024 * 1024 ? 1 : 0);
	A->cfg.beta = 0;

	int rc = 0;
	const char * pf = A->f32 ? "f" : "d";
	switch (format)
	{
		case SPMV_MI355X_CSR_SCALAR:
		case SPMV_MI355X_CSR_VECTOR:
		case SPMV_MI355X_CSR_MERGE:
		case SPMV_MI355X_CSR_STREAM:
		{
			rc = upload_ints(rp, (size_t) lm + 1, &A->d_row_ptr) || upload_ints(ci, (size_t) lnnz, &A->d_col) ||
			     upload_values(A, va, (size_t) lnnz, &A->d_val);
			if (rc)
				break;
			A->mem_footprint = A->csr_mem_footprint;
			if (format == SPMV_MI355X_CSR_SCALAR)
			{
				A->cfg.map = xcd_map_balanced(rp, lm, csr_scalar_rows_per_tile(), resolve_remap(A->remap, lm / csr_scalar_rows_per_tile()));
				snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_SCALAR_%s", pf);
				snprintf(A->kernel_name, sizeof(A->kernel_name), "csr_scalar_kernel");
			}
			else if (format == SPMV_MI355X_CSR_STREAM)
			{
				int R = o.lanes_per_row;
				const double mean = lm > 0 ? (double) lnnz / lm : 0;
				int mode = o.stream_mode;
				// x window in LDS (kernels_csr_window.hip): nnz-balanced row blocks, a multiple of the 256 CUs. Forced by
				// stream_mode 4; in auto mode adopted when (nearly) every block's window fits the LDS budget and rows are long
				// enough to amortise the per-row butterfly (measured: pwtk twin fp32 22.1 -> 18.6 us; short-row / scattered
				// matrices keep the other modes). With 16-bit window-relative indices it also wins in fp64 (cant twin 11.2 -> 9.9 us,
				// pwtk twin 24.7 -> 23.7 us); fp64 with 32-bit indices ties with csr_vector and is not adopted automatically.
				auto try_window = [&](bool force) -> int {
					int G = R ? R : std::max(8, pick_lanes_per_row(mean));
					if (G != 8 && G != 16 && G != 32 && G != 64)
					{
						if (!force)
							return 0;
						set_error("csr_stream mode 4: lanes_per_row must be 8, 16, 32 or 64 (got %d)", G);
						return -1;
					}
					const int NG = 1024 / G;
					long nb = 256L * (o.merge_items > 0 ? o.merge_items : std::max(1L, std::min(8L, lnnz / (256L * 24576L))));
					nb = std::max(1L, std::min(nb, (lm + 2 * NG - 1) / (2 * NG)));
					std::vector<int> b_row, b_lo, b_w;
					std::vector<long> b_nnz;
					const long budget = csr_window_lds_budget() / (long) A->vbytes;
					long with_window = 0;
					int max_w = 0;
					for (int attempt = 0; attempt < 2; attempt++)
					{
						b_row.assign((size_t) nb + 1, 0);
						b_lo.assign((size_t) nb, 0);
						b_w.assign((size_t) nb, 0);
						b_nnz.assign((size_t) nb + 1, 0);
						for (long b = 0; b <= nb; b++)
						{
							const long target = (long) ((double) lnnz * b / nb);
							long r = std::lower_bound(rp, rp + lm + 1, (int) std::min<long>(target, 0x7fffffffL)) - rp;
							b_row[b] = (int) (b == 0 ? 0 : b == nb ? lm : std::min<long>(std::max<long>(r, b_row[b - 1]), lm));
							b_nnz[b] = rp[b_row[b]];
						}
						with_window = 0;
						max_w = 0;
						#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4) reduction(+ : with_window) reduction(max : max_w)
						for (long b = 0; b < nb; b++)
						{
							int lo = 0x7fffffff, hi = -1;
							for (long j = rp[b_row[b]]; j < rp[b_row[b + 1]]; j++)
							{
								lo = std::min(lo, ci[j]);
								hi = std::max(hi, ci[j]);
							}
							if (hi >= 0 && (long) hi - lo + 1 <= budget)
							{
								b_lo[b] = lo;
								b_w[b] = hi - lo + 1;
								with_window++;
								max_w = std::max(max_w, b_w[b]);
							}
						}
						// One 1024-thread block per CU is half the CU's wave slots. When the window is set by the matrix's
						// bandwidth rather than by the block (it does not shrink with the block) and two of them fit the CU's
						// 160 KiB of LDS, twice the blocks put two on every CU: pwtk twin fp32 (54 KiB windows) 17.8 -> 16.2 us.
						// Small blocks lose more than they gain (cant twin: 7.8 k non-zeros per block, 9.9 -> 13.0 us).
						if (attempt == 0 && o.merge_items == 0 && nb == 256 && with_window == nb && (long) max_w * A->vbytes > 16 * 1024 &&
						    2L * (((long) max_w * A->vbytes + 15) / 16 * 16) <= 144L * 1024 && lnnz / 512 >= 20000 && lm >= 2L * 512 * NG)
						{
							nb = 512;
							continue;
						}
						break;
					}
					// 16-bit window-relative indices need every non-empty block to have a window of at most 65 536 columns
					bool eligible16 = max_w <= 65536 && lnnz > 0;
					for (long b = 0; b < nb && eligible16; b++)
						eligible16 = b_w[b] > 0 || rp[b_row[b + 1]] == rp[b_row[b]];
					if (!force && (with_window * 100 < nb * 95 || mean < 16 || lnnz < (2L << 20) || !(A->f32 || eligible16)))
						return 0;
					A->stream_mode = 4;
					A->lanes_per_row = G;
					A->win_blocks = (int) nb;
					A->win_lds_bytes = (int) (((long) max_w * A->vbytes + 15) / 16 * 16);
					if (upload_ints(b_row.data(), (size_t) nb + 1, &A->d_win_row) || upload_ints(b_lo.data(), (size_t) nb, &A->d_win_lo) ||
					    upload_ints(b_w.data(), (size_t) nb, &A->d_win_w))
						return -1;
					A->cfg.map = xcd_map_balanced(b_nnz.data(), nb, 1, resolve_remap(A->remap, nb));
					A->mem_footprint += (3.0 * nb + 1) * 4;
					// every block has its window and none is wider than 65 536 columns: store the column indices relative to the
					// window in 16 bits (2 B per non-zero of stream instead of 4) and let go of the int32 array
					const bool short_idx = eligible16;
					if (short_idx)
					{
						std::vector<unsigned short> c16((size_t) lnnz);
						#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4)
						for (long b = 0; b < nb; b++)
							for (long j = rp[b_row[b]]; j < rp[b_row[b + 1]]; j++)
								c16[j] = (unsigned short) (ci[j] - b_lo[b]);
						if (dev_alloc_bytes((void **) &A->d_col16, ((size_t) lnnz + STREAM_SLACK) * 2))
							return -1;
						if (hipMemcpy(A->d_col16, c16.data(), (size_t) lnnz * 2, hipMemcpyHostToDevice) != hipSuccess ||
						    hipMemset(A->d_col16 + lnnz, 0, STREAM_SLACK * 2) != hipSuccess)
						{
							set_error("upload of the 16-bit column indices failed");
							return -1;
						}
						(void) hipFree(A->d_col);
						A->d_col = nullptr;
						A->mem_footprint -= 2.0 * lnnz;
					}
					snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_WINDOW%s_g%d_b%ld_w%ld_%s", short_idx ? "16" : "", G, nb,
							with_window * 100 / nb, pf);
					snprintf(A->kernel_name, sizeof(A->kernel_name), "csr_window_kernel");
					return 1;
				};
				if (mode == 4 || (mode == 0 && R == 0))
				{
					const int took = try_window(mode == 4);
					if (took < 0)
					{
						rc = 1;
						break;
					}
					if (took > 0)
						break;
				}
				if (mode < 1 || mode > 3)
				{
					// The lane-per-row walk (modes 2/3) pays when neighbouring rows touch neighbouring columns (stencil /
					// FEM matrices: one x gather instruction then hits a few lines). Estimate that on a sample of rows; with
					// scattered columns (graphs) the row-major product staging of mode 1 is the better CSR-Stream.
					long similar = 0, tried = 0;
					const long stride = std::max<long>(1, lm / 4096);
					for (long i = 0; i + 1 < lm; i += stride)
					{
						if (rp[i + 1] == rp[i] || rp[i + 2] == rp[i + 1])
							continue;
						tried++;
						long d = (long) ci[rp[i + 1]] - ci[rp[i]];
						similar += (d >= -2 && d <= 2);
					}
					mode = (tried == 0 || 4 * similar >= tried) ? 3 : 1;      // twins: stencil/FEM 0.8-1.0, circuit 0.38, social graph 0.13
				}
				A->stream_mode = mode;
				if (R == 0)
				{
					if (mode == 1)
					{
						// largest power of two with R * mean nnz/row <= 60% of the LDS strip
						R = 16;                     // more rows per wave only lengthen the per-lane LDS walk (measured)
						while (R > 4 && R * mean > 0.6 * csr_stream_cap())
							R /= 2;
					}
					else
					{
						// largest R <= 16 whose row blocks overflow the LDS strip (slow path) in at most 0.5 % of the cases:
						// measured optimum is a block of ~150-450 non-zeros per wave (profiles/sweep_r01.md)
						for (R = (mode == 3 ? 16 : 32); R > (mode == 3 ? 4 : 8); R /= 2)
						{
							const long cap = mode == 3 ? csr_stream_d_cap(R) : csr_stream_t_cap(R);
							long over = 0, blocks = (lm + R - 1) / R;
							#pragma omp parallel for num_threads(spmv::host_threads()) reduction(+ : over)
							for (long b = 0; b < blocks; b++)
								over += (rp[std::min(lm, (b + 1) * R)] - rp[b * R]) > cap;
							if (over * 200 <= blocks)
								break;
						}
					}
				}
				const bool okR = (A->stream_mode != 2) ? (R == 4 || R == 8 || R == 16 || R == 32 || R == 64)
				                                       : (R == 8 || R == 16 || R == 32 || R == 64);
				if (!okR)
				{
					set_error("csr_stream: rows per wavefront (lanes_per_row) must be %s (got %d)",
							A->stream_mode != 2 ? "4,8,16,32 or 64" : "8,16,32 or 64", R);
					rc = 1;
					break;
				}
				A->lanes_per_row = R;
				{
					const long rpt = A->stream_mode == 3 ? csr_stream_d_rows_per_tile(R) : csr_stream_rows_per_tile(R);
					A->cfg.map = xcd_map_balanced(rp, lm, rpt, resolve_remap(A->remap, lm / rpt));
				}
				snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_STREAM%s_r%d_%s",
						A->stream_mode == 3 ? "D" : A->stream_mode == 2 ? "T" : "", R, pf);
				snprintf(A->kernel_name, sizeof(A->kernel_name), A->stream_mode == 3 ? "csr_stream_d_kernel" :
						A->stream_mode == 2 ? "csr_stream_t_kernel" : "csr_stream_kernel");
			}
			else if (format == SPMV_MI355X_CSR_VECTOR)
			{
				int G = o.lanes_per_row;
				if (G == 0)
					G = pick_lanes_per_row(lm > 0 ? (double) lnnz / lm : 0);
				if (G != 2 && G != 4 && G != 8 && G != 16 && G != 32 && G != 64)
				{
					set_error("lanes_per_row must be 2,4,8,16,32 or 64 (got %d)", G);
					rc = 1;
					break;
				}
				A->lanes_per_row = G;
				int RPG = o.rows_per_group;
				if (RPG == 0)
				{
					// two rows of a lane group in flight: measured +10 % on the nlpkkt240 twin (2.90 -> 2.63 ms), +8 % on
					// scircuit as 16 lanes x 2 rows instead of 8 lanes x 1, neutral on cant / pwtk fp64; four rows cost occupancy
					if (G == 8 && o.lanes_per_row == 0)
						G = 16;
					RPG = (G == 16 || G == 32) ? 2 : 1;
					A->lanes_per_row = G;
				}
				if ((RPG != 1 && RPG != 2 && RPG != 4) || (RPG > 1 && G < 8))
				{
					set_error("rows_per_group must be 1, 2 or 4 (2 and 4 need lanes_per_row >= 8), got %d with %d lanes", RPG, G);
					rc = 1;
					break;
				}
				A->rows_per_group = RPG;
				const long rpt = csr_vector_rows_per_tile(G, RPG);
				A->cfg.map = xcd_map_balanced(rp, lm, rpt, resolve_remap(A->remap, lm / rpt));
				if (RPG > 1)
					snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_VECTOR_g%d_r%d_%s", G, RPG, pf);
				else
					snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_VECTOR_g%d_%s", G, pf);
				snprintf(A->kernel_name, sizeof(A->kernel_name), RPG > 1 ? "csr_vector_multi_kernel" : "csr_vector_kernel");
			}
			else
			{
				A->merge_ipt = o.merge_items;
				A->merge_tile = merge_tile_items(A->f32, A->merge_ipt);
				A->merge_ipt = A->merge_tile / 256;
				long total = lm + lnnz;
				A->merge_num_tiles = (int) ((total + A->merge_tile - 1) / A->merge_tile);
				rc = dev_alloc(&A->d_coords, 2 * ((size_t) A->merge_num_tiles + 1)) ||
				     dev_alloc(&A->d_carry_row, (size_t) A->merge_num_tiles) ||
				     dev_alloc_bytes(&A->d_carry_val, (size_t) A->merge_num_tiles * A->vbytes);
				if (rc)
					break;
				rc = launch_merge_search(A->d_row_ptr, (int) lm, (int) lnnz, A->merge_tile, A->merge_num_tiles, A->d_coords, nullptr);
				if (rc)
					break;
				if (hipDeviceSynchronize() != hipSuccess)
				{
					set_error("merge tile search failed");
					rc = 1;
					break;
				}
				A->cfg.map = xcd_map_uniform((unsigned) A->merge_num_tiles, resolve_remap(A->remap, 0));      // tiles hold equal work by construction
				A->mem_footprint += 2.0 * (A->merge_num_tiles + 1) * 4;
				// Pattern matrices (Matrix-Market `pattern`: every value is the dummy 1.0, matrix_market.c:308-317 — the
				// soc-LiveJournal1 configuration) carry no information in the value array: keep the constant, drop the stream.
				bool uniform = lnnz > 0;
				{
					long differs = 0;
					const double v0 = lnnz > 0 ? (A->f32 ? (double) (float) va[0] : va[0]) : 0.0;
					#pragma omp parallel for num_threads(spmv::host_threads()) reduction(+ : differs)
					for (long j = 0; j < lnnz; j++)
						differs += (A->f32 ? (double) (float) va[j] : va[j]) != v0;
					uniform = uniform && differs == 0 && v0 == v0;
					if (uniform)
					{
						(void) hipFree(A->d_val);
						A->d_val = nullptr;
						A->cfg.unit = 1;
						A->cfg.unit_value = v0;
						A->mem_footprint -= (double) lnnz * A->vbytes;
					}
				}
				if (uniform)
					snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_MERGE_i%d_unit_%s", A->merge_ipt, pf);
				else
				snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_MERGE_i%d_%s", A->merge_ipt, pf);
				snprintf(A->kernel_name, sizeof(A->kernel_name), "merge_kernel");
			}
			break;
		}
		case SPMV_MI355X_SELL_C_SIGMA:
		{
			// auto: one row per lane (C = 64, bit-exact) when there are enough slices to fill the chip several times over,
			// else 16-row slices with 4 lanes per row (4x the wavefronts, 1/4 of the dependent chain) — profiles/sweep_r01.md
			int C = o.sell_c ? o.sell_c : 64;
			if (C != 16 && C != 32 && C != 64)
			{
				set_error("sell_c must be 16, 32 or 64 (got %d)", C);
				rc = 1;
				break;
			}
			long sigma = o.sell_sigma ? o.sell_sigma : 16384;
			if (sigma < C || sigma % C)
			{
				set_error("sell_sigma (%ld) must be a positive multiple of sell_c (%d)", sigma, C);
				rc = 1;
				break;
			}
			A->sell_c = C;
			A->sell_sigma = sigma;
			A->sell_delta = (C == 64) && (o.sell_delta != 2);      // 0 = auto (on for 64-row slices), 1 = on, 2 = off
			A->convert_on_device = o.convert_on != 2 && !getenv("SPMV_MI355X_HOST_CONVERT");
			{
				// waves per slice: enough wavefronts to occupy 256 CUs several times over
				const long slices = (lm + 63) / 64;
				int S = o.sell_split ? o.sell_split : (slices >= 16384 ? 1 : slices >= 8192 ? 2 : 4);
				if (S != 1 && S != 2 && S != 4)
				{
					set_error("sell_split must be 1, 2 or 4 (got %d)", S);
					rc = 1;
					break;
				}
				A->sell_split = A->sell_delta ? S : 1;
			}
			if (o.sell_delta == 1 && C != 64)
			{
				set_error("sell_delta needs sell_c = 64 (one lane per row)");
				rc = 1;
				break;
			}
			rc = A->sell_delta ? build_sell_delta(A, rp, ci, va) : build_sell(A, rp, ci, va);
			if (A->sell_delta && A->sell_split > 1)
				snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELLD_%d_%ld_w%d_%s", C, sigma, A->sell_split, pf);
			else
				snprintf(A->format_name, sizeof(A->format_name), "MI355X_SELL%s_%d_%ld_%s", A->sell_delta ? "D" : "", C, sigma, pf);
			snprintf(A->kernel_name, sizeof(A->kernel_name), A->sell_delta ? "sell_delta_kernel" : "sell_kernel");
			break;
		}
		case SPMV_MI355X_COO:
		{
			if (o.col_blocks != 0)
			{
				rc = build_coo_blocked(A, rp, ci, va, o.col_blocks, pf);
				break;
			}
			rc = upload_ints(rp, (size_t) lm + 1, &A->d_row_ptr) || upload_ints(ci, (size_t) lnnz, &A->d_col) ||
			     upload_values(A, va, (size_t) lnnz, &A->d_val) || dev_alloc(&A->d_rowind, (size_t) lnnz);
			if (rc)
				break;
			rc = launch_expand_rows(A->d_row_ptr, (int) lm, A->d_rowind, nullptr);
			if (rc)
				break;
			if (hipDeviceSynchronize() != hipSuccess)
			{
				set_error("COO row expansion failed");
				rc = 1;
				break;
			}
			(void) hipFree(A->d_row_ptr);          // COO keeps (rowind, colind, val) only: mkl_coo.cpp:65
			A->d_row_ptr = nullptr;
			int per_wave = coo_wave_items(o.merge_items);
			A->coo_k = per_wave / WAVE;
			A->coo_num_waves = (int) ((lnnz + per_wave - 1) / per_wave);
			rc = dev_alloc(&A->d_carry_row, (size_t) A->coo_num_waves) ||
			     dev_alloc_bytes(&A->d_carry_val, (size_t) A->coo_num_waves * A->vbytes);
			A->cfg.map = xcd_map_uniform((unsigned) ((A->coo_num_waves + coo_waves_per_tile() - 1) / coo_waves_per_tile()), resolve_remap(A->remap, 0));
			A->mem_footprint = (double) lnnz * (A->vbytes + 8);
			snprintf(A->format_name, sizeof(A->format_name), "MI355X_COO_k%d_%s", A->coo_k, pf);
			snprintf(A->kernel_name, sizeof(A->kernel_name), "coo_kernel");
			break;
		}
	}
	if (rc)
	{
		free_all(A);
		delete A;
		return 1;
	}
	*out = A;
	return 0;
}

int
spmv_mi355x_destroy(spmv_mi355x_matrix * A)
{
	if (!A)
		return 0;
	(void) hipSetDevice(A->device);
	free_all(A);
	delete A;
	return 0;
}

const char * spmv_mi355x_format_name(const spmv_mi355x_matrix * A) { return A->format_name; }
double spmv_mi355x_mem_footprint(const spmv_mi355x_matrix * A) { return A->mem_footprint; }
double spmv_mi355x_csr_mem_footprint(const spmv_mi355x_matrix * A) { return A->csr_mem_footprint; }
long spmv_mi355x_rows(const spmv_mi355x_matrix * A) { return A->m; }
long spmv_mi355x_cols(const spmv_mi355x_matrix * A) { return A->n; }
long spmv_mi355x_nnz(const spmv_mi355x_matrix * A) { return A->nnz; }
int spmv_mi355x_precision(const spmv_mi355x_matrix * A) { return A->precision; }
int spmv_mi355x_device(const spmv_mi355x_matrix * A) { return A->device; }

int
spmv_mi355x_spmv_device_async(spmv_mi355x_matrix * A, const void * x, void * y, int beta, void * hip_stream)
{
	hipStream_t st = (hipStream_t) hip_stream;
	LaunchCfg cfg = A->cfg;
	cfg.beta = beta ? 1 : 0;
	long grid = 0;
	int rc = 1;
	switch (A->format)
	{
		case SPMV_MI355X_CSR_SCALAR:
			rc = launch_csr_scalar(A->f32, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid);
			break;
		case SPMV_MI355X_CSR_VECTOR:
			rc = launch_csr_vector(A->f32, A->lanes_per_row, A->rows_per_group, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid);
			break;
		case SPMV_MI355X_CSR_STREAM:
			rc = (A->stream_mode == 4)
			     ? launch_csr_window(A->f32, A->lanes_per_row, A->d_row_ptr, A->d_col16 ? (const void *) A->d_col16 : (const void *) A->d_col,
					A->d_col16 ? 1 : 0, A->d_val, x, y, A->d_win_row, A->d_win_lo, A->d_win_w,
					A->win_lds_bytes, cfg, st, &grid)
			     : (A->stream_mode == 3)
			     ? launch_csr_stream_d(A->f32, A->lanes_per_row, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid)
			     : (A->stream_mode == 2)
			     ? launch_csr_stream_t(A->f32, A->lanes_per_row, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid)
			     : launch_csr_stream(A->f32, A->lanes_per_row, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid);
			break;
		case SPMV_MI355X_CSR_MERGE:
			rc = launch_merge(A->f32, A->merge_ipt, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, (int) A->nnz,
					A->merge_num_tiles, A->d_coords, A->d_carry_row, A->d_carry_val, cfg, st, &grid);
			break;
		case SPMV_MI355X_SELL_C_SIGMA:
			rc = A->sell_delta
			     ? launch_sell_delta(A->f32, A->sell_split, A->d_sell_desc, A->d_sell_idx, A->d_val, A->d_row_of_sorted, x, y, (int) A->m,
					(int) A->sell_slices, cfg, st, &grid)
			     : launch_sell(A->f32, A->sell_c, A->d_slice_ptr, A->d_col, A->d_val, A->d_row_of_sorted, x, y, (int) A->m,
					(int) A->sell_slices, cfg, st, &grid);
			break;
		case SPMV_MI355X_COO:
			if (A->coob_segs > 0)
			{
				rc = launch_coo_blocked(A->f32, A->d_coob_seg_row, A->d_coob_seg_blk, A->d_col, A->d_coob_lrow, A->d_val, x, y, A->coob_segs,
						A->coob_blocks, A->coob_lds, cfg, st, &grid);
				break;
			}
			rc = launch_coo(A->f32, A->coo_k, A->d_rowind, A->d_col, A->d_val, x, y, (int) A->m, A->nnz, A->coo_num_waves,
					A->d_carry_row, A->d_carry_val, cfg, st, &grid);
			break;
		default:
			set_error("bad handle");
	}
	A->last_grid = grid;
	return rc;
}

int
spmv_mi355x_copy_device_async(void * dst, const void * src, long bytes, void * hip_stream)
{
	if (bytes < 0 || (bytes > 0 && (!dst || !src)))
	{
		set_error("copy_device_async: bad argument");
		return 1;
	}
	if (bytes)
		HIP_TRY(hipMemcpyAsync(dst, src, (size_t) bytes, hipMemcpyDeviceToDevice, (hipStream_t) hip_stream));
	return 0;
}

int
spmv_mi355x_time_device(spmv_mi355x_matrix * A, const void * x, void * y, int iters, void * hip_stream, double * ms_out)
{
	hipStream_t st = (hipStream_t) hip_stream;
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipEventRecord(e0, st));
	for (int i = 0; i < iters; i++)
		if (spmv_mi355x_spmv_device_async(A, x, y, 0, hip_stream))
			return 1;
	HIP_TRY(hipEventRecord(e1, st));
	HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	(void) hipEventDestroy(e0);
	(void) hipEventDestroy(e1);
	*ms_out = iters > 0 ? (double) ms / iters : 0;
	return 0;
}

int
spmv_mi355x_kernel_info(const spmv_mi355x_matrix * A, char * name_out, long name_n, long * grid_out, int * block_out)
{
	if (name_out && name_n > 0)
		snprintf(name_out, name_n, "%s", A->kernel_name);
	if (grid_out)
		*grid_out = A->last_grid;
	if (block_out)
		*block_out = 256;
	return 0;
}

static int
ensure_xy(spmv_mi355x_matrix * A)
{
	HIP_TRY(hipSetDevice(A->device));
	if (!A->stream)
		HIP_TRY(hipStreamCreate(&A->stream));
	if (!A->d_x && dev_alloc_bytes(&A->d_x, (size_t) std::max<long>(A->n, 1) * A->vbytes))
		return 1;
	if (!A->d_y && dev_alloc_bytes(&A->d_y, (size_t) (A->m + 64) * A->vbytes))
		return 1;
	return 0;
}

void * spmv_mi355x_x_device(spmv_mi355x_matrix * A) { return ensure_xy(A) ? nullptr : A->d_x; }
void * spmv_mi355x_y_device(spmv_mi355x_matrix * A) { return ensure_xy(A) ? nullptr : A->d_y; }

int
spmv_mi355x_set_always_copy(spmv_mi355x_matrix * A, int on)
{
	A->always_copy = on != 0;
	return 0;
}

int
spmv_mi355x_upload_x(spmv_mi355x_matrix * A, const void * x_host)
{
	if (ensure_xy(A))
		return 1;
	HIP_TRY(hipMemcpyAsync(A->d_x, x_host, (size_t) A->n * A->vbytes, hipMemcpyHostToDevice, A->stream));
	HIP_TRY(hipStreamSynchronize(A->stream));
	A->cached_x_host = x_host;
	return 0;
}

int
spmv_mi355x_download_y(spmv_mi355x_matrix * A, void * y_host)
{
	if (ensure_xy(A))
		return 1;
	HIP_TRY(hipMemcpyAsync(y_host, A->d_y, (size_t) A->m * A->vbytes, hipMemcpyDeviceToHost, A->stream));
	HIP_TRY(hipStreamSynchronize(A->stream));
	A->y_downloaded = true;
	return 0;
}

// Matrix_Format::spmv with the reference GPU backends' caching convention (csr_rocm_vector.cpp:224-257).
int
spmv_mi355x_spmv(spmv_mi355x_matrix * A, const void * x_host, void * y_host)
{
	if (ensure_xy(A))
		return 1;
	if (A->always_copy || A->cached_x_host != x_host)
		if (spmv_mi355x_upload_x(A, x_host))
			return 1;
	if (spmv_mi355x_spmv_device_async(A, A->d_x, A->d_y, 0, A->stream))
		return 1;
	HIP_TRY(hipStreamSynchronize(A->stream));
	if (A->always_copy || !A->y_downloaded)
		if (spmv_mi355x_download_y(A, y_host))
			return 1;
	return 0;
}

int
spmv_mi355x_sell_layout(const spmv_mi355x_matrix * A, long * C_out, long * sigma_out, long * num_slices_out,
		long * nnz_ext_out, int64_t ** slice_ptr_out, int32_t ** col_out, double ** val_out, int32_t ** row_of_sorted_out)
{
	if (A->format != SPMV_MI355X_SELL_C_SIGMA)
	{
		set_error("not a SELL handle");
		return 1;
	}
	HIP_TRY(hipSetDevice(A->device));
	if (C_out) *C_out = A->sell_c;
	if (sigma_out) *sigma_out = A->sell_sigma;
	if (num_slices_out) *num_slices_out = A->sell_slices;
	if (nnz_ext_out) *nnz_ext_out = A->sell_nnz_ext;
	std::vector<int64_t> h_desc;
	std::vector<unsigned char> h_idx;
	if (A->sell_delta)
	{
		h_desc.resize(2 * ((size_t) A->sell_slices + 1));
		HIP_TRY(hipMemcpy(h_desc.data(), A->d_sell_desc, h_desc.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
		h_idx.resize((size_t) std::max<long>(A->sell_idx_bytes, 1));
		HIP_TRY(hipMemcpy(h_idx.data(), A->d_sell_idx, (size_t) A->sell_idx_bytes, hipMemcpyDeviceToHost));
	}
	if (slice_ptr_out)
	{
		*slice_ptr_out = (int64_t *) malloc(((size_t) A->sell_slices + 1) * sizeof(int64_t));
		if (A->sell_delta)
			for (long sl = 0; sl <= A->sell_slices; sl++)
				(*slice_ptr_out)[sl] = h_desc[2 * sl];
		else
			HIP_TRY(hipMemcpy(*slice_ptr_out, A->d_slice_ptr, ((size_t) A->sell_slices + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
	}
	size_t ne = (size_t) std::max<long>(A->sell_nnz_ext, 1);
	if (col_out)
	{
		*col_out = (int32_t *) malloc(ne * sizeof(int32_t));
		if (A->sell_delta)
		{
			// decode the compressed indices back to the plain column-major layout
			for (long sl = 0; sl < A->sell_slices; sl++)
			{
				const int64_t vb = h_desc[2 * sl];
				const long width = (h_desc[2 * sl + 2] - vb) / 64;
				const int md = (int) (h_desc[2 * sl + 1] & 7);
				const unsigned char * ib = h_idx.data() + (h_desc[2 * sl + 1] & ~(int64_t) 15);
				const int * offs = reinterpret_cast<const int *>(ib);
				if (md == 3)
					ib += 4 * 64;
				const long gbytes = (md == 0 || md == 3) ? 16 : md == 1 ? 272 : md == 2 ? 528 : 1024;
				for (long k = 0; k < width; k++)
				{
					const unsigned char * gp = ib + (k / 4) * gbytes;
					const long u = k % 4;
					for (int r = 0; r < 64; r++)
					{
						int c;
						if (md == 0)
							c = reinterpret_cast<const int *>(gp)[u] + r;
						else if (md == 3)
							c = reinterpret_cast<const int *>(gp)[u] + offs[r];
						else if (md == 1)
							c = reinterpret_cast<const int *>(gp)[u] + gp[16 + r * 4 + u];
						else if (md == 2)
							c = reinterpret_cast<const int *>(gp)[u] + reinterpret_cast<const unsigned short *>(gp + 16)[r * 4 + u];
						else
							c = reinterpret_cast<const int *>(gp)[u * 64 + r];
						(*col_out)[vb + k * 64 + r] = c;
					}
				}
			}
		}
		else
			HIP_TRY(hipMemcpy(*col_out, A->d_col, (size_t) A->sell_nnz_ext * sizeof(int32_t), hipMemcpyDeviceToHost));
	}
	if (val_out)
	{
		*val_out = (double *) malloc(ne * sizeof(double));
		if (!A->f32)
			HIP_TRY(hipMemcpy(*val_out, A->d_val, (size_t) A->sell_nnz_ext * sizeof(double), hipMemcpyDeviceToHost));
		else
		{
			std::vector<float> tmp(ne);
			HIP_TRY(hipMemcpy(tmp.data(), A->d_val, (size_t) A->sell_nnz_ext * sizeof(float), hipMemcpyDeviceToHost));
			for (size_t i = 0; i < (size_t) A->sell_nnz_ext; i++)
				(*val_out)[i] = tmp[i];
		}
	}
	if (row_of_sorted_out)
	{
		*row_of_sorted_out = (int32_t *) malloc((size_t) std::max<long>(A->m, 1) * sizeof(int32_t));
		HIP_TRY(hipMemcpy(*row_of_sorted_out, A->d_row_of_sorted, (size_t) A->m * sizeof(int32_t), hipMemcpyDeviceToHost));
	}
	return 0;
}

int
spmv_mi355x_merge_tiles(const spmv_mi355x_matrix * A, long * num_tiles_out, long * tile_items_out, int32_t ** coords_out)
{
	if (A->format != SPMV_MI355X_CSR_MERGE)
	{
		set_error("not a merge handle");
		return 1;
	}
	HIP_TRY(hipSetDevice(A->device));
	if (num_tiles_out) *num_tiles_out = A->merge_num_tiles;
	if (tile_items_out) *tile_items_out = A->merge_tile;
	if (coords_out)
	{
		size_t cnt = 2 * ((size_t) A->merge_num_tiles + 1);
		*coords_out = (int32_t *) malloc(cnt * sizeof(int32_t));
		HIP_TRY(hipMemcpy(*coords_out, A->d_coords, cnt * sizeof(int32_t), hipMemcpyDeviceToHost));
	}
	return 0;
}

void
spmv_mi355x_free(void * p)
{
	free(p);
}

}  // extern "C"
