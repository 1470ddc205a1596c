// Row-sorted COO SpMV for gfx950.
//
// In the reference the COO multiply is third-party (mkl_cspblas_?coogemv at benchmark_code/BENCH/src/spmv_kernels/
// mkl_coo.cpp:102-104, rocsparse_dcoomv at GPU_clean/rocsparse_coo.cpp:294); only the CSR -> COO row expansion
// (mkl_coo.cpp:79-90) is in-repo, and it yields entries sorted by row. This kernel relies on that order.
//
// Each wavefront owns 64*K consecutive entries. All K (row, col, val) triples per lane are loaded up front (coalesced
// 4+4+sizeof(V) byte streams), then for each of the K wave-steps:
//   - p = val * x[col]; lane 0 adds the partial sum carried from the previous step if it continues the same row;
//   - segmented inclusive scan by row over the wave (6 shuffle steps: rows are sorted, so "same row as lane-off"
//     means the whole span is one segment);
//   - a lane whose successor has a different row ends a segment and stores y[row] (plain store: a row that ends
//     inside the wave's chunk has all of its in-chunk entries in that sum).
// The segment still open at the end of the chunk goes to (carry_row, carry_val)[wave]; coo_fixup_kernel adds the carries
// in wave order. y is cleared first (rows without entries must read 0; a row ending exactly at a chunk end is written
// only by the fix-up) unless beta == 1. No atomics: results are reproducible run to run.

#include <algorithm>

#include "launch.hpp"

namespace spmv {

constexpr int COO_BLOCK = 256;
constexpr int COO_WAVES = COO_BLOCK / WAVE;

template <typename T, int K, bool NT>
__global__ __launch_bounds__(COO_BLOCK) void
coo_kernel(const int * __restrict__ rowind, const int * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, T * __restrict__ y, long nnz, int num_waves,
		int * __restrict__ carry_row, T * __restrict__ carry_val, int beta, XcdMap map)
{
	unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int lane = threadIdx.x % WAVE;
	const int w = tile * COO_WAVES + threadIdx.x / WAVE;
	if (w >= num_waves)
		return;
	const long start = (long) w * (WAVE * K);

	int r[K], c[K];
	T v[K];
	#pragma unroll
	for (int k = 0; k < K; k++)
	{
		long idx = start + (long) k * WAVE + lane;
		bool ok = idx < nnz;
		long j = ok ? idx : 0;
		r[k] = ok ? ld_stream<NT>(rowind + j) : -1;     // -1 = past the end (never equals a real row)
		c[k] = ok ? ld_stream<NT>(col + j) : 0;
		v[k] = ok ? ld_stream<NT>(val + j) : T(0);
	}
	T p[K];
	#pragma unroll
	for (int k = 0; k < K; k++)
		p[k] = (r[k] >= 0) ? v[k] * x[c[k]] : T(0);

	int open_row = -1;      // row of lane 63 after the previous step (wave-uniform)
	T open_val = 0;         // its inclusive segment sum
	#pragma unroll
	for (int k = 0; k < K; k++)
	{
		T s = p[k];
		const int row = r[k];
		if (lane == 0 && row >= 0 && row == open_row)
			s += open_val;
		#pragma unroll
		for (int off = 1; off < WAVE; off <<= 1)
		{
			T so = shfl_up_t(s, off);
			int ro = __shfl_up(row, off, WAVE);
			if (lane >= off && ro == row)
				s += so;
		}
		// successor row: the next lane, or for lane 63 the first lane of the next step (-2 after the last step)
		int next_row = __shfl_down(row, 1, WAVE);
		const int next_first = (k + 1 < K) ? __shfl(r[(k + 1 < K) ? k + 1 : k], 0, WAVE) : -2;
		if (lane == WAVE - 1)
			next_row = next_first;
		if (row >= 0 && next_row != row)
		{
			// a segment ends at this lane. If nothing valid follows inside the chunk it may continue in the next
			// wave's chunk: exactly one lane per wave takes this branch and hands the sum to the fix-up.
			const bool chunk_final = (next_row == -1) || (k + 1 == K && lane == WAVE - 1);
			if (chunk_final)
			{
				carry_row[w] = row;
				carry_val[w] = s;
			}
			else
			{
				T * yp = y + row;
				*yp = beta ? *yp + s : s;
			}
		}
		open_row = __shfl(row, WAVE - 1, WAVE);
		open_val = __shfl(s, WAVE - 1, WAVE);
	}
}

template <typename T>
__global__ __launch_bounds__(COO_BLOCK) void
coo_fixup_kernel(const int * __restrict__ carry_row, const T * __restrict__ carry_val, int ncarry, int m, T * __restrict__ y)
{
	int t = blockIdx.x * COO_BLOCK + threadIdx.x;
	if (t >= ncarry)
		return;
	int row = carry_row[t];
	if (row < 0 || row >= m)
		return;
	if (t > 0 && carry_row[t - 1] == row)
		return;
	T sum = carry_val[t];
	for (int u = t + 1; u < ncarry && carry_row[u] == row; u++)
		sum += carry_val[u];
	y[row] += sum;
}

// ------------------------------------------------------------------------------------------------ launchers

static int
coo_default_k(int items_per_lane)
{
	if (items_per_lane == 2 || items_per_lane == 4 || items_per_lane == 8)
		return items_per_lane;
	return 4;
}

int
coo_wave_items(int items_per_lane)
{
	return WAVE * coo_default_k(items_per_lane);
}

template <typename T, int K>
static int
coo_launch_k(const int * rowind, const int * col, const void * val, const void * x, void * y, int m, long nnz, int num_waves,
		int * carry_row, void * carry_val, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (!cfg.beta && m > 0)
		HIP_TRY(hipMemsetAsync(y, 0, (size_t) m * sizeof(T), stream));
	if (grid == 0)
		return 0;
	if (cfg.nt)
		hipLaunchKernelGGL((coo_kernel<T, K, true>), dim3(grid), dim3(COO_BLOCK), 0, stream, rowind, col, (const T *) val,
				(const T *) x, (T *) y, nnz, num_waves, carry_row, (T *) carry_val, cfg.beta, cfg.map);
	else
		hipLaunchKernelGGL((coo_kernel<T, K, false>), dim3(grid), dim3(COO_BLOCK), 0, stream, rowind, col, (const T *) val,
				(const T *) x, (T *) y, nnz, num_waves, carry_row, (T *) carry_val, cfg.beta, cfg.map);
	HIP_TRY(hipGetLastError());
	unsigned fgrid = (unsigned) ((num_waves + COO_BLOCK - 1) / COO_BLOCK);
	hipLaunchKernelGGL((coo_fixup_kernel<T>), dim3(fgrid), dim3(COO_BLOCK), 0, stream, carry_row, (const T *) carry_val,
			num_waves, m, (T *) y);
	HIP_TRY(hipGetLastError());
	return 0;
}

template <typename T>
static int
coo_dispatch(int K, const int * rowind, const int * col, const void * val, const void * x, void * y, int m, long nnz, int num_waves,
		int * carry_row, void * carry_val, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	switch (K)
	{
		case 2: return coo_launch_k<T, 2>(rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out);
		case 4: return coo_launch_k<T, 4>(rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out);
		case 8: return coo_launch_k<T, 8>(rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out);
	}
	set_error("coo: unsupported items per lane %d", K);
	return 1;
}

int
launch_coo(bool f32, int items_per_lane, const int * rowind, const int * col, const void * val, const void * x, void * y,
		int m, long nnz, int num_waves, int * carry_row, void * carry_val,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	int K = coo_default_k(items_per_lane);
	return f32 ? coo_dispatch<float>(K, rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out)
	           : coo_dispatch<double>(K, rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out);
}

// ------------------------------------------------------------------------------------------------ column-blocked COO
//
// For graph matrices (soc-LiveJournal1: 14 entries per row scattered over a 39 MB x) the row-sorted kernels sit on the
// fabric's random-sector rate: every gather misses the 4 MiB L2 of its XCD. Here the rows are cut into segments of at most
// COOB_ROWS rows (their y lives in LDS), 512 segments run at a time — all resident, two workgroups per CU — and every
// workgroup walks ITS entries column block by column block (blocks of ~384 KiB of x, a barrier of the workgroup after every block), adding v*x[col] into the LDS copy of y
// with LDS atomics. Segments hold equal numbers of entries, so the workgroups sweep the column blocks in step and the block
// of x being gathered stays in every XCD's L2. Entry = (column: 4 B, row inside the segment: 2 B, value unless all values
// are equal). The LDS atomics make the order of a row's additions run-dependent (last-bit differences; bar 1e-12 / 1e-5).
constexpr int COOB_THREADS = 512;
constexpr int COOB_U = 4;
constexpr int COOB_SYNC_EVERY = 1;

template <typename T, bool UNIT>
__global__ __launch_bounds__(COOB_THREADS) void
coo_blocked_kernel(const int * __restrict__ seg_row, const int * __restrict__ seg_blk, const int * __restrict__ col,
		const unsigned short * __restrict__ lrow, const T * __restrict__ val, const T * __restrict__ x, T * __restrict__ y,
		int seg_base, int num_segs, int num_blocks, T unit, int beta)
{
	extern __shared__ __align__(16) unsigned char coob_smem[];
	T * ys = reinterpret_cast<T *>(coob_smem);
	const int seg = seg_base + (int) blockIdx.x;
	if (seg >= num_segs)
		return;
	const int r0 = seg_row[seg], nrows = seg_row[seg + 1] - r0;
	for (int i = threadIdx.x; i < nrows; i += COOB_THREADS)
		ys[i] = 0;
	__syncthreads();
	// block by block: measured faster than one flat walk over the segment's entries with 8 in flight (561 vs 589 us on the
	// soc-LiveJournal1 twin) — the per-block loop is what keeps the workgroups on the same block of x
	const int * sb = seg_blk + (size_t) seg * (num_blocks + 1);
	for (int b = 0; b < num_blocks; b++)
	{
		const int e1 = sb[b + 1];
		for (int e = sb[b] + (int) threadIdx.x; e < e1; e += COOB_U * COOB_THREADS)
		{
			int c[COOB_U];
			unsigned short r[COOB_U];
			T v[COOB_U];
			#pragma unroll
			for (int u = 0; u < COOB_U; u++)
			{
				const int ee = e + u * COOB_THREADS;
				const bool ok = ee < e1;
				// the entry streams are read once: nontemporal, so that they do not push the x block out of L2
				c[u] = ok ? ld_stream<true>(col + ee) : -1;
				r[u] = ok ? ld_stream<true>(lrow + ee) : (unsigned short) 0;
				v[u] = UNIT ? unit : (ok ? ld_stream<true>(val + ee) : (T) 0);
			}
			T xv[COOB_U];
			#pragma unroll
			for (int u = 0; u < COOB_U; u++)
				xv[u] = c[u] >= 0 ? x[c[u]] : (T) 0;
			#pragma unroll
			for (int u = 0; u < COOB_U; u++)
				if (c[u] >= 0)
					unsafeAtomicAdd(&ys[r[u]], v[u] * xv[u]);
		}
		if (COOB_SYNC_EVERY > 0 && (b + 1) % COOB_SYNC_EVERY == 0)
			__syncthreads();                 // keep the waves of the workgroup on the same column block
	}
	__syncthreads();
	for (int i = threadIdx.x; i < nrows; i += COOB_THREADS)
		y[r0 + i] = beta ? y[r0 + i] + ys[i] : ys[i];
}

int
coo_blocked_rows_cap(bool f32)
{
	static const int rows = [] {
		const char * e = getenv("SPMV_MI355X_COOB_ROWS");             // experiments only
		return e && atoi(e) >= 256 && atoi(e) <= 16384 ? atoi(e) : 8192;
	}();
	return f32 ? 2 * rows : rows;      // default 64 KiB of LDS per workgroup: two workgroups per CU
}

int
coo_blocked_segments_per_launch()
{
	// as many as are resident together (160 KiB of LDS per CU, 256 CUs), so that everything launched sweeps in step
	const int per_cu = std::max(1, std::min(4, (160 * 1024) / (coo_blocked_rows_cap(false) * 8 + 1024)));
	return 256 * per_cu;
}

template <typename T>
static int
coo_blocked_launch(const int * seg_row, const int * seg_blk, const int * col, const unsigned short * lrow, const void * val,
		const void * x, void * y, int num_segs, int num_blocks, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	static int granted = 0;
	if (lds_bytes > granted)
	{
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&coo_blocked_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&coo_blocked_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		granted = lds_bytes;
	}
	const int per = coo_blocked_segments_per_launch();
	if (grid_out)
		*grid_out = std::min(num_segs, per);
	for (int base = 0; base < num_segs; base += per)
	{
		const unsigned grid = (unsigned) std::min(per, num_segs - base);
		if (cfg.unit)
			hipLaunchKernelGGL((coo_blocked_kernel<T, true>), dim3(grid), dim3(COOB_THREADS), lds_bytes, stream, seg_row, seg_blk, col, lrow,
					(const T *) nullptr, (const T *) x, (T *) y, base, num_segs, num_blocks, (T) cfg.unit_value, cfg.beta);
		else
			hipLaunchKernelGGL((coo_blocked_kernel<T, false>), dim3(grid), dim3(COOB_THREADS), lds_bytes, stream, seg_row, seg_blk, col, lrow,
					(const T *) val, (const T *) x, (T *) y, base, num_segs, num_blocks, (T) 0, cfg.beta);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

int
launch_coo_blocked(bool f32, const int * seg_row, const int * seg_blk, const int * col, const unsigned short * lrow, const void * val,
		const void * x, void * y, int num_segs, int num_blocks, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? coo_blocked_launch<float>(seg_row, seg_blk, col, lrow, val, x, y, num_segs, num_blocks, lds_bytes, cfg, stream, grid_out)
	           : coo_blocked_launch<double>(seg_row, seg_blk, col, lrow, val, x, y, num_segs, num_blocks, lds_bytes, cfg, stream, grid_out);
}

}  // namespace spmv

