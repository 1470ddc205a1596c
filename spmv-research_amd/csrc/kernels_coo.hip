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

}  // namespace spmv

