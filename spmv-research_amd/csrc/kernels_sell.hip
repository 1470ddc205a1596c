// SELL-C-sigma SpMV for gfx950.
//
// Replaces the reference's sliced-ELL kernels (benchmark_code/BENCH/src/spmv_kernels/sell_sorted.cpp:338-419 — C = SIMD
// width, hardware gather, y scattered through rev_permutation — and the BSC library driven by sell_c_s.cpp:124-131,
// sell-C-s/RISC-V/sellcs_mv_kernels_epi.c:178-257 — C = 256, sigma = 16384, descending radix sort per window).
//
// Layout (built in spmv_mi355x.cpp): rows are sorted by length (descending, stable) inside windows of sigma rows; a slice
// is C consecutive sorted rows, stored column-major and padded to the slice's longest row, so that the 64 lanes of ONE
// wavefront read 64 consecutive values / column indices per step:
//       element (row r of the slice, column k)  ->  slice_ptr[s] + k*C + r
// One wavefront owns one slice. With C = 64 a lane owns a row and walks it left to right with one FMA per element
// (padding multiplies 0 by a valid x entry), i.e. y is bit-identical to the sequential CSR row loop. With C = 32 / 16 the
// wave covers 2 / 4 consecutive columns per step (lane = (k % TPR)*C + r); the TPR partial sums of a row are combined by
// a fixed xor-butterfly. The slice width is padded to a multiple of TPR. Smaller C = more wavefronts and shorter
// dependent chains for small matrices, at the cost of the butterfly.
//
// y is scattered through row_of_sorted (the reference does the same: sell_sorted.cpp:392-395).

#include "launch.hpp"

namespace spmv {

constexpr int SELL_BLOCK = 256;
constexpr int SELL_WAVES = SELL_BLOCK / WAVE;

template <typename T, int C, bool NT>
__global__ __launch_bounds__(SELL_BLOCK) void
sell_kernel(const int64_t * __restrict__ slice_ptr, const int * __restrict__ col, const T * __restrict__ val,
		const int * __restrict__ row_of_sorted, const T * __restrict__ x, T * __restrict__ y,
		int m, int num_slices, int beta, XcdMap map)
{
	unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int lane = threadIdx.x % WAVE;
	const int slice = tile * SELL_WAVES + threadIdx.x / WAVE;
	if (slice >= num_slices)
		return;                       // whole wavefront leaves together: no shuffle after a partial exit
	const int64_t p_s = slice_ptr[slice];
	const int64_t p_e = slice_ptr[slice + 1];
	T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
	int64_t p = p_s + lane;
	// 4 wave-steps per trip: 4 value + 4 index loads in flight per lane before the first dependent x gather.
	// A lane's own partial sums s0..s3 belong to the same row only when C == 64 and are then added in column order
	// below; for the bit-exact C == 64 path a single accumulator chain is used instead.
	if constexpr (C == WAVE)
	{
		for (; p + 3 * WAVE < p_e; p += 4 * WAVE)
		{
			const int c0 = ld_stream<NT>(col + p);
			const int c1 = ld_stream<NT>(col + p + WAVE);
			const int c2 = ld_stream<NT>(col + p + 2 * WAVE);
			const int c3 = ld_stream<NT>(col + p + 3 * WAVE);
			const T v0 = ld_stream<NT>(val + p);
			const T v1 = ld_stream<NT>(val + p + WAVE);
			const T v2 = ld_stream<NT>(val + p + 2 * WAVE);
			const T v3 = ld_stream<NT>(val + p + 3 * WAVE);
			const T x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
			s0 = fma_t<T>(v0, x0, s0);
			s0 = fma_t<T>(v1, x1, s0);
			s0 = fma_t<T>(v2, x2, s0);
			s0 = fma_t<T>(v3, x3, s0);
		}
		if (p < p_e)
		{
			// 1..3 leftover steps as ONE masked batch (a serial tail would expose a full memory round trip per step)
			const bool k1 = p + WAVE < p_e, k2 = p + 2 * WAVE < p_e;
			const int c0 = ld_stream<NT>(col + p);
			const int c1 = k1 ? ld_stream<NT>(col + p + WAVE) : 0;
			const int c2 = k2 ? ld_stream<NT>(col + p + 2 * WAVE) : 0;
			const T v0 = ld_stream<NT>(val + p);
			const T v1 = k1 ? ld_stream<NT>(val + p + WAVE) : T(0);
			const T v2 = k2 ? ld_stream<NT>(val + p + 2 * WAVE) : T(0);
			const T x0 = x[c0];
			const T x1 = k1 ? x[c1] : T(0);
			const T x2 = k2 ? x[c2] : T(0);
			s0 = fma_t<T>(v0, x0, s0);
			if (k1) s0 = fma_t<T>(v1, x1, s0);
			if (k2) s0 = fma_t<T>(v2, x2, s0);
		}
	}
	else
	{
		for (; p + 3 * WAVE < p_e; p += 4 * WAVE)
		{
			const int c0 = ld_stream<NT>(col + p);
			const int c1 = ld_stream<NT>(col + p + WAVE);
			const int c2 = ld_stream<NT>(col + p + 2 * WAVE);
			const int c3 = ld_stream<NT>(col + p + 3 * WAVE);
			const T v0 = ld_stream<NT>(val + p);
			const T v1 = ld_stream<NT>(val + p + WAVE);
			const T v2 = ld_stream<NT>(val + p + 2 * WAVE);
			const T v3 = ld_stream<NT>(val + p + 3 * WAVE);
			s0 = fma_t<T>(v0, x[c0], s0);
			s1 = fma_t<T>(v1, x[c1], s1);
			s2 = fma_t<T>(v2, x[c2], s2);
			s3 = fma_t<T>(v3, x[c3], s3);
		}
		if (p < p_e)
		{
			const bool k1 = p + WAVE < p_e, k2 = p + 2 * WAVE < p_e;
			const int c0 = ld_stream<NT>(col + p);
			const int c1 = k1 ? ld_stream<NT>(col + p + WAVE) : 0;
			const int c2 = k2 ? ld_stream<NT>(col + p + 2 * WAVE) : 0;
			const T v0 = ld_stream<NT>(val + p);
			const T v1 = k1 ? ld_stream<NT>(val + p + WAVE) : T(0);
			const T v2 = k2 ? ld_stream<NT>(val + p + 2 * WAVE) : T(0);
			s0 = fma_t<T>(v0, x[c0], s0);
			if (k1) s1 = fma_t<T>(v1, x[c1], s1);
			if (k2) s2 = fma_t<T>(v2, x[c2], s2);
		}
		s0 = (s0 + s1) + (s2 + s3);
		// combine the TPR = 64/C column phases of each row: lanes r, r+C, r+2C, ...
		#pragma unroll
		for (int off = C; off < WAVE; off <<= 1)
			s0 += shfl_xor_t(s0, off);
	}
	if (lane < C)
	{
		const long sorted_row = (long) slice * C + lane;
		if (sorted_row < m)
		{
			T * yp = y + row_of_sorted[sorted_row];
			*yp = beta ? *yp + s0 : s0;
		}
	}
}

template <typename T, int C>
static int
sell_launch_c(const int64_t * slice_ptr, const int * col, const void * val, const int * row_of_sorted, const void * x, void * y,
		int m, int num_slices, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	if (cfg.nt)
		hipLaunchKernelGGL((sell_kernel<T, C, true>), dim3(grid), dim3(SELL_BLOCK), 0, stream, slice_ptr, col, (const T *) val,
				row_of_sorted, (const T *) x, (T *) y, m, num_slices, cfg.beta, cfg.map);
	else
		hipLaunchKernelGGL((sell_kernel<T, C, false>), dim3(grid), dim3(SELL_BLOCK), 0, stream, slice_ptr, col, (const T *) val,
				row_of_sorted, (const T *) x, (T *) y, m, num_slices, cfg.beta, cfg.map);
	HIP_TRY(hipGetLastError());
	return 0;
}

template <typename T>
static int
sell_dispatch(int C, const int64_t * slice_ptr, const int * col, const void * val, const int * row_of_sorted, const void * x, void * y,
		int m, int num_slices, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	switch (C)
	{
		case 16: return sell_launch_c<T, 16>(slice_ptr, col, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out);
		case 32: return sell_launch_c<T, 32>(slice_ptr, col, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out);
		case 64: return sell_launch_c<T, 64>(slice_ptr, col, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out);
	}
	set_error("sell: C must be 16, 32 or 64 (got %d)", C);
	return 1;
}

int
launch_sell(bool f32, int C, const int64_t * slice_ptr, const int * col, const void * val, const int * row_of_sorted,
		const void * x, void * y, int m, int num_slices, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? sell_dispatch<float>(C, slice_ptr, col, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out)
	           : sell_dispatch<double>(C, slice_ptr, col, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out);
}

}  // namespace spmv
