// CSR-Stream SpMV for gfx950: one wavefront per block of R consecutive rows, plain CSR arrays, no conversion.
//
// Role in the reference: the "row block" CSR kernels of the ROCm backends (CSR-Stream half of
// benchmark_code/BENCH/src/spmv_kernels/GPU_clean/spmv_subkernel_csr_rocm_adaptive.cpp:76-153 with the row blocks of
// csr_adaptive_cuda.cu:46-79); re-designed for wave64/LDS of CDNA4 rather than ported.
//
// Why: with one group of lanes per row (csr_vector) every row is a chain of three dependent memory round trips
// (row_ptr -> col/val -> x) with only a few hundred bytes in flight per wave; on 27 nnz/row matrices that is
// latency-bound at ~1/3 of the HBM rate. Here the R rows of a wave are ONE contiguous range of val/col (CSR is
// row-major), so the wave streams that range like a dense vector:
//   1. lanes 0..R load row_ptr[r0..r0+R] (one coalesced load), the range [j0,j1) is broadcast;
//   2. up to STEPS wave-steps of 64 consecutive (val, col) pairs are issued back to back (8+4 bytes per lane and
//      step, all in flight before the first use), then the dependent x gathers, products go to the wave's LDS strip;
//   3. 64/R lanes per row sum that row's products out of LDS (strided by 64/R, then a xor-butterfly) and the group's
//      first lane stores y. Wave-local: no workgroup barrier anywhere.
// Row blocks longer than CAP = 64*STEPS non-zeros (rare in the regular matrices this kernel is chosen for) fall back to a
// whole-wave loop per row. Every y[i] is written exactly once; empty rows get 0. Summation order differs from the
// sequential CPU loop (strided partial sums), so parity with the reference is to tolerance, reproducible run to run.

#include "launch.hpp"

namespace spmv {

constexpr int STREAM_BLOCK = 256;
constexpr int STREAM_WAVES = STREAM_BLOCK / WAVE;

template <typename T, int R, int STEPS, bool NT>
__global__ __launch_bounds__(STREAM_BLOCK) void
csr_stream_kernel(const int * __restrict__ row_ptr, const int * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, T * __restrict__ y, int m, int beta, XcdMap map)
{
	constexpr int CAP = WAVE * STEPS;
	constexpr int L = WAVE / R;              // lanes per row in the reduction
	__shared__ T s_prod[STREAM_WAVES][CAP];

	const unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int lane = threadIdx.x % WAVE;
	const int wave = threadIdx.x / WAVE;
	const long r0 = ((long) tile * STREAM_WAVES + wave) * R;
	if (r0 >= m)
		return;                               // whole wave leaves together
	const int rows = (m - r0 < R) ? (int) (m - r0) : R;

	// row_ptr[r0 .. r0+rows] in lanes 0..rows (R <= 64 so R+1 <= 65: the last boundary of a full block rides in lane 0's
	// second register)
	int rp = row_ptr[r0 + (lane <= rows ? lane : rows)];
	int rp_last = row_ptr[r0 + rows];
	const int j0 = __shfl(rp, 0, WAVE);
	const int j1 = rp_last;
	const int len = j1 - j0;
	T * __restrict__ prod = s_prod[wave];

	if (len <= CAP)
	{
		int c[STEPS];
		T v[STEPS];
		#pragma unroll
		for (int s = 0; s < STEPS; s++)
		{
			const int idx = s * WAVE + lane;
			const bool ok = idx < len;
			const long j = (long) j0 + (ok ? idx : 0);
			if (s * WAVE < len)              // wave-uniform: whole steps beyond the range are skipped
			{
				c[s] = ok ? ld_stream<NT>(col + j) : 0;
				v[s] = ok ? ld_stream<NT>(val + j) : T(0);
			}
			else
			{
				c[s] = 0;
				v[s] = T(0);
			}
		}
		#pragma unroll
		for (int s = 0; s < STEPS; s++)
			if (s * WAVE < len)
			{
				const int idx = s * WAVE + lane;
				if (idx < len)
					prod[idx] = v[s] * x[c[s]];
			}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		const int q = lane / L;              // row of this lane inside the block
		const int sub = lane % L;
		// boundaries of row q: lanes q and q+1 hold them (q+1 == R only for lane group R-1 -> rp_last)
		int b0 = __shfl(rp, q, WAVE);
		int b1 = __shfl(rp, (q + 1 < WAVE) ? q + 1 : q, WAVE);
		if (q + 1 >= rows)
			b1 = rp_last;
		if (q >= rows)
			b0 = b1;
		T sum = 0;
		for (int k = b0 - j0 + sub; k < b1 - j0; k += L)
			sum += prod[k];
		#pragma unroll
		for (int off = L / 2; off >= 1; off >>= 1)
			sum += shfl_xor_t(sum, off);
		if (sub == 0 && q < rows)
		{
			T * yp = y + (r0 + q);
			*yp = beta ? *yp + sum : sum;
		}
	}
	else
	{
		// long row block: whole wave per row, two accumulators, butterfly
		for (int q = 0; q < rows; q++)
		{
			const int b0 = __shfl(rp, q, WAVE);
			int b1 = __shfl(rp, (q + 1 < WAVE) ? q + 1 : q, WAVE);
			if (q + 1 >= rows)
				b1 = rp_last;
			T sum = 0, sum2 = 0;
			int j = b0 + lane;
			for (; j + WAVE < b1; j += 2 * WAVE)
			{
				const int c0 = ld_stream<NT>(col + j);
				const int c1 = ld_stream<NT>(col + j + WAVE);
				const T v0 = ld_stream<NT>(val + j);
				const T v1 = ld_stream<NT>(val + j + WAVE);
				sum = fma_t<T>(v0, x[c0], sum);
				sum2 = fma_t<T>(v1, x[c1], sum2);
			}
			if (j < b1)
				sum = fma_t<T>(ld_stream<NT>(val + j), x[ld_stream<NT>(col + j)], sum);
			sum = group_reduce_sum<T, WAVE>(sum + sum2);
			if (lane == 0)
			{
				T * yp = y + (r0 + q);
				*yp = beta ? *yp + sum : sum;
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------ launchers

template <typename T, int R, int STEPS>
static int
stream_launch(const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	if (cfg.nt)
		hipLaunchKernelGGL((csr_stream_kernel<T, R, STEPS, true>), dim3(grid), dim3(STREAM_BLOCK), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	else
		hipLaunchKernelGGL((csr_stream_kernel<T, R, STEPS, false>), dim3(grid), dim3(STREAM_BLOCK), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	HIP_TRY(hipGetLastError());
	return 0;
}

template <typename T>
static int
stream_dispatch(int R, const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	constexpr int STEPS = 12;
	switch (R)
	{
		case 4:  return stream_launch<T, 4, STEPS>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 8:  return stream_launch<T, 8, STEPS>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 16: return stream_launch<T, 16, STEPS>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 32: return stream_launch<T, 32, STEPS>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 64: return stream_launch<T, 64, STEPS>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
	}
	set_error("csr_stream: rows per wave must be 4,8,16,32 or 64 (got %d)", R);
	return 1;
}

int
csr_stream_cap()
{
	return WAVE * 12;
}

int
launch_csr_stream(bool f32, int rows_per_wave, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? stream_dispatch<float>(rows_per_wave, row_ptr, col, val, x, y, m, cfg, stream, grid_out)
	           : stream_dispatch<double>(rows_per_wave, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
}

}  // namespace spmv
