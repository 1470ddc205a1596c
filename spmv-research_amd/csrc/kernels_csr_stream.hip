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

// Variant T ("transposed consumption"): the wave still streams its R rows' contiguous (val, col) range coalesced, but
// parks the PAIRS in LDS and lets L = 64/R lanes walk each row. Lanes that are neighbours in the wave then hold
// neighbouring ROWS at the same position k, so on banded / stencil matrices the x gather of one instruction hits a few
// consecutive cache lines (as in SELL-C-sigma) instead of ~25 scattered ones (measured: the row-major gather above
// doubles the L1->L2 request count on the nlpkkt240 twin). With R = 64 a lane owns a row and accumulates it left to
// right with one FMA per element: bit-identical to the sequential CPU loop, straight from CSR storage.
template <typename T, int R, int STEPS, bool NT>
__global__ __launch_bounds__(STREAM_BLOCK) void
csr_stream_t_kernel(const int * __restrict__ row_ptr, const int * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, T * __restrict__ y, int m, int beta, XcdMap map)
{
	constexpr int CAP = WAVE * STEPS;
	constexpr int L = WAVE / R;
	__shared__ T s_val[STREAM_WAVES][CAP];
	__shared__ int s_col[STREAM_WAVES][CAP];

	const unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int lane = threadIdx.x % WAVE;
	const int wave = threadIdx.x / WAVE;
	const long r0 = ((long) tile * STREAM_WAVES + wave) * R;
	if (r0 >= m)
		return;
	const int rows = (m - r0 < R) ? (int) (m - r0) : R;
	int rp = row_ptr[r0 + (lane <= rows ? lane : rows)];
	const int rp_last = row_ptr[r0 + rows];
	const int j0 = __shfl(rp, 0, WAVE);
	const int len = rp_last - j0;
	T * __restrict__ lv = s_val[wave];
	int * __restrict__ lc = s_col[wave];

	const int q = lane % R;                  // row of this lane: neighbouring lanes = neighbouring rows
	const int sub = lane / R;                // which of the L interleaved walkers of that row
	int b0 = __shfl(rp, q, WAVE);
	int b1 = __shfl(rp, (q + 1 < WAVE) ? q + 1 : q, WAVE);
	if (q + 1 >= rows)
		b1 = rp_last;
	if (q >= rows)
		b0 = b1;

	T sum = 0;
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
			if (s * WAVE < len)
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
				lv[idx] = v[s];              // entries past len hold (0, col 0): harmless if ever read
				lc[idx] = c[s];
			}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		// walk: 4 elements per trip so that 4 x gathers are in flight per lane
		int k = b0 - j0 + sub;
		const int ke = b1 - j0;
		for (; k + 3 * L < ke; k += 4 * L)
		{
			const int c0 = lc[k], c1 = lc[k + L], c2 = lc[k + 2 * L], c3 = lc[k + 3 * L];
			const T v0 = lv[k], v1 = lv[k + L], v2 = lv[k + 2 * L], v3 = lv[k + 3 * L];
			const T x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
			sum = fma_t<T>(v0, x0, sum);
			sum = fma_t<T>(v1, x1, sum);
			sum = fma_t<T>(v2, x2, sum);
			sum = fma_t<T>(v3, x3, sum);
		}
		if (k < ke)
		{
			const bool k1 = k + L < ke, k2 = k + 2 * L < ke;
			const int c0 = lc[k];
			const int c1 = k1 ? lc[k + L] : 0;
			const int c2 = k2 ? lc[k + 2 * L] : 0;
			const T v0 = lv[k];
			const T v1 = k1 ? lv[k + L] : T(0);
			const T v2 = k2 ? lv[k + 2 * L] : T(0);
			const T x0 = x[c0];
			const T x1 = k1 ? x[c1] : T(0);
			const T x2 = k2 ? x[c2] : T(0);
			sum = fma_t<T>(v0, x0, sum);
			if (k1) sum = fma_t<T>(v1, x1, sum);
			if (k2) sum = fma_t<T>(v2, x2, sum);
		}
	}
	else
	{
		// long row block: same walk straight from global memory (uncoalesced val/col, rare)
		for (int k = b0 + sub; k < b1; k += L)
			sum = fma_t<T>(ld_stream<NT>(val + k), x[ld_stream<NT>(col + k)], sum);
	}
	// combine the L walkers of a row: lanes q, q+R, q+2R, ...
	#pragma unroll
	for (int off = R; off < WAVE; off <<= 1)
		sum += shfl_xor_t(sum, off);
	if (sub == 0 && q < rows)
	{
		T * yp = y + (r0 + q);
		*yp = beta ? *yp + sum : sum;
	}
}

// Variant D: as T, but the (val, col) range is copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per
// wave-instruction, no VGPR destination, no ds_write) instead of being staged through registers. The register-staged T
// kernel issues ~385 VALU + ~320 SALU instructions per wave (addresses, predicates, 24 unrolled steps) and was measured
// issue-bound at 2 waves/SIMD (SIMDs 56 % busy issuing, removing either the loads or the gathers did not change its
// time); the DMA form needs 17 copy instructions for a 1312-entry block.
// The copy starts at j0 rounded down to a multiple of 4 entries so that every per-lane source address is 16-byte
// aligned; LDS indices are shifted by the 0..3 skipped entries. The last chunk may read up to 1 KiB past the range:
// the device arrays carry that much slack (spmv_mi355x.hip: STREAM_SLACK).
// WPB waves per workgroup: chosen so that a workgroup's LDS stays below 64 KiB (the LDS-DMA destination base travels in M0)
template <typename T, int R, int CAPQ, int WPB, bool NT>
__global__ __launch_bounds__(WPB * WAVE) void
csr_stream_d_kernel(const int * __restrict__ row_ptr, const int * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, T * __restrict__ y, int m, int beta, XcdMap map)
{
	constexpr int CAP = 256 * CAPQ;          // entries per wave strip (multiple of one col chunk)
	constexpr int L = WAVE / R;
	constexpr int VPER = 16 / (int) sizeof(T);               // values per lane and DMA instruction (2 doubles / 4 floats)
	constexpr int VCHUNK = WAVE * VPER;                      // values per DMA instruction
	__shared__ __attribute__((aligned(16))) T s_val[WPB][CAP];
	__shared__ __attribute__((aligned(16))) int s_col[WPB][CAP];

	const unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int lane = threadIdx.x % WAVE;
	const int wave = threadIdx.x / WAVE;
	const long r0 = ((long) tile * WPB + wave) * R;
	if (r0 >= m)
		return;
	const int rows = (m - r0 < R) ? (int) (m - r0) : R;
	int rp = row_ptr[r0 + (lane <= rows ? lane : rows)];
	const int rp_last = row_ptr[r0 + rows];
	const int j0 = __builtin_amdgcn_readfirstlane(rp);
	const int j0a = j0 & ~3;                 // 16-byte aligned start for both arrays
	const int shift = j0 - j0a;
	const int len = rp_last - j0a;           // entries to copy (including the skipped ones)
	T * lv = s_val[wave];
	int * lc = s_col[wave];

	const int q = lane % R;
	const int sub = lane / R;
	int b0 = __shfl(rp, q, WAVE);
	int b1 = __shfl(rp, (q + 1 < WAVE) ? q + 1 : q, WAVE);
	if (q + 1 >= rows)
		b1 = rp_last;
	if (q >= rows)
		b0 = b1;

	T sum = 0;
	if (len <= CAP)
	{
		typedef __attribute__((address_space(1))) const void gvoid;
		typedef __attribute__((address_space(3))) void lvoid;
		const int * csrc = col + j0a + lane * 4;
		const T * vsrc = val + j0a + lane * VPER;
		#pragma unroll 1
		for (int e = 0; e < len; e += 256)
			__builtin_amdgcn_global_load_lds((gvoid *) (csrc + e), (lvoid *) (lc + e), 16, 0, NT ? 2 : 0);
		#pragma unroll 1
		for (int e = 0; e < len; e += VCHUNK)
			__builtin_amdgcn_global_load_lds((gvoid *) (vsrc + e), (lvoid *) (lv + e), 16, 0, NT ? 2 : 0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_wave_barrier();

		int k = b0 - j0a + sub;
		const int ke = b1 - j0a;
		(void) shift;
		for (; k + 3 * L < ke; k += 4 * L)
		{
			const int c0 = lc[k], c1 = lc[k + L], c2 = lc[k + 2 * L], c3 = lc[k + 3 * L];
			const T v0 = lv[k], v1 = lv[k + L], v2 = lv[k + 2 * L], v3 = lv[k + 3 * L];
			const T x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
			sum = fma_t<T>(v0, x0, sum);
			sum = fma_t<T>(v1, x1, sum);
			sum = fma_t<T>(v2, x2, sum);
			sum = fma_t<T>(v3, x3, sum);
		}
		if (k < ke)
		{
			const bool k1 = k + L < ke, k2 = k + 2 * L < ke;
			const int c0 = lc[k];
			const int c1 = k1 ? lc[k + L] : 0;
			const int c2 = k2 ? lc[k + 2 * L] : 0;
			const T v0 = lv[k];
			const T v1 = k1 ? lv[k + L] : T(0);
			const T v2 = k2 ? lv[k + 2 * L] : T(0);
			const T x0 = x[c0];
			const T x1 = k1 ? x[c1] : T(0);
			const T x2 = k2 ? x[c2] : T(0);
			sum = fma_t<T>(v0, x0, sum);
			if (k1) sum = fma_t<T>(v1, x1, sum);
			if (k2) sum = fma_t<T>(v2, x2, sum);
		}
	}
	else
	{
		// row block too long for the LDS strip (a few very long rows): whole wave per row, coalesced, butterfly
		for (int q2 = 0; q2 < rows; q2++)
		{
			const int c0r = __shfl(rp, q2, WAVE);
			int c1r = __shfl(rp, (q2 + 1 < WAVE) ? q2 + 1 : q2, WAVE);
			if (q2 + 1 >= rows)
				c1r = rp_last;
			T s1 = 0, s2 = 0;
			int j = c0r + lane;
			for (; j + WAVE < c1r; j += 2 * WAVE)
			{
				const int ca = ld_stream<NT>(col + j);
				const int cb = ld_stream<NT>(col + j + WAVE);
				const T va = ld_stream<NT>(val + j);
				const T vb = ld_stream<NT>(val + j + WAVE);
				s1 = fma_t<T>(va, x[ca], s1);
				s2 = fma_t<T>(vb, x[cb], s2);
			}
			if (j < c1r)
				s1 = fma_t<T>(ld_stream<NT>(val + j), x[ld_stream<NT>(col + j)], s1);
			s1 = group_reduce_sum<T, WAVE>(s1 + s2);
			if (lane == 0)
			{
				T * yp = y + (r0 + q2);
				*yp = beta ? *yp + s1 : s1;
			}
		}
		return;
	}
	#pragma unroll
	for (int off = R; off < WAVE; off <<= 1)
		sum += shfl_xor_t(sum, off);
	if (sub == 0 && q < rows)
	{
		T * yp = y + (r0 + q);
		*yp = beta ? *yp + sum : sum;
	}
}

// ------------------------------------------------------------------------------------------------ launchers

template <typename T, int R, int STEPS>
static int
stream_t_launch(const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	if (cfg.nt)
		hipLaunchKernelGGL((csr_stream_t_kernel<T, R, STEPS, true>), dim3(grid), dim3(STREAM_BLOCK), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	else
		hipLaunchKernelGGL((csr_stream_t_kernel<T, R, STEPS, false>), dim3(grid), dim3(STREAM_BLOCK), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	HIP_TRY(hipGetLastError());
	return 0;
}

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

template <typename T, int R, int CAPQ, int WPB>
static int
stream_d_launch(const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	if (cfg.nt)
		hipLaunchKernelGGL((csr_stream_d_kernel<T, R, CAPQ, WPB, true>), dim3(grid), dim3(WPB * WAVE), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	else
		hipLaunchKernelGGL((csr_stream_d_kernel<T, R, CAPQ, WPB, false>), dim3(grid), dim3(WPB * WAVE), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	HIP_TRY(hipGetLastError());
	return 0;
}

// LDS-DMA variant: strip = 256*CAPQ entries per wave
template <typename T>
static int
stream_d_dispatch(int R, const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	switch (R)
	{
		// one strip size for every R: 512 entries = 6 KiB per wave (fp64), 24 KiB per 4-wave workgroup -> 6 workgroups per CU
		case 4:  return stream_d_launch<T, 4, 2, 4>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 8:  return stream_d_launch<T, 8, 2, 4>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 16: return stream_d_launch<T, 16, 2, 4>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 32: return stream_d_launch<T, 32, 2, 4>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 64: return stream_d_launch<T, 64, 2, 4>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
	}
	set_error("csr_stream (LDS-DMA): rows per wave must be 4,8,16,32 or 64 (got %d)", R);
	return 1;
}

long
csr_stream_d_rows_per_tile(int rows_per_wave)
{
	return (long) rows_per_wave * 4;
}

int
csr_stream_d_cap(int rows_per_wave)
{
	(void) rows_per_wave;
	return 256 * 2 - 4;
}

int
launch_csr_stream_d(bool f32, int rows_per_wave, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? stream_d_dispatch<float>(rows_per_wave, row_ptr, col, val, x, y, m, cfg, stream, grid_out)
	           : stream_d_dispatch<double>(rows_per_wave, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
}

// transposed variant: LDS holds (val, col) pairs, so the strip is sized per R (rows per wave) for ~48 nnz/row
template <typename T>
static int
stream_t_dispatch(int R, const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	switch (R)
	{
		case 8:  return stream_t_launch<T, 8, 8>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 16: return stream_t_launch<T, 16, 12>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 32: return stream_t_launch<T, 32, 24>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 64: return stream_t_launch<T, 64, 24>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
	}
	set_error("csr_stream (transposed): rows per wave must be 8,16,32 or 64 (got %d)", R);
	return 1;
}

int
csr_stream_t_cap(int rows_per_wave)
{
	return WAVE * (rows_per_wave <= 8 ? 8 : rows_per_wave <= 16 ? 12 : 24);
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
launch_csr_stream_t(bool f32, int rows_per_wave, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? stream_t_dispatch<float>(rows_per_wave, row_ptr, col, val, x, y, m, cfg, stream, grid_out)
	           : stream_t_dispatch<double>(rows_per_wave, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
}

int
launch_csr_stream(bool f32, int rows_per_wave, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? stream_dispatch<float>(rows_per_wave, row_ptr, col, val, x, y, m, cfg, stream, grid_out)
	           : stream_dispatch<double>(rows_per_wave, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
}

}  // namespace spmv
