// CSR y = A*x kernels for gfx950: one lane per row (scalar) and one group of G lanes per row (vector).
//
// What they replace in the reference (paths under benchmark_code/BENCH/src/spmv_kernels/):
//   csr_scalar  : the CPU inner loop subkernel_csr_scalar (csr.cpp:334-350). One lane walks its row left to right
//                 with one FMA per non-zero, so y is BIT-IDENTICAL to the reference CPU kernel (used as the on-device
//                 reference for the reordering kernels).
//   csr_vector  : "one wavefront per row" (GPU_clean/spmv_subkernel_csr_rocm_vector.cpp:5-54; CPU analogue
//                 csr_vec.cpp:182-213), generalised to G = 2..64 lanes per row so short rows do not idle a wave64.
//                 Lane l of a group accumulates elements l, l+G, ... (coalesced 8/4-byte streams of val/col) in four
//                 interleaved partial sums, then a fixed xor-butterfly over the group gives the row sum: no LDS, no
//                 atomics, every y[i] written once, same bits from run to run.
//
// HBM-bound: per non-zero sizeof(V)+4 bytes streamed, x gathered through L2 / Infinity Cache.

#include "launch.hpp"

namespace spmv {

constexpr int CSR_BLOCK = 256;

// KAHAN: the reference's compensated variant (csr.cpp:353-373, -DCUSTOM_KAHAN): per element val = a*x - compensation (one FMA in
// the reference build), tmp = sum + val, compensation = (tmp - sum) - val — the same operations in the same order, so y is
// bit-identical to that build as well.
template <typename T, bool NT, bool KAHAN>
__global__ __launch_bounds__(CSR_BLOCK) void
csr_scalar_kernel(const int * __restrict__ row_ptr, const int * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, T * __restrict__ y, int m, int beta, XcdMap map)
{
	unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	int row = tile * CSR_BLOCK + threadIdx.x;
	if (row >= m)
		return;
	int j = row_ptr[row];
	int j_e = row_ptr[row + 1];
	T sum = 0;
	if constexpr (KAHAN)
	{
		T compensation = 0;
		for (; j < j_e; j++)
		{
			const T v = fma_t<T>(ld_stream<NT>(val + j), x[ld_stream<NT>(col + j)], -compensation);
			const T tmp = sum + v;
			compensation = (tmp - sum) - v;
			sum = tmp;
		}
	}
	else
	{
		for (; j < j_e; j++)
			sum = fma_t<T>(ld_stream<NT>(val + j), x[ld_stream<NT>(col + j)], sum);
	}
	y[row] = beta ? y[row] + sum : sum;
}

// Software-pipelined: a lane owns elements lane, lane+G, ... of its row and handles them in batches of U. The (col, val)
// loads of batch b+1 are issued BEFORE the x gathers of batch b are consumed, so a row costs about one memory round trip
// per batch instead of two (index -> gather are dependent). On the cache-resident matrices (cant: 62 k rows x 64) the
// whole grid is a single wave of workgroups and the kernel time IS that dependent-latency chain.
template <typename T, int G, bool NT>
__global__ __launch_bounds__(CSR_BLOCK) void
csr_vector_kernel(const int * __restrict__ row_ptr, const int * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, T * __restrict__ y, int m, int beta, XcdMap map)
{
	constexpr int ROWS_PER_BLOCK = CSR_BLOCK / G;
	constexpr int U = 4;
	unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int row = tile * ROWS_PER_BLOCK + threadIdx.x / G;
	const int lane = threadIdx.x % G;
	T acc[U];
	#pragma unroll
	for (int u = 0; u < U; u++)
		acc[u] = 0;
	if (row < m)
	{
		const int j_s = row_ptr[row];
		const int j_e = row_ptr[row + 1];
		int j = j_s + lane;
		int c[U];
		T v[U];
		#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const bool ok = j + u * G < j_e;
			c[u] = ok ? ld_stream<NT>(col + j + u * G) : -1;
			v[u] = ok ? ld_stream<NT>(val + j + u * G) : (T) 0;
		}
		while (j < j_e)
		{
			T xv[U];
			#pragma unroll
			for (int u = 0; u < U; u++)
				xv[u] = c[u] >= 0 ? x[c[u]] : (T) 0;
			j += U * G;
			int cn[U];
			T vn[U];
			#pragma unroll
			for (int u = 0; u < U; u++)
			{
				const bool ok = j + u * G < j_e;
				cn[u] = ok ? ld_stream<NT>(col + j + u * G) : -1;
				vn[u] = ok ? ld_stream<NT>(val + j + u * G) : (T) 0;
			}
			#pragma unroll
			for (int u = 0; u < U; u++)
			{
				acc[u] = c[u] >= 0 ? fma_t<T>(v[u], xv[u], acc[u]) : acc[u];
				c[u] = cn[u];
				v[u] = vn[u];
			}
		}
	}
	T sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
	sum = group_reduce_sum<T, G>(sum);
	if (row < m && lane == 0)
		y[row] = beta ? y[row] + sum : sum;
}

// Several rows per lane group IN FLIGHT. A group of G lanes owns RPG consecutive rows and issues the (col, val) loads of a
// batch of U elements per lane for ALL of them before the first x gather: the three dependent memory round trips of a row
// (row_ptr -> indices/values -> x) are paid once per RPG rows instead of once per row. That is what bounds the mid-size
// matrices: pwtk (218 k rows x 53) ran at 24 us in fp64 AND fp32 with one row per group — wavefront turnaround, not bytes.
// Per row a lane keeps one accumulator and adds its elements in index order; the butterfly then sums the lanes.
template <typename T, int G, int RPG, bool NT>
__global__ __launch_bounds__(CSR_BLOCK) void
csr_vector_multi_kernel(const int * __restrict__ row_ptr, const int * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, T * __restrict__ y, int m, int beta, XcdMap map)
{
	constexpr int ROWS_PER_BLOCK = CSR_BLOCK / G * RPG;
	constexpr int U = 4;
	static_assert(RPG + 1 <= G, "one lane per row pointer");
	unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int row0 = tile * ROWS_PER_BLOCK + (threadIdx.x / G) * RPG;
	const int lane = threadIdx.x % G;
	// lanes 0..RPG of the group fetch the RPG+1 row pointers with one load; rows past m read row_ptr[m] twice = empty
	const int rp_mine = row_ptr[min(row0 + min(lane, RPG), m)];
	int j[RPG], je[RPG];
	T acc[RPG];
	bool any = false;
	#pragma unroll
	for (int r = 0; r < RPG; r++)
	{
		j[r] = __shfl(rp_mine, r, G) + lane;
		je[r] = __shfl(rp_mine, r + 1, G);
		acc[r] = 0;
		any |= j[r] < je[r];
	}
	while (any)
	{
		int c[RPG][U];
		T v[RPG][U];
		#pragma unroll
		for (int r = 0; r < RPG; r++)
			#pragma unroll
			for (int u = 0; u < U; u++)
			{
				const bool ok = j[r] + u * G < je[r];
				c[r][u] = ok ? ld_stream<NT>(col + j[r] + u * G) : -1;
				v[r][u] = ok ? ld_stream<NT>(val + j[r] + u * G) : (T) 0;
			}
		T xv[RPG][U];
		#pragma unroll
		for (int r = 0; r < RPG; r++)
			#pragma unroll
			for (int u = 0; u < U; u++)
				xv[r][u] = c[r][u] >= 0 ? x[c[r][u]] : (T) 0;
		any = false;
		#pragma unroll
		for (int r = 0; r < RPG; r++)
		{
			#pragma unroll
			for (int u = 0; u < U; u++)
				acc[r] = c[r][u] >= 0 ? fma_t<T>(v[r][u], xv[r][u], acc[r]) : acc[r];
			j[r] += U * G;
			any |= j[r] < je[r];
		}
	}
	T mine = 0;                       // lane r of the group ends up holding row r's total and writes it: one coalesced store
	#pragma unroll
	for (int r = 0; r < RPG; r++)
	{
		const T total = group_reduce_sum<T, G>(acc[r]);
		mine = lane == r ? total : mine;
	}
	if (lane < RPG && row0 + lane < m)
		y[row0 + lane] = beta ? y[row0 + lane] + mine : mine;
}

// CSR -> COO row expansion on the device (the reference does it on the host: mkl_coo.cpp:79-90).
__global__ __launch_bounds__(CSR_BLOCK) void
expand_rows_kernel(const int * __restrict__ row_ptr, int m, int * __restrict__ rowind)
{
	// one group of 8 lanes per row: short rows dominate
	constexpr int G = 8;
	long row = ((long) blockIdx.x * CSR_BLOCK + threadIdx.x) / G;
	int lane = threadIdx.x % G;
	if (row >= m)
		return;
	int j_e = row_ptr[row + 1];
	for (int j = row_ptr[row] + lane; j < j_e; j += G)
		rowind[j] = (int) row;
}

// ------------------------------------------------------------------------------------------------ launchers

template <typename T>
static int
csr_scalar_dispatch(const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	#define CSR_SCALAR_LAUNCH(NTV, K) hipLaunchKernelGGL((csr_scalar_kernel<T, NTV, K>), dim3(grid), dim3(CSR_BLOCK), 0, stream, row_ptr, col, \
			(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map)
	if (cfg.kahan)
	{
		if (cfg.nt) CSR_SCALAR_LAUNCH(true, true);
		else        CSR_SCALAR_LAUNCH(false, true);
	}
	else
	{
		if (cfg.nt) CSR_SCALAR_LAUNCH(true, false);
		else        CSR_SCALAR_LAUNCH(false, false);
	}
	#undef CSR_SCALAR_LAUNCH
	HIP_TRY(hipGetLastError());
	return 0;
}

int
launch_csr_scalar(bool f32, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? csr_scalar_dispatch<float>(row_ptr, col, val, x, y, m, cfg, stream, grid_out)
	           : csr_scalar_dispatch<double>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
}

template <typename T, int G, int RPG>
static int
csr_vector_multi_launch(const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	if (cfg.nt)
		hipLaunchKernelGGL((csr_vector_multi_kernel<T, G, RPG, true>), dim3(grid), dim3(CSR_BLOCK), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	else
		hipLaunchKernelGGL((csr_vector_multi_kernel<T, G, RPG, false>), dim3(grid), dim3(CSR_BLOCK), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	HIP_TRY(hipGetLastError());
	return 0;
}

template <typename T, int G>
static int
csr_vector_launch_g(int rows_per_group, const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	if constexpr (G >= 8)
	{
		if (rows_per_group == 2)
			return csr_vector_multi_launch<T, G, 2>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		if (rows_per_group == 4)
			return csr_vector_multi_launch<T, G, 4>(row_ptr, col, val, x, y, m, cfg, stream, grid_out);
	}
	if (rows_per_group != 1)
	{
		set_error("csr_vector: rows_per_group must be 1, 2 or 4 (2 and 4 need lanes_per_row >= 8), got %d", rows_per_group);
		return 1;
	}
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	if (cfg.nt)
		hipLaunchKernelGGL((csr_vector_kernel<T, G, true>), dim3(grid), dim3(CSR_BLOCK), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	else
		hipLaunchKernelGGL((csr_vector_kernel<T, G, false>), dim3(grid), dim3(CSR_BLOCK), 0, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, m, cfg.beta, cfg.map);
	HIP_TRY(hipGetLastError());
	return 0;
}

template <typename T>
static int
csr_vector_dispatch(int G, int rpg, const int * row_ptr, const int * col, const void * val, const void * x, void * y, int m,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	switch (G)
	{
		case 2:  return csr_vector_launch_g<T, 2>(rpg, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 4:  return csr_vector_launch_g<T, 4>(rpg, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 8:  return csr_vector_launch_g<T, 8>(rpg, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 16: return csr_vector_launch_g<T, 16>(rpg, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 32: return csr_vector_launch_g<T, 32>(rpg, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
		case 64: return csr_vector_launch_g<T, 64>(rpg, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
	}
	set_error("csr_vector: lanes_per_row must be 2,4,8,16,32 or 64 (got %d)", G);
	return 1;
}

int
launch_csr_vector(bool f32, int lanes_per_row, int rows_per_group, const int * row_ptr, const int * col, const void * val,
		const void * x, void * y, int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? csr_vector_dispatch<float>(lanes_per_row, rows_per_group, row_ptr, col, val, x, y, m, cfg, stream, grid_out)
	           : csr_vector_dispatch<double>(lanes_per_row, rows_per_group, row_ptr, col, val, x, y, m, cfg, stream, grid_out);
}

int
launch_expand_rows(const int * row_ptr, int m, int * rowind, hipStream_t stream)
{
	if (m <= 0)
		return 0;
	long threads = (long) m * 8;
	unsigned grid = (unsigned) ((threads + CSR_BLOCK - 1) / CSR_BLOCK);
	hipLaunchKernelGGL(expand_rows_kernel, dim3(grid), dim3(CSR_BLOCK), 0, stream, row_ptr, m, rowind);
	HIP_TRY(hipGetLastError());
	return 0;
}

}  // namespace spmv
