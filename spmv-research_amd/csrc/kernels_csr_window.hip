// CSR with the block's window of x held in LDS (CSR_STREAM, stream_mode 4).
//
// Why: on FEM-like matrices the x gathers — not the matrix stream — bound the CSR kernels. Measured on the pwtk twin
// (csr_vector, rocprofv3 --pmc): the texture addressers are busy 76 % of the kernel and the vector L1 serves 13 M 64-byte
// accesses for 11.6 M non-zeros, i.e. nearly every gathered x element costs its own L1 access (fp32 and fp64 take the
// same 24 us). But a block of consecutive rows of such a matrix only touches a narrow window of columns (cant twin:
// 256 rows -> 770 columns; pwtk twin: ~13 000 columns for any block, the band). MI355X has 160 KiB of LDS per CU, so:
//   * the host cuts the rows into nnz-balanced blocks (a multiple of the 256 CUs) and records each block's column
//     window [lo, lo+w); a block whose window exceeds the LDS budget keeps gathering from global memory;
//   * a workgroup of 1024 lanes copies x[lo .. lo+w) into LDS with coalesced loads (w/8 L1 accesses instead of one per
//     non-zero) and then gathers from LDS;
//   * a group of G lanes walks the rows r0+g, r0+g+NG, ... of the block with a two-deep software pipeline ACROSS rows:
//     the row pointers of the row after next and the first (col, val) batch of the next row are in flight while the
//     current row is consumed, so the row_ptr -> indices -> x dependency chain is not paid per row.
// Reference point: the reference's CSR-Adaptive kernel stages the row block's PRODUCTS in LDS
// (GPU_clean/spmv_subkernel_csr_rocm_adaptive.cpp:76-153); staging x instead needs the large LDS of this chip.
// Per row a lane sums its elements in index order into U interleaved accumulators, then the xor-butterfly of the group:
// reordering kernel (tolerance parity), deterministic from run to run.

#include "launch.hpp"

namespace spmv {

constexpr int WIN_BLOCK = 1024;

template <typename T, typename CI, int G, bool NT, bool LDSX>
__device__ __forceinline__ void
window_rows(const int * __restrict__ row_ptr, const CI * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, const T * __restrict__ xs, T * __restrict__ y, int r0, int r1, int lo, int beta)
{
	constexpr int U = 4;
	constexpr int NG = WIN_BLOCK / G;
	const int group = threadIdx.x / G;
	const int lane = threadIdx.x % G;
	int row = r0 + group;
	int nrow = row + NG;
	int js0 = 0, je0 = 0, js1 = 0, je1 = 0;
	if (row < r1)
	{
		js0 = row_ptr[row];
		je0 = row_ptr[row + 1];
	}
	if (nrow < r1)
	{
		js1 = row_ptr[nrow];
		je1 = row_ptr[nrow + 1];
	}
	int c[U];
	T v[U];
	#pragma unroll
	for (int u = 0; u < U; u++)
	{
		const int j = js0 + lane + u * G;
		const bool ok = j < je0;
		c[u] = ok ? (int) ld_stream<NT>(col + j) : -1;
		v[u] = ok ? ld_stream<NT>(val + j) : (T) 0;
	}
	while (row < r1)
	{
		// in flight while this row is consumed: row pointers of the row after next, first batch of the next row
		const int nnrow = nrow + NG;
		int js2 = 0, je2 = 0;
		if (nnrow < r1)
		{
			js2 = row_ptr[nnrow];
			je2 = row_ptr[nnrow + 1];
		}
		int cn0[U];
		T vn0[U];
		#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int j = js1 + lane + u * G;
			const bool ok = j < je1;
			cn0[u] = ok ? (int) ld_stream<NT>(col + j) : -1;
			vn0[u] = ok ? ld_stream<NT>(val + j) : (T) 0;
		}
		T acc[U];
		#pragma unroll
		for (int u = 0; u < U; u++)
			acc[u] = 0;
		int jb = js0;                              // first element of the current batch (uniform over the group)
		while (true)
		{
			T xv[U];
			#pragma unroll
			for (int u = 0; u < U; u++)
			{
				if (LDSX)
					xv[u] = c[u] >= 0 ? xs[c[u] - lo] : (T) 0;
				else
					xv[u] = c[u] >= 0 ? x[c[u]] : (T) 0;
			}
			jb += U * G;
			const bool more = jb < je0;
			int cn[U];
			T vn[U];
			if (more)
			{
				#pragma unroll
				for (int u = 0; u < U; u++)
				{
					const int j = jb + lane + u * G;
					const bool ok = j < je0;
					cn[u] = ok ? (int) ld_stream<NT>(col + j) : -1;
					vn[u] = ok ? ld_stream<NT>(val + j) : (T) 0;
				}
			}
			#pragma unroll
			for (int u = 0; u < U; u++)
				acc[u] = c[u] >= 0 ? fma_t<T>(v[u], xv[u], acc[u]) : acc[u];
			if (!more)
				break;
			#pragma unroll
			for (int u = 0; u < U; u++)
			{
				c[u] = cn[u];
				v[u] = vn[u];
			}
		}
		T sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
		sum = group_reduce_sum<T, G>(sum);
		if (lane == 0)
			y[row] = beta ? y[row] + sum : sum;
		row = nrow;
		nrow = nnrow;
		js0 = js1;
		je0 = je1;
		js1 = js2;
		je1 = je2;
		#pragma unroll
		for (int u = 0; u < U; u++)
		{
			c[u] = cn0[u];
			v[u] = vn0[u];
		}
	}
}


// RPG consecutive rows of a lane group in flight together (as csr_vector_multi_kernel), x from the LDS window. The row
// pointers of the group's NEXT set of rows are loaded before the current set is consumed, so a set costs one exposed
// global round trip (its index/value batches); the two-deep single-row pipeline of window_rows() exposes one per row.
template <typename T, typename CI, int G, int RPG, bool NT, bool LDSX>
__device__ __forceinline__ void
window_rows_multi(const int * __restrict__ row_ptr, const CI * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, const T * __restrict__ xs, T * __restrict__ y, int r0, int r1, int lo, int beta)
{
	constexpr int U = 4;
	constexpr int NG = WIN_BLOCK / G;
	static_assert(RPG + 1 <= G, "one lane per row pointer");
	const int group = threadIdx.x / G;
	const int lane = threadIdx.x % G;
	int base = r0 + group * RPG;
	int rp_mine = row_ptr[min(base + min(lane, RPG), r1)];
	while (base < r1)
	{
		const int nbase = base + NG * RPG;
		const int rp_next = row_ptr[min(nbase + min(lane, RPG), r1)];       // in flight while this set is consumed
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
					c[r][u] = ok ? (int) ld_stream<NT>(col + j[r] + u * G) : -1;
					v[r][u] = ok ? ld_stream<NT>(val + j[r] + u * G) : (T) 0;
				}
			any = false;
			#pragma unroll
			for (int r = 0; r < RPG; r++)
			{
				#pragma unroll
				for (int u = 0; u < U; u++)
				{
					T xv;
					if (LDSX)
						xv = c[r][u] >= 0 ? xs[c[r][u] - lo] : (T) 0;
					else
						xv = c[r][u] >= 0 ? x[c[r][u]] : (T) 0;
					acc[r] = c[r][u] >= 0 ? fma_t<T>(v[r][u], xv, acc[r]) : acc[r];
				}
				j[r] += U * G;
				any |= j[r] < je[r];
			}
		}
		T mine = 0;
		#pragma unroll
		for (int r = 0; r < RPG; r++)
		{
			const T total = group_reduce_sum<T, G>(acc[r]);
			mine = lane == r ? total : mine;
		}
		if (lane < RPG && base + lane < r1)
			y[base + lane] = beta ? y[base + lane] + mine : mine;
		base = nbase;
		rp_mine = rp_next;
	}
}

// CI = int: global column indices. CI = unsigned short: indices RELATIVE to the block's window (every block has one and
// no window is wider than 65 536 columns) — 2 bytes per non-zero instead of 4 in the stream, and no subtraction.
template <typename T, typename CI, int G, bool NT>
__global__ __launch_bounds__(WIN_BLOCK) void
csr_window_kernel(const int * __restrict__ row_ptr, const CI * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, T * __restrict__ y, int beta, const int * __restrict__ blk_row,
		const int * __restrict__ blk_lo, const int * __restrict__ blk_w, int multi_rows, XcdMap map)
{
	extern __shared__ __align__(16) unsigned char window_smem[];
	T * xs = reinterpret_cast<T *>(window_smem);
	const unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int r0 = blk_row[tile], r1 = blk_row[tile + 1];
	const int lo_x = blk_lo[tile], w = blk_w[tile];
	const int lo = sizeof(CI) == 2 ? 0 : lo_x;          // what the stored indices are relative to
	if (w > 0)
	{
		for (int i = threadIdx.x; i < w; i += WIN_BLOCK)
			xs[i] = x[lo_x + i];
		__syncthreads();
		// measured: several rows in flight win while the window is small (cant twin fp32 8.3 -> 7.1 us; many workgroups per
		// CU), the two-deep single-row pipeline wins with ~50 KiB windows (pwtk twin fp32 18.6 vs 19.6 us)
		if constexpr (G <= 32)
		{
			if (multi_rows)
				window_rows_multi<T, CI, G, (G <= 16 ? 4 : 2), NT, true>(row_ptr, col, val, x, xs, y, r0, r1, lo, beta);
			else
				window_rows<T, CI, G, NT, true>(row_ptr, col, val, x, xs, y, r0, r1, lo, beta);
		}
		else
			window_rows<T, CI, G, NT, true>(row_ptr, col, val, x, xs, y, r0, r1, lo, beta);
	}
	else
		window_rows<T, CI, G, NT, false>(row_ptr, col, val, x, xs, y, r0, r1, lo, beta);
}

// ------------------------------------------------------------------------------------------------ launcher

template <typename T, typename CI, int G>
static int
csr_window_launch(const int * row_ptr, const void * col_v, const void * val, const void * x, void * y, const int * blk_row,
		const int * blk_lo, const int * blk_w, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	const CI * col = (const CI *) col_v;
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	// more than 64 KiB of dynamic LDS has to be granted per kernel function, once per device
	static int granted_nt[64] = {0}, granted[64] = {0};
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	dev = dev < 0 || dev >= 64 ? 0 : dev;
	int & have = cfg.nt ? granted_nt[dev] : granted[dev];
	if (lds_bytes > have)
	{
		if (cfg.nt)
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&csr_window_kernel<T, CI, G, true>),
					hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		else
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&csr_window_kernel<T, CI, G, false>),
					hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		have = lds_bytes;
	}
	if (cfg.nt)
		hipLaunchKernelGGL((csr_window_kernel<T, CI, G, true>), dim3(grid), dim3(WIN_BLOCK), lds_bytes, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, cfg.beta, blk_row, blk_lo, blk_w, lds_bytes <= 16384 ? 1 : 0, cfg.map);
	else
		hipLaunchKernelGGL((csr_window_kernel<T, CI, G, false>), dim3(grid), dim3(WIN_BLOCK), lds_bytes, stream, row_ptr, col,
				(const T *) val, (const T *) x, (T *) y, cfg.beta, blk_row, blk_lo, blk_w, lds_bytes <= 16384 ? 1 : 0, cfg.map);
	HIP_TRY(hipGetLastError());
	return 0;
}

template <typename T, typename CI>
static int
csr_window_dispatch(int G, const int * row_ptr, const void * col, const void * val, const void * x, void * y, const int * blk_row,
		const int * blk_lo, const int * blk_w, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	switch (G)
	{
		case 8:  return csr_window_launch<T, CI, 8>(row_ptr, col, val, x, y, blk_row, blk_lo, blk_w, lds_bytes, cfg, stream, grid_out);
		case 16: return csr_window_launch<T, CI, 16>(row_ptr, col, val, x, y, blk_row, blk_lo, blk_w, lds_bytes, cfg, stream, grid_out);
		case 32: return csr_window_launch<T, CI, 32>(row_ptr, col, val, x, y, blk_row, blk_lo, blk_w, lds_bytes, cfg, stream, grid_out);
		case 64: return csr_window_launch<T, CI, 64>(row_ptr, col, val, x, y, blk_row, blk_lo, blk_w, lds_bytes, cfg, stream, grid_out);
	}
	set_error("csr_stream mode 4: lanes_per_row must be 8, 16, 32 or 64 (got %d)", G);
	return 1;
}

int
csr_window_lds_budget()
{
	return 128 * 1024;                 // of the 160 KiB per CU: one workgroup per CU at the limit
}

int
launch_csr_window(bool f32, int lanes_per_row, const int * row_ptr, const void * col, int col16, const void * val, const void * x, void * y,
		const int * blk_row, const int * blk_lo, const int * blk_w, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream,
		long * grid_out)
{
	if (col16)
		return f32 ? csr_window_dispatch<float, unsigned short>(lanes_per_row, row_ptr, col, val, x, y, blk_row, blk_lo, blk_w, lds_bytes, cfg, stream, grid_out)
		           : csr_window_dispatch<double, unsigned short>(lanes_per_row, row_ptr, col, val, x, y, blk_row, blk_lo, blk_w, lds_bytes, cfg, stream, grid_out);
	return f32 ? csr_window_dispatch<float, int>(lanes_per_row, row_ptr, col, val, x, y, blk_row, blk_lo, blk_w, lds_bytes, cfg, stream, grid_out)
	           : csr_window_dispatch<double, int>(lanes_per_row, row_ptr, col, val, x, y, blk_row, blk_lo, blk_w, lds_bytes, cfg, stream, grid_out);
}

}  // namespace spmv
