// SELL-64-sigma with the slice group's window of x in LDS and 16-bit window-relative column indices.
//
// Why: on FEM / banded matrices (pwtk, cant) the SELL kernels of kernels_sell.hip are bound by the x gathers, not by the
// matrix stream — the vector memory pipeline handles a gather lane by lane (one L1 access per gathered element), so a
// 64-row slice costs 64 L1 accesses per step on top of its coalesced value / index loads. But the rows of a few
// neighbouring slices only touch a narrow window of columns (cant twin: 128 rows -> ~800 columns; pwtk twin: ~13 000
// columns for any group of rows, the band), and MI355X has 160 KiB of LDS per CU. So, as csr_window_kernel does for CSR:
//   * a workgroup owns a GROUP of NS consecutive 64-row slices; the rows are sorted by length (descending, stable) inside
//     the group only (sigma = 64*NS), so the group is a contiguous range of the matrix's rows;
//   * the group's column window x[lo .. lo+w) is copied into LDS once with coalesced loads and every gather is an LDS read;
//   * column indices are stored RELATIVE to lo in 16 bits (w <= 65 536), four steps of a lane packed into one 8-byte load:
//         index group g of a slice:  [64 lanes][4 steps] u16  = 512 bytes;   values: [step][64 lanes]; both padded to 4 steps
//     -> sizeof(V) + 2 bytes per stored entry instead of CSR's sizeof(V) + 4 (fp32: 6 instead of 8);
//   * S waves share a slice when the matrix has too few slices to fill 256 CUs (wave p takes index groups p, p+S, ...; the
//     partial sums meet in LDS and are added in wave order). With S = 1 a lane walks its row left to right with one FMA per
//     element: bit-identical to the sequential CSR loop (csr.cpp:334-350), like sell_kernel<64>.
// Reference counterparts: sell_sorted.cpp:338-419 (gather + y scatter through the permutation), sell_c_s.cpp:58-60.

#include "launch.hpp"

namespace spmv {

typedef unsigned sellw_uint2 __attribute__((ext_vector_type(2)));

// A batch = up to NG index groups (4 steps each) of one lane: NG 8-byte index loads + 4*NG value loads, all issued before the
// first LDS read. Values and indices are both padded to whole groups (<= 3 zero steps per slice), so every group is full.
constexpr int SELLW_NG = 4;

template <typename T>
struct SellwBatch {
	sellw_uint2 d[SELLW_NG];
	T v[SELLW_NG][4];
};

// groups g0, g0 + S, ... (cnt of them, 1 <= cnt <= NG, wave-uniform); slots past cnt re-load the last group and are ignored
template <typename T, bool NT>
__device__ __forceinline__ void
sellw_load(SellwBatch<T> & b, const sellw_uint2 * __restrict__ ip, const T * __restrict__ vp, int g0, int S, int cnt)
{
	#pragma unroll
	for (int k = 0; k < SELLW_NG; k++)
	{
		const int g = g0 + (k < cnt ? k : cnt - 1) * S;
		b.d[k] = ld_stream<NT>(ip + (size_t) g * WAVE);
		#pragma unroll
		for (int u = 0; u < 4; u++)
			b.v[k][u] = ld_stream<NT>(vp + (size_t) g * 4 * WAVE + u * WAVE);
	}
}

template <typename T>
__device__ __forceinline__ void
sellw_consume(const SellwBatch<T> & b, const T * __restrict__ xs, T & s, int cnt)
{
	T xv[SELLW_NG][4];
	#pragma unroll
	for (int k = 0; k < SELLW_NG; k++)
	{
		xv[k][0] = xs[b.d[k].x & 0xffffu];
		xv[k][1] = xs[b.d[k].x >> 16];
		xv[k][2] = xs[b.d[k].y & 0xffffu];
		xv[k][3] = xs[b.d[k].y >> 16];
	}
	#pragma unroll
	for (int k = 0; k < SELLW_NG; k++)
		if (k < cnt)                                 // wave-uniform
		{
			#pragma unroll
			for (int u = 0; u < 4; u++)
				s = fma_t<T>(b.v[k][u], xv[k][u], s);
		}
}

// grp[4*g .. 4*g+3] = first column of the window, its width, first slice, number of slices of group g
// sdesc[2*s] = first value element of slice s, sdesc[2*s+1] = first index element (u16 units); both arrays hold the slice's
// width rounded up to 4 steps; terminated by one more pair
template <typename T, int S, bool NT>
__global__ __launch_bounds__(1024) void
sell_window_kernel(const int * __restrict__ grp, const int64_t * __restrict__ sdesc, const unsigned short * __restrict__ idx,
		const T * __restrict__ val, const int * __restrict__ row_of_sorted, const T * __restrict__ x, T * __restrict__ y,
		int m, int beta, int part_off /* bytes from the window to the partial sums */, XcdMap map)
{
	extern __shared__ __align__(16) unsigned char sellw_smem[];
	T * xs = reinterpret_cast<T *>(sellw_smem);
	const unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int lo = grp[4 * tile], w = grp[4 * tile + 1], slice0 = grp[4 * tile + 2], ns = grp[4 * tile + 3];
	const int lane = threadIdx.x % WAVE;
	const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x / WAVE);
	const int part = wave % S;
	const bool active = wave / S < ns;
	const int slice = slice0 + (active ? wave / S : 0);
	const int64_t v_off = sdesc[2 * slice], i_off = sdesc[2 * slice + 1], v_next = sdesc[2 * slice + 2];
	const int groups = (int) ((v_next - v_off) / (4 * WAVE));
	int left = active && groups > part ? (groups - part + S - 1) / S : 0;      // index groups this wave sums
	const T * vp = val + v_off + lane;
	const sellw_uint2 * ip = reinterpret_cast<const sellw_uint2 *>(idx + i_off) + lane;
	// the first batch of the matrix stream is in flight while the window of x is copied into LDS
	SellwBatch<T> b0, b1;
	int g = part;
	if (left > 0)
		sellw_load<T, NT>(b0, ip, vp, g, S, min(left, SELLW_NG));
	for (int i = threadIdx.x; i < w; i += blockDim.x)
		xs[i] = x[lo + i];
	__syncthreads();
	T s = 0;
	while (left > 0)
	{
		// b0 holds the groups from g on; the batch after it is loaded before b0 is consumed
		const int c0 = min(left, SELLW_NG), l1 = left - c0;
		if (l1 > 0)
			sellw_load<T, NT>(b1, ip, vp, g + SELLW_NG * S, S, min(l1, SELLW_NG));
		sellw_consume<T>(b0, xs, s, c0);
		if (l1 <= 0)
			break;
		const int c1 = min(l1, SELLW_NG), l2 = l1 - c1;
		if (l2 > 0)
			sellw_load<T, NT>(b0, ip, vp, g + 2 * SELLW_NG * S, S, min(l2, SELLW_NG));
		sellw_consume<T>(b1, xs, s, c1);
		g += 2 * SELLW_NG * S;
		left = l2;
	}
	if constexpr (S > 1)
	{
		T * sp = reinterpret_cast<T *>(sellw_smem + part_off);
		sp[(size_t) wave * WAVE + lane] = s;
		__syncthreads();
		if (part != 0 || !active)
			return;
		s = sp[(size_t) wave * WAVE + lane];
		#pragma unroll
		for (int u = 1; u < S; u++)
			s += sp[(size_t) (wave + u) * WAVE + lane];
	}
	else if (!active)
		return;
	const long sorted_row = (long) slice * WAVE + lane;
	if (sorted_row < m)
	{
		T * yp = y + row_of_sorted[sorted_row];
		*yp = beta ? *yp + s : s;
	}
}

template <typename T, int S>
static int
sell_window_launch_s(int threads, const int * grp, const int64_t * sdesc, const unsigned short * idx, const void * val, const int * row_of_sorted,
		const void * x, void * y, int m, int lds_window_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	const unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	const int part_off = (lds_window_bytes + 15) / 16 * 16;
	const int lds_bytes = part_off + (S > 1 ? threads * (int) sizeof(T) : 0);
	// more than 64 KiB of dynamic LDS has to be granted per kernel function, once per device
	static int granted[64][2] = {{0}};
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	dev = dev < 0 || dev >= 64 ? 0 : dev;
	int & have = granted[dev][cfg.nt ? 1 : 0];
	if (lds_bytes > have)
	{
		if (cfg.nt)
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&sell_window_kernel<T, S, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		else
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&sell_window_kernel<T, S, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		have = lds_bytes;
	}
	if (cfg.nt)
		hipLaunchKernelGGL((sell_window_kernel<T, S, true>), dim3(grid), dim3(threads), lds_bytes, stream, grp, sdesc, idx, (const T *) val,
				row_of_sorted, (const T *) x, (T *) y, m, cfg.beta, part_off, cfg.map);
	else
		hipLaunchKernelGGL((sell_window_kernel<T, S, false>), dim3(grid), dim3(threads), lds_bytes, stream, grp, sdesc, idx, (const T *) val,
				row_of_sorted, (const T *) x, (T *) y, m, cfg.beta, part_off, cfg.map);
	HIP_TRY(hipGetLastError());
	return 0;
}

template <typename T>
static int
sell_window_dispatch(int S, int threads, const int * grp, const int64_t * sdesc, const unsigned short * idx, const void * val,
		const int * row_of_sorted, const void * x, void * y, int m, int lds_window_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	switch (S)
	{
		case 1: return sell_window_launch_s<T, 1>(threads, grp, sdesc, idx, val, row_of_sorted, x, y, m, lds_window_bytes, cfg, stream, grid_out);
		case 2: return sell_window_launch_s<T, 2>(threads, grp, sdesc, idx, val, row_of_sorted, x, y, m, lds_window_bytes, cfg, stream, grid_out);
		case 4: return sell_window_launch_s<T, 4>(threads, grp, sdesc, idx, val, row_of_sorted, x, y, m, lds_window_bytes, cfg, stream, grid_out);
	}
	set_error("sell window: waves per slice must be 1, 2 or 4 (got %d)", S);
	return 1;
}

int
sell_window_lds_budget()
{
	return 128 * 1024;                 // bytes of x a group's window may take (of the 160 KiB per CU)
}

int
launch_sell_window(bool f32, int waves_per_slice, int slices_per_group, const int * grp, const int64_t * sdesc, const unsigned short * idx,
		const void * val, const int * row_of_sorted, const void * x, void * y, int m, int lds_window_bytes, const LaunchCfg & cfg,
		hipStream_t stream, long * grid_out)
{
	const int threads = waves_per_slice * slices_per_group * WAVE;
	if (threads < WAVE || threads > 1024)
	{
		set_error("sell window: %d slices x %d waves per workgroup (at most 16 waves)", slices_per_group, waves_per_slice);
		return 1;
	}
	return f32 ? sell_window_dispatch<float>(waves_per_slice, threads, grp, sdesc, idx, val, row_of_sorted, x, y, m, lds_window_bytes, cfg, stream, grid_out)
	           : sell_window_dispatch<double>(waves_per_slice, threads, grp, sdesc, idx, val, row_of_sorted, x, y, m, lds_window_bytes, cfg, stream, grid_out);
}

}  // namespace spmv
