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
//         index group g of a slice:  [64 lanes][4 steps] u16  = 512 bytes;   both arrays padded to whole groups of 4 steps
//         values of group g: 16 bytes per lane and load — fp64 [2 pairs][64 lanes][2 steps], fp32 [64 lanes][4 steps]
//         (launch.hpp: sellw_val_pos): the kernels run at the issue rate of their vector-memory instructions, and a group is
//         2 + 1 (fp64) or 1 + 1 (fp32) of them instead of 4 + 1 with one value per lane and load
//     -> sizeof(V) + 2 bytes per stored entry instead of CSR's sizeof(V) + 4 (fp32: 6 instead of 8);
//   * S waves share a slice when the matrix has too few slices to fill 256 CUs (wave p takes index groups p, p+S, ...; the
//     partial sums meet in LDS and are added in wave order). With S = 1 a lane walks its row left to right with one FMA per
//     element: bit-identical to the sequential CSR loop (csr.cpp:334-350), like sell_kernel<64>.
// Reference counterparts: sell_sorted.cpp:338-419 (gather + y scatter through the permutation), sell_c_s.cpp:58-60.

#include "launch.hpp"

namespace spmv {

typedef unsigned sellw_uint2 __attribute__((ext_vector_type(2)));

// One or two index groups (4 steps each) of one lane: an 8-byte index load + 4 value loads per group, all issued before the
// first LDS read. Values and indices are both padded to whole groups (<= 3 zero steps per slice), so every group is full.
// (Measured slower on the same twins: a deeper batch of 4 groups with clamped re-loads for short tails — cant 10.2 vs 9.6 us; all
// of a wave's first 6 groups loaded up front behind wave-uniform branches — cant 10.0 vs 9.3 us, pwtk fp32 18.1 vs 16.5 us.)
template <typename T>
struct SellwPair {
	sellw_uint2 d0, d1;
	T a[4], b[4];
};

// the 4 values of one lane in one group; vp = the group's first element + (16 / sizeof(T)) * lane
template <typename T, bool NT>
__device__ __forceinline__ void
sellw_values(const T * __restrict__ vp, T (&v)[4])
{
	if constexpr (sizeof(T) == 8)
	{
		typedef T T2 __attribute__((ext_vector_type(2)));
		const T2 w0 = ld_stream<NT>(reinterpret_cast<const T2 *>(vp));
		const T2 w1 = ld_stream<NT>(reinterpret_cast<const T2 *>(vp + 2 * WAVE));
		v[0] = w0.x;
		v[1] = w0.y;
		v[2] = w1.x;
		v[3] = w1.y;
	}
	else
	{
		typedef T T4 __attribute__((ext_vector_type(4)));
		const T4 w = ld_stream<NT>(reinterpret_cast<const T4 *>(vp));
		v[0] = w.x;
		v[1] = w.y;
		v[2] = w.z;
		v[3] = w.w;
	}
}

template <typename T, bool NT>
__device__ __forceinline__ void
sellw_load2(SellwPair<T> & p, const sellw_uint2 * __restrict__ ip0, const T * __restrict__ vp0, const sellw_uint2 * __restrict__ ip1,
		const T * __restrict__ vp1)
{
	p.d0 = ld_stream<NT>(ip0);
	p.d1 = ld_stream<NT>(ip1);
	sellw_values<T, NT>(vp0, p.a);
	sellw_values<T, NT>(vp1, p.b);
}

template <typename T>
__device__ __forceinline__ void
sellw_consume2(const SellwPair<T> & p, const T * __restrict__ xs, T & s)
{
	const T x0 = xs[p.d0.x & 0xffffu], x1 = xs[p.d0.x >> 16], x2 = xs[p.d0.y & 0xffffu], x3 = xs[p.d0.y >> 16];
	const T z0 = xs[p.d1.x & 0xffffu], z1 = xs[p.d1.x >> 16], z2 = xs[p.d1.y & 0xffffu], z3 = xs[p.d1.y >> 16];
	s = fma_t<T>(p.a[0], x0, s);
	s = fma_t<T>(p.a[1], x1, s);
	s = fma_t<T>(p.a[2], x2, s);
	s = fma_t<T>(p.a[3], x3, s);
	s = fma_t<T>(p.b[0], z0, s);
	s = fma_t<T>(p.b[1], z1, s);
	s = fma_t<T>(p.b[2], z2, s);
	s = fma_t<T>(p.b[3], z3, s);
}

template <typename T, bool NT>
__device__ __forceinline__ void
sellw_group1(const sellw_uint2 * __restrict__ ip, const T * __restrict__ vp, const T * __restrict__ xs, T & s)
{
	const sellw_uint2 d = ld_stream<NT>(ip);
	T v[4];
	sellw_values<T, NT>(vp, v);
	const T x0 = xs[d.x & 0xffffu], x1 = xs[d.x >> 16], x2 = xs[d.y & 0xffffu], x3 = xs[d.y >> 16];
	s = fma_t<T>(v[0], x0, s);
	s = fma_t<T>(v[1], x1, s);
	s = fma_t<T>(v[2], x2, s);
	s = fma_t<T>(v[3], x3, s);
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
	const T * vp = val + v_off + (16 / (int) sizeof(T)) * lane;
	const sellw_uint2 * ip = reinterpret_cast<const sellw_uint2 *>(idx + i_off) + lane;
	// the wave's first two index groups are in flight while the window of x is copied into LDS
	int g = part;
	const bool head = active && g + S < groups;
	SellwPair<T> p;
	if (head)
		sellw_load2<T, NT>(p, ip + (size_t) g * WAVE, vp + (size_t) g * 4 * WAVE, ip + (size_t) (g + S) * WAVE, vp + (size_t) (g + S) * 4 * WAVE);
	for (int i = threadIdx.x; i < w; i += blockDim.x)
		xs[i] = x[lo + i];
	if (threadIdx.x == 0)
		xs[w] = 0;                          // what the padding of EMPTY rows points at: 0 * 0, never 0 * Inf from some other row's column
	__syncthreads();
	T s = 0;
	if (active)
	{
		if (head)
		{
			sellw_consume2<T>(p, xs, s);
			g += 2 * S;
		}
		for (; g + S < groups; g += 2 * S)
		{
			sellw_load2<T, NT>(p, ip + (size_t) g * WAVE, vp + (size_t) g * 4 * WAVE, ip + (size_t) (g + S) * WAVE, vp + (size_t) (g + S) * 4 * WAVE);
			sellw_consume2<T>(p, xs, s);
		}
		if (g < groups)
			sellw_group1<T, NT>(ip + (size_t) g * WAVE, vp + (size_t) g * 4 * WAVE, xs, s);
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
		case 8: return sell_window_launch_s<T, 8>(threads, grp, sdesc, idx, val, row_of_sorted, x, y, m, lds_window_bytes, cfg, stream, grid_out);
	}
	set_error("sell window: waves per slice must be 1, 2, 4 or 8 (got %d)", S);
	return 1;
}

// ------------------------------------------------------------------------------------------------ symmetric storage
// One stored triangle T of a symmetric matrix (KEEP_SYMMETRY builds of the harness: csr_to_format(..., symmetric = 1,
// symmetry_expanded = 0), csr_sym.cpp:118-123; the product is y = (T + T^t - diag T) x, csr_sym.cpp:191-267, gold in
// bench_spmv.cpp:135-148) in the SAME layout — slice groups, 16-bit window-relative columns — multiplied WITHOUT expanding it:
// half the matrix stream. A stored entry (i, j, a) adds a*x[j] to the row's accumulator (a register, as above) and, when j != i,
// a*x[i] to y[j]. That second, scattered addition is what makes symmetric storage a bad trade on a GPU in general (fp64 atomics on
// scattered global addresses: 24 G/s, tools/atomic_bench.hip) — but for a BANDED matrix j lies inside the slice group's own column
// window, which this kernel already keeps in LDS for x: the group keeps a y window beside it (fp64 for both precisions) and the
// scatter is an LDS atomic (ds_add_f64: 0.44 clocks per entry and CU, profiles/r03_gather_bench2.txt). When the group is done its
// y window — the group's own rows and the rows above / below them that its entries reach — is added to y in global memory with
// one coalesced sweep of atomics (contiguous fp64 atomics: 175 G/s), so neighbouring groups' windows may overlap freely. y is
// cleared first (beta = 0). The window covers the group's rows as well as its columns (the diagonal need not be stored).
// Padding entries point at the spare slot behind the window (x = 0, a y nobody reads). Sums to the tolerance, not bit-reproducible
// (atomics) — neither is the reference's csr_sym kernel with more than one thread (compare-and-swap scatter, csr_sym.cpp:204-232).
template <typename T>
__device__ __forceinline__ void
sellw_sym_group(sellw_uint2 d, const T (&v)[4], const T * __restrict__ xs, double * __restrict__ ys, unsigned me, T xi, T & s)
{
	const unsigned j0 = d.x & 0xffffu, j1 = d.x >> 16, j2 = d.y & 0xffffu, j3 = d.y >> 16;
	const T x0 = xs[j0], x1 = xs[j1], x2 = xs[j2], x3 = xs[j3];
	s = fma_t<T>(v[0], x0, s);
	s = fma_t<T>(v[1], x1, s);
	s = fma_t<T>(v[2], x2, s);
	s = fma_t<T>(v[3], x3, s);
	// the mirrored entries: nothing for a diagonal entry (j == i) — branch-free, an add of zero
	unsafeAtomicAdd(&ys[j0], j0 == me ? 0.0 : (double) v[0] * (double) xi);
	unsafeAtomicAdd(&ys[j1], j1 == me ? 0.0 : (double) v[1] * (double) xi);
	unsafeAtomicAdd(&ys[j2], j2 == me ? 0.0 : (double) v[2] * (double) xi);
	unsafeAtomicAdd(&ys[j3], j3 == me ? 0.0 : (double) v[3] * (double) xi);
}

template <typename T, int S, bool NT>
__global__ __launch_bounds__(1024) void
sell_window_sym_kernel(const int * __restrict__ grp, const int64_t * __restrict__ sdesc, const unsigned short * __restrict__ idx,
		const T * __restrict__ val, const int * __restrict__ row_of_sorted, const T * __restrict__ x, T * __restrict__ y,
		int m, int ys_off /* bytes from the x window to the y window */, XcdMap map)
{
	extern __shared__ __align__(16) unsigned char sellw_smem[];
	T * xs = reinterpret_cast<T *>(sellw_smem);
	double * ys = reinterpret_cast<double *>(sellw_smem + ys_off);
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
	const T * vp = val + v_off + (16 / (int) sizeof(T)) * lane;
	const sellw_uint2 * ip = reinterpret_cast<const sellw_uint2 *>(idx + i_off) + lane;
	const long sorted_row = (long) slice * WAVE + lane;
	const bool mine = active && sorted_row < m;
	const int row = mine ? row_of_sorted[sorted_row] : lo;
	for (int i = threadIdx.x; i < w; i += blockDim.x)
	{
		xs[i] = x[lo + i];
		ys[i] = 0;
	}
	if (threadIdx.x == 0)
	{
		xs[w] = 0;                          // what padding entries read ...
		ys[w] = 0;                          // ... and add into
	}
	__syncthreads();
	const unsigned me = (unsigned) (row - lo);
	const T xi = mine ? xs[me] : T(0);
	T s = 0;
	if (active)
	{
		int g = part;
		for (; g + S < groups; g += 2 * S)
		{
			const sellw_uint2 d0 = ld_stream<NT>(ip + (size_t) g * WAVE), d1 = ld_stream<NT>(ip + (size_t) (g + S) * WAVE);
			T a[4], b[4];
			sellw_values<T, NT>(vp + (size_t) g * 4 * WAVE, a);
			sellw_values<T, NT>(vp + (size_t) (g + S) * 4 * WAVE, b);
			sellw_sym_group<T>(d0, a, xs, ys, me, xi, s);
			sellw_sym_group<T>(d1, b, xs, ys, me, xi, s);
		}
		if (g < groups)
		{
			const sellw_uint2 d0 = ld_stream<NT>(ip + (size_t) g * WAVE);
			T a[4];
			sellw_values<T, NT>(vp + (size_t) g * 4 * WAVE, a);
			sellw_sym_group<T>(d0, a, xs, ys, me, xi, s);
		}
		if (mine)
			unsafeAtomicAdd(&ys[me], (double) s);          // the row's own sum (S waves of a slice each add their part)
	}
	__syncthreads();
	// the group's y window goes to global memory: contiguous atomics, zeros skipped (rows the group does not reach)
	for (int i = threadIdx.x; i < w; i += blockDim.x)
	{
		const double v = ys[i];
		if (v != 0.0)
			unsafeAtomicAdd(&y[lo + i], (T) v);
	}
}

template <typename T, int S>
static int
sell_window_sym_launch_s(int threads, const int * grp, const int64_t * sdesc, const unsigned short * idx, const void * val, const int * row_of_sorted,
		const void * x, void * y, int m, int lds_window_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	const unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (!cfg.beta && m > 0)
		HIP_TRY(hipMemsetAsync(y, 0, (size_t) m * sizeof(T), stream));
	if (grid == 0)
		return 0;
	// lds_window_bytes = bytes of the x window incl. its spare slot; the y window (fp64) follows, 16-byte aligned
	const int slots = lds_window_bytes / (int) sizeof(T);
	const int ys_off = (lds_window_bytes + 15) / 16 * 16;
	const int lds_bytes = ys_off + slots * 8;
	static int granted[64][2] = {{0}};
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	dev = dev < 0 || dev >= 64 ? 0 : dev;
	int & have = granted[dev][cfg.nt ? 1 : 0];
	if (lds_bytes > have)
	{
		if (cfg.nt)
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&sell_window_sym_kernel<T, S, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		else
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&sell_window_sym_kernel<T, S, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		have = lds_bytes;
	}
	if (cfg.nt)
		hipLaunchKernelGGL((sell_window_sym_kernel<T, S, true>), dim3(grid), dim3(threads), lds_bytes, stream, grp, sdesc, idx, (const T *) val,
				row_of_sorted, (const T *) x, (T *) y, m, ys_off, cfg.map);
	else
		hipLaunchKernelGGL((sell_window_sym_kernel<T, S, false>), dim3(grid), dim3(threads), lds_bytes, stream, grp, sdesc, idx, (const T *) val,
				row_of_sorted, (const T *) x, (T *) y, m, ys_off, cfg.map);
	HIP_TRY(hipGetLastError());
	return 0;
}

int
launch_sell_window_sym(bool f32, int waves_per_slice, int slices_per_group, const int * grp, const int64_t * sdesc, const unsigned short * idx,
		const void * val, const int * row_of_sorted, const void * x, void * y, int m, int lds_window_bytes, const LaunchCfg & cfg,
		hipStream_t stream, long * grid_out)
{
	const int threads = waves_per_slice * slices_per_group * WAVE;
	if (threads < WAVE || threads > 1024)
	{
		set_error("sell window (symmetric): %d slices x %d waves per workgroup (at most 16 waves)", slices_per_group, waves_per_slice);
		return 1;
	}
	#define SYM_ARGS threads, grp, sdesc, idx, val, row_of_sorted, x, y, m, lds_window_bytes, cfg, stream, grid_out
	switch (waves_per_slice)
	{
		case 1: return f32 ? sell_window_sym_launch_s<float, 1>(SYM_ARGS) : sell_window_sym_launch_s<double, 1>(SYM_ARGS);
		case 2: return f32 ? sell_window_sym_launch_s<float, 2>(SYM_ARGS) : sell_window_sym_launch_s<double, 2>(SYM_ARGS);
		case 4: return f32 ? sell_window_sym_launch_s<float, 4>(SYM_ARGS) : sell_window_sym_launch_s<double, 4>(SYM_ARGS);
	}
	#undef SYM_ARGS
	set_error("sell window (symmetric): waves per slice must be 1, 2 or 4 (got %d)", waves_per_slice);
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
