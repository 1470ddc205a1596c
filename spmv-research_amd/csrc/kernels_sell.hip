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

#include <type_traits>

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

// The BSC library's own slice height (sell_c_s.cpp:58-60: C = 256): one workgroup of 256 lanes = one slice, lane r owns sorted
// row r of the slice and walks it left to right with one FMA per element (bit-identical to the sequential CSR loop, as C = 64).
constexpr int SELL_WIDE_C = 256;

template <typename T, bool NT>
__global__ __launch_bounds__(SELL_WIDE_C) void
sell_wide_kernel(const int64_t * __restrict__ slice_ptr, const int * __restrict__ col, const T * __restrict__ val,
		const int * __restrict__ row_of_sorted, const T * __restrict__ x, T * __restrict__ y,
		int m, int num_slices, int beta, XcdMap map)
{
	constexpr int C = SELL_WIDE_C;
	const unsigned slice = xcd_tile(blockIdx.x, map);
	if (slice == NO_TILE || (int) slice >= num_slices)
		return;
	const int64_t p_e = slice_ptr[slice + 1];
	int64_t p = slice_ptr[slice] + threadIdx.x;
	T s = 0;
	for (; p + 3 * C < p_e; p += 4 * C)
	{
		const int c0 = ld_stream<NT>(col + p), c1 = ld_stream<NT>(col + p + C), c2 = ld_stream<NT>(col + p + 2 * C), c3 = ld_stream<NT>(col + p + 3 * C);
		const T v0 = ld_stream<NT>(val + p), v1 = ld_stream<NT>(val + p + C), v2 = ld_stream<NT>(val + p + 2 * C), v3 = ld_stream<NT>(val + p + 3 * C);
		const T x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
		s = fma_t<T>(v0, x0, s);
		s = fma_t<T>(v1, x1, s);
		s = fma_t<T>(v2, x2, s);
		s = fma_t<T>(v3, x3, s);
	}
	for (; p < p_e; p += C)
		s = fma_t<T>(ld_stream<NT>(val + p), x[ld_stream<NT>(col + p)], s);
	const long sorted_row = (long) slice * C + threadIdx.x;
	if (sorted_row < m)
	{
		T * yp = y + row_of_sorted[sorted_row];
		*yp = beta ? *yp + s : s;
	}
}

// ------------------------------------------------------------------------------------------------ SELL-64-sigma-delta
// Column indices are the only compressible stream of SpMV (values are data, x and y are compulsory). In a 64-row slice
// the 64 lanes of step k hold neighbouring rows, and on banded / stencil / FEM matrices their columns sit within a few
// hundred entries of each other. Per slice the indices are therefore stored as one int32 base per step plus one
// unsigned delta per lane: 8 bits (max delta < 256), 16 bits, or plain int32 when neither fits (mode per slice).
// Steps come in groups of 4 so that a lane's four deltas are ONE dword (8-bit) / ONE dwordx2 (16-bit) load:
//     group of 4 steps, mode 1:  [4 x int32 base][64 lanes x 4 x u8 ]  = 16 + 256 bytes   ( 9.06 B per fp64 non-zero)
//                      mode 2:  [4 x int32 base][64 lanes x 4 x u16]  = 16 + 512 bytes   (10.06 B)
//                      mode 4:  [4 steps][64 lanes] int32               = 1024 bytes       (12 B, as plain SELL)
// The bases are wave-uniform (scalar loads). Arithmetic and its order are exactly those of sell_kernel<C = 64>: one lane
// per row, one FMA per element, left to right -> bit-identical to the sequential CSR loop; lossless by construction.
// HBM bytes per non-zero drop by up to 24 % (fp64) / 37 % (fp32) below the CSR-normalised "algorithmic" 12 / 8 bytes.
typedef int sell_int4 __attribute__((ext_vector_type(4)));
typedef unsigned sell_uint2 __attribute__((ext_vector_type(2)));

// VALUES of a slice are stored in PAIRS of steps: a lane's steps 2p and 2p+1 lie side by side — [pair][lane][2] — so that a group of 4
// steps is TWO 16-byte loads per lane (fp64; global_load_dwordx4) instead of four 8-byte ones. The kernel sits at the issue rate of
// its vector-memory instructions (4 value loads + 4 gathers per group: with the value loads at half the count the nlpkkt240 twin runs
// 9 % faster on the same bytes, profiles/r03_sell_value_pairs.txt). The last step of an odd width stands alone, one element per lane
// (launch.hpp: sell_pair_pos). `vp` = the group's first element + 2 * lane.
template <typename T, bool NT>
__device__ __forceinline__ void
sell_group_values(const T * __restrict__ vp, T (&v)[4])
{
	typedef T T2 __attribute__((ext_vector_type(2)));
	const T2 w0 = ld_stream<NT>(reinterpret_cast<const T2 *>(vp));
	const T2 w1 = ld_stream<NT>(reinterpret_cast<const T2 *>(vp + 2 * WAVE));
	v[0] = w0.x;
	v[1] = w0.y;
	v[2] = w1.x;
	v[3] = w1.y;
}

// the 1..3 real steps of a slice's last group
template <typename T, bool NT, int NSTEPS>
__device__ __forceinline__ void
sell_tail_values(const T * __restrict__ vp, int lane, T (&v)[3])
{
	typedef T T2 __attribute__((ext_vector_type(2)));
	v[1] = v[2] = T(0);
	if (NSTEPS == 1)
		v[0] = ld_stream<NT>(vp - lane);
	else
	{
		const T2 w0 = ld_stream<NT>(reinterpret_cast<const T2 *>(vp));
		v[0] = w0.x;
		v[1] = w0.y;
		if (NSTEPS == 3)
			v[2] = ld_stream<NT>(vp + 2 * WAVE - lane);
	}
}

template <typename T, int MODE, bool NT, int NSTEPS = 4>
__device__ __forceinline__ void
sell_delta_group(const unsigned char * __restrict__ gp /* uniform */, const T * __restrict__ vp, int lane, const T * __restrict__ x, T & s,
		int off = 0)
{
	int c0, c1, c2, c3;
	if constexpr (MODE == 0 || MODE == 3)
	{
		// step-invariant lane offsets: column of lane l at step k = base_k + off_l. MODE 0 (affine slice: 64 consecutive rows of
		// a stencil diagonal) has off_l = l; MODE 3 stores the 64 offsets once per slice (rows of one kind that are not
		// consecutive). Either way no per-step index bytes per lane, one scalar base per step.
		const sell_int4 base = *reinterpret_cast<const sell_int4 *>(gp);
		c0 = base.x + off;
		c1 = base.y + off;
		c2 = base.z + off;
		c3 = base.w + off;
	}
	else if constexpr (MODE == 1)
	{
		const sell_int4 base = *reinterpret_cast<const sell_int4 *>(gp);
		const unsigned d = ld_stream<NT>(reinterpret_cast<const unsigned *>(gp + 16) + lane);
		c0 = base.x + (int) (d & 255u);
		c1 = base.y + (int) ((d >> 8) & 255u);
		c2 = base.z + (int) ((d >> 16) & 255u);
		c3 = base.w + (int) (d >> 24);
	}
	else if constexpr (MODE == 2)
	{
		const sell_int4 base = *reinterpret_cast<const sell_int4 *>(gp);
		const sell_uint2 d = ld_stream<NT>(reinterpret_cast<const sell_uint2 *>(gp + 16) + lane);
		c0 = base.x + (int) (d.x & 0xffffu);
		c1 = base.y + (int) (d.x >> 16);
		c2 = base.z + (int) (d.y & 0xffffu);
		c3 = base.w + (int) (d.y >> 16);
	}
	else
	{
		const int * cp = reinterpret_cast<const int *>(gp) + lane;
		c0 = ld_stream<NT>(cp);
		c1 = ld_stream<NT>(cp + WAVE);
		c2 = ld_stream<NT>(cp + 2 * WAVE);
		c3 = ld_stream<NT>(cp + 3 * WAVE);
	}
	if (NSTEPS == 4)
	{
		T v[4];
		sell_group_values<T, NT>(vp, v);
		const T x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
		s = fma_t<T>(v[0], x0, s);
		s = fma_t<T>(v[1], x1, s);
		s = fma_t<T>(v[2], x2, s);
		s = fma_t<T>(v[3], x3, s);
	}
	else
	{
		// last group of a slice whose width is not a multiple of 4: the value array holds only the real steps
		T v[3];
		sell_tail_values<T, NT, NSTEPS>(vp, lane, v);
		const T x0 = x[c0];
		const T x1 = NSTEPS > 1 ? x[c1] : T(0);
		const T x2 = NSTEPS > 2 ? x[c2] : T(0);
		s = fma_t<T>(v[0], x0, s);
		if (NSTEPS > 1) s = fma_t<T>(v[1], x1, s);
		if (NSTEPS > 2) s = fma_t<T>(v[2], x2, s);
	}
}

// hipcc sinks a load whose result is only used in the NEXT trip of a loop (through `a = na`) down to that use — the loads are
// invariant, so it may — and the hand-written prefetch becomes a dependent load in front of the gathers again. Passing the prefetched
// registers through an empty asm statement at the END of the trip pins them to the trip they were issued in: the wait for them lands
// behind this trip's gathers and FMAs, where it costs nothing.
__device__ __forceinline__ void
sell_pin(unsigned & v)
{
	asm volatile("" : "+v"(v));
}
__device__ __forceinline__ void
sell_pin(sell_uint2 & v, bool both)
{
	unsigned a = v.x, b = both ? v.y : 0u;
	asm volatile("" : "+v"(a), "+v"(b));
	v.x = a;
	if (both)
		v.y = b;
}

// Modes 1 and 2 put a dependent load in front of every gather (deltas -> column -> x): their index words are fetched one pair of
// groups AHEAD, so that a trip costs one exposed round trip (the gathers) like the index-free modes, not two.
template <int MODE>
struct SellDeltaIdx {
	sell_int4 base;
	sell_uint2 d;                                  // MODE 1 uses d.x only
};

template <int MODE, bool NT>
__device__ __forceinline__ void
sell_delta_load_idx(SellDeltaIdx<MODE> & q, const unsigned char * __restrict__ gp /* uniform */, int lane)
{
	q.base = *reinterpret_cast<const sell_int4 *>(gp);
	if constexpr (MODE == 1)
		q.d.x = ld_stream<NT>(reinterpret_cast<const unsigned *>(gp + 16) + lane);
	else
		q.d = ld_stream<NT>(reinterpret_cast<const sell_uint2 *>(gp + 16) + lane);
}

template <int MODE>
__device__ __forceinline__ void
sell_delta_cols(const SellDeltaIdx<MODE> & q, int (&c)[4])
{
	if constexpr (MODE == 1)
	{
		c[0] = q.base.x + (int) (q.d.x & 255u);
		c[1] = q.base.y + (int) ((q.d.x >> 8) & 255u);
		c[2] = q.base.z + (int) ((q.d.x >> 16) & 255u);
		c[3] = q.base.w + (int) (q.d.x >> 24);
	}
	else
	{
		c[0] = q.base.x + (int) (q.d.x & 0xffffu);
		c[1] = q.base.y + (int) (q.d.x >> 16);
		c[2] = q.base.z + (int) (q.d.y & 0xffffu);
		c[3] = q.base.w + (int) (q.d.y >> 16);
	}
}

// the full 4-step groups number g0, g0+gs, ... (n of them) of a slice in mode 1 / 2: FOUR groups per trip (16 steps in flight: all loads of a
// trip are issued before its first FMA), their index words fetched one trip ahead; what is left (0..3 groups) as a pair and / or a single
// group on the index words the last trip already fetched. (Two groups per trip: 1 347 us with every index-free mode off; four: see
// profiles/r03_sell_value_pairs.txt.)
template <typename T, int MODE, bool NT, int NG>
__device__ __forceinline__ void
sell_delta_consume(const SellDeltaIdx<MODE> * q, const T * __restrict__ vp, const int * g, const T * __restrict__ x, T & s)
{
	T v[NG][4];
	int c[NG][4];
	T xv[NG][4];
	#pragma unroll
	for (int u = 0; u < NG; u++)
		sell_group_values<T, NT>(vp + (size_t) g[u] * 4 * WAVE, v[u]);
	#pragma unroll
	for (int u = 0; u < NG; u++)
		sell_delta_cols<MODE>(q[u], c[u]);
	#pragma unroll
	for (int u = 0; u < NG; u++)
		#pragma unroll
		for (int t = 0; t < 4; t++)
			xv[u][t] = x[c[u][t]];
	#pragma unroll
	for (int u = 0; u < NG; u++)
		#pragma unroll
		for (int t = 0; t < 4; t++)
			s = fma_t<T>(v[u][t], xv[u][t], s);
}

template <typename T, int MODE, bool NT>
__device__ __forceinline__ void
sell_delta_piped(const unsigned char * __restrict__ ip, const T * __restrict__ vp, int lane, const T * __restrict__ x, T & s, int g0, int gs, int n)
{
	constexpr int GB = MODE == 1 ? 272 : 528;
	if (n <= 0)
		return;
	auto gidx = [&](int k) { return g0 + (k < n ? k : n - 1) * gs; };       // past the end: the last group again (loaded, not used)
	SellDeltaIdx<MODE> q[4], nq[4];
	#pragma unroll
	for (int u = 0; u < 4; u++)
		sell_delta_load_idx<MODE, NT>(q[u], ip + (size_t) gidx(u) * GB, lane);
	int k = 0;
	for (; k + 4 <= n; k += 4)
	{
		#pragma unroll
		for (int u = 0; u < 4; u++)
			sell_delta_load_idx<MODE, NT>(nq[u], ip + (size_t) gidx(k + 4 + u) * GB, lane);
		const int g[4] = {gidx(k), gidx(k + 1), gidx(k + 2), gidx(k + 3)};
		sell_delta_consume<T, MODE, NT, 4>(q, vp, g, x, s);
		#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			sell_pin(nq[u].d, MODE == 2);
			q[u] = nq[u];
		}
	}
	const int r = n - k;                               // 0..3 groups left; q[0..r-1] hold their index words (wave-uniform branches)
	if (r >= 2)
	{
		const int g[2] = {gidx(k), gidx(k + 1)};
		sell_delta_consume<T, MODE, NT, 2>(q, vp, g, x, s);
	}
	if (r == 1)                                        // (constant indices into q: a run-time one would send the array to scratch memory)
	{
		const int g[1] = {gidx(k)};
		sell_delta_consume<T, MODE, NT, 1>(q, vp, g, x, s);
	}
	if (r == 3)
	{
		const int g[1] = {gidx(k + 2)};
		sell_delta_consume<T, MODE, NT, 1>(q + 2, vp, g, x, s);
	}
}

// groups g0, g0+gs, g0+2gs, ... of one slice (gs = 1: the whole slice, in order)
template <typename T, int MODE, bool NT>
__device__ __forceinline__ T
sell_delta_slice(const unsigned char * __restrict__ ip, const T * __restrict__ vp, int width, int lane, const T * __restrict__ x,
		int g0 = 0, int gs = 1)
{
	constexpr int GB = (MODE == 0 || MODE == 3) ? 16 : MODE == 1 ? 272 : MODE == 2 ? 528 : 1024;     // bytes of one index group
	const int groups = (width + 3) / 4;     // index groups cover the width rounded up to 4 steps, values only the real steps
	const int rem = width - 4 * (groups - 1);
	int off = lane;
	if constexpr (MODE == 3)
	{
		off = ld_stream<NT>(reinterpret_cast<const int *>(ip) + lane);          // the slice's 64 lane offsets, then the groups
		ip += 4 * WAVE;
	}
	T s = 0;
	int g = g0;
	const int last = groups - 1;            // the last group holds `rem` (1..4) steps
	if constexpr (MODE == 1 || MODE == 2)
	{
		const int full = rem == 4 ? groups : last;
		const int n = full > g0 ? (full - g0 + gs - 1) / gs : 0;
		sell_delta_piped<T, MODE, NT>(ip, vp, lane, x, s, g0, gs, n);
		g = g0 + n * gs;
	}
	// 16, then 12, then 8 steps in flight per trip: a slice of 7 full groups (the nlpkkt240 twin's 27-28 entries per row) is 4 + 3. The
	// loads of a trip are independent of its FMAs, so the compiler issues all of them first; deeper trips measured 2-3 % faster than
	// pairs alone on the twin (1 152-1 185 against 1 198-1 207 us, profiles/r03_sell_value_pairs.txt).
	const int full_end = rem == 4 ? groups : last;
	for (; g + 3 * gs < full_end; g += 4 * gs)
	{
		sell_delta_group<T, MODE, NT>(ip + (size_t) g * GB, vp + (size_t) g * 4 * WAVE, lane, x, s, off);
		sell_delta_group<T, MODE, NT>(ip + (size_t) (g + gs) * GB, vp + (size_t) (g + gs) * 4 * WAVE, lane, x, s, off);
		sell_delta_group<T, MODE, NT>(ip + (size_t) (g + 2 * gs) * GB, vp + (size_t) (g + 2 * gs) * 4 * WAVE, lane, x, s, off);
		sell_delta_group<T, MODE, NT>(ip + (size_t) (g + 3 * gs) * GB, vp + (size_t) (g + 3 * gs) * 4 * WAVE, lane, x, s, off);
	}
	for (; g + 2 * gs < full_end; g += 3 * gs)
	{
		sell_delta_group<T, MODE, NT>(ip + (size_t) g * GB, vp + (size_t) g * 4 * WAVE, lane, x, s, off);
		sell_delta_group<T, MODE, NT>(ip + (size_t) (g + gs) * GB, vp + (size_t) (g + gs) * 4 * WAVE, lane, x, s, off);
		sell_delta_group<T, MODE, NT>(ip + (size_t) (g + 2 * gs) * GB, vp + (size_t) (g + 2 * gs) * 4 * WAVE, lane, x, s, off);
	}
	for (; g + gs < (rem == 4 ? groups : last); g += 2 * gs)    // 8 steps in flight per trip
	{
		sell_delta_group<T, MODE, NT>(ip + (size_t) g * GB, vp + (size_t) g * 4 * WAVE, lane, x, s, off);
		sell_delta_group<T, MODE, NT>(ip + (size_t) (g + gs) * GB, vp + (size_t) (g + gs) * 4 * WAVE, lane, x, s, off);
	}
	for (; g < (rem == 4 ? groups : last); g += gs)
		sell_delta_group<T, MODE, NT>(ip + (size_t) g * GB, vp + (size_t) g * 4 * WAVE, lane, x, s, off);
	if (rem != 4 && g == last)
	{
		if (rem == 1)
			sell_delta_group<T, MODE, NT, 1>(ip + (size_t) g * GB, vp + (size_t) g * 4 * WAVE, lane, x, s, off);
		else if (rem == 2)
			sell_delta_group<T, MODE, NT, 2>(ip + (size_t) g * GB, vp + (size_t) g * 4 * WAVE, lane, x, s, off);
		else
			sell_delta_group<T, MODE, NT, 3>(ip + (size_t) g * GB, vp + (size_t) g * 4 * WAVE, lane, x, s, off);
	}
	return s;
}

// MODE 5: lane offsets WITH EXCEPTIONS. Modes 0 and 3 need all 64 rows of a slice to follow one pattern; one row out of line (a
// boundary row of a stencil, a perturbed row) used to send the whole slice back to 8/16-bit deltas per lane and step. Here EVERY lane
// has an offset (its first column minus the reference lane's), the rows that fit (at least 48 of 64) have column = base_k + off_lane
// at every step, and the E <= 16 others add a signed 8-bit correction per step:
//     slice header: [64 x int32 lane offsets][u64 exception mask][8 bytes of padding]                       = 272 bytes
//     group of 4 steps: [4 x int32 base][E x (4 x int8 corrections of exception lane number j)], padded to 16 = 32 .. 80 bytes
// An exception lane loads its four corrections as ONE dword at gp + 16 + 4 * (its rank among the exception lanes); the other lanes
// load the first exception's (one address for all of them) and drop it. (A first version stored the exception lanes' columns as
// 4 x int32 and loaded them with a dwordx4 per lane: 1 447 us on the 5 %-jittered nlpkkt240 twin against 1 425 us for plain 8/16-bit
// deltas although it moves 7 % fewer bytes — the wide load of all 64 lanes cost more than the bytes saved.) That load sits in front
// of the lane's gathers, so it is fetched one pair of groups ahead like the delta words of modes 1 / 2. Same FMAs in the same order
// as every other mode: bit-identical results. A slice with a row that needs more than 8 bits falls back to modes 1 / 2 / 4.
// Slices with at most FOUR exception rows (the common case: 5 % of the rows out of line puts 3.2 into a slice on average) take the
// corrections through the SCALAR cache instead: the 16 bytes behind the bases hold all of them, one s_load brings bases and corrections,
// and a compare-and-select per exception puts its dword into its lane — no vector-memory instruction at all for the indices, as in modes 0 / 3.
// (The kernel runs at the issue rate of its vector-memory instructions: the per-lane dword load of the general path is a ninth
// instruction per group of 4 steps beside 2 value loads and 4 gathers.)
struct SellDeltaIdx5 {
	sell_int4 base;                                // wave-uniform
	unsigned d;                                    // 4 x int8, exception lanes only            (general path)
	sell_int4 corr;                                // corrections of exceptions 0..3, uniform   (scalar path)
};

// `rank` is 0 for the lanes that are no exception: they load the first exception's corrections (one address for all of them) and drop
// them. No branch around the load: with one the compiler cannot count what is outstanding and waits for everything at every trip.
template <bool NT, bool SCALAR>
__device__ __forceinline__ void
sell_delta_load_idx5(SellDeltaIdx5 & q, const unsigned char * __restrict__ gp /* uniform */, int rank)
{
	q.base = *reinterpret_cast<const sell_int4 *>(gp);
	if constexpr (SCALAR)
		q.corr = *reinterpret_cast<const sell_int4 *>(gp + 16);
	else
		q.d = ld_stream<NT>(reinterpret_cast<const unsigned *>(gp + 16) + rank);
}

template <bool SCALAR>
__device__ __forceinline__ void
sell_delta_cols5(const SellDeltaIdx5 & q, bool ex, int lane, int off, const int (&xl)[4], int (&c)[4])
{
	int d;
	if constexpr (SCALAR)
	{
		// exception j's dword into its lane; slots past the slice's last exception name a lane that is none and hold zeros (a spare lane keeps 0)
		d = lane == xl[0] ? q.corr.x : 0;
		d = lane == xl[1] ? q.corr.y : d;
		d = lane == xl[2] ? q.corr.z : d;
		d = lane == xl[3] ? q.corr.w : d;
	}
	else
		d = ex ? (int) q.d : 0;
	c[0] = q.base.x + off + ((d << 24) >> 24);
	c[1] = q.base.y + off + ((d << 16) >> 24);
	c[2] = q.base.z + off + ((d << 8) >> 24);
	c[3] = q.base.w + off + (d >> 24);
}

// keeps what was fetched a trip ahead in the trip it was fetched in (sell_pin above)
template <bool SCALAR>
__device__ __forceinline__ void
sell_pin5(SellDeltaIdx5 & q)
{
	if constexpr (SCALAR)
	{
		int a = q.corr.x, b = q.corr.y, c = q.corr.z, d = q.corr.w;
		asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
		q.corr.x = a;
		q.corr.y = b;
		q.corr.z = c;
		q.corr.w = d;
	}
	else
		sell_pin(q.d);
}

template <typename T, bool NT, bool SCALAR>
__device__ __forceinline__ T
sell_delta_slice5_body(const unsigned char * __restrict__ ip, const T * __restrict__ vp, int width, int lane, const T * __restrict__ x,
		int g0, int gs, unsigned long long mask, int off)
{
	const int E = __popcll(mask);
	const bool ex = (mask >> lane) & 1ull;
	const int rank = ex ? __popcll(mask & ((1ull << lane) - 1ull)) : 0;
	int xl[4] = {0, 0, 0, 0};
	if constexpr (SCALAR)
	{
		unsigned long long mm = mask;
		const int spare = __builtin_ctzll(~mask);                       // a lane that is no exception (at least 48 are not)
		#pragma unroll
		for (int j = 0; j < 4; j++)
		{
			xl[j] = mm ? __builtin_ctzll(mm) : spare;
			mm &= mm - 1ull;
		}
	}
	const size_t GB = 16 + ((size_t) E + 3) / 4 * 16;
	const int groups = (width + 3) / 4;
	const int rem = width - 4 * (groups - 1);
	const int last = groups - 1;
	const int full = rem == 4 ? groups : last;
	const int n = full > g0 ? (full - g0 + gs - 1) / gs : 0;     // full groups this wave takes: g0, g0 + gs, ...
	T s = 0;
	if (n > 0)
	{
		// four groups per trip, index words (and corrections) one trip ahead; the 0..3 groups left as a pair and / or a single one
		// (sell_delta_piped above)
		auto gidx = [&](int k) { return g0 + (k < n ? k : n - 1) * gs; };       // past the end: the last group again (loaded, not used)
		auto consume = [&](const SellDeltaIdx5 * q, const int * g, auto ng) {
			constexpr int NG = decltype(ng)::value;
			T v[NG][4];
			int c[NG][4];
			T xv[NG][4];
			#pragma unroll
			for (int u = 0; u < NG; u++)
				sell_group_values<T, NT>(vp + (size_t) g[u] * 4 * WAVE, v[u]);
			#pragma unroll
			for (int u = 0; u < NG; u++)
				sell_delta_cols5<SCALAR>(q[u], ex, lane, off, xl, c[u]);
			#pragma unroll
			for (int u = 0; u < NG; u++)
				#pragma unroll
				for (int t = 0; t < 4; t++)
					xv[u][t] = x[c[u][t]];
			#pragma unroll
			for (int u = 0; u < NG; u++)
				#pragma unroll
				for (int t = 0; t < 4; t++)
					s = fma_t<T>(v[u][t], xv[u][t], s);
		};
		SellDeltaIdx5 q[4], nq[4];
		#pragma unroll
		for (int u = 0; u < 4; u++)
			sell_delta_load_idx5<NT, SCALAR>(q[u], ip + (size_t) gidx(u) * GB, rank);
		int k = 0;
		for (; k + 4 <= n; k += 4)
		{
			#pragma unroll
			for (int u = 0; u < 4; u++)
				sell_delta_load_idx5<NT, SCALAR>(nq[u], ip + (size_t) gidx(k + 4 + u) * GB, rank);
			const int g[4] = {gidx(k), gidx(k + 1), gidx(k + 2), gidx(k + 3)};
			consume(q, g, std::integral_constant<int, 4>());
			#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				sell_pin5<SCALAR>(nq[u]);
				q[u] = nq[u];
			}
		}
		const int r = n - k;
		if (r >= 2)
		{
			const int g[2] = {gidx(k), gidx(k + 1)};
			consume(q, g, std::integral_constant<int, 2>());
		}
		if (r == 1)
		{
			const int g[1] = {gidx(k)};
			consume(q, g, std::integral_constant<int, 1>());
		}
		if (r == 3)
		{
			const int g[1] = {gidx(k + 2)};
			consume(q + 2, g, std::integral_constant<int, 1>());
		}
	}
	// the last group of a slice whose width is not a multiple of 4 (the value array holds only its `rem` real steps); with several
	// waves per slice it belongs to the wave whose sequence g0, g0 + gs, ... reaches it
	if (rem != 4 && last >= g0 && (last - g0) % gs == 0)
	{
		SellDeltaIdx5 q;
		sell_delta_load_idx5<NT, SCALAR>(q, ip + (size_t) last * GB, rank);
		int c[4];
		sell_delta_cols5<SCALAR>(q, ex, lane, off, xl, c);
		const T * vl = vp + (size_t) last * 4 * WAVE;
		T tv[3];
		if (rem == 1)
			sell_tail_values<T, NT, 1>(vl, lane, tv);
		else if (rem == 2)
			sell_tail_values<T, NT, 2>(vl, lane, tv);
		else
			sell_tail_values<T, NT, 3>(vl, lane, tv);
		const T v0 = tv[0], v1 = tv[1], v2 = tv[2];
		const T x0 = x[c[0]];
		const T x1 = rem > 1 ? x[c[1]] : T(0);
		const T x2 = rem > 2 ? x[c[2]] : T(0);
		s = fma_t<T>(v0, x0, s);
		if (rem > 1) s = fma_t<T>(v1, x1, s);
		if (rem > 2) s = fma_t<T>(v2, x2, s);
	}
	return s;
}

template <typename T, bool NT>
__device__ __forceinline__ T
sell_delta_slice5(const unsigned char * __restrict__ ip, const T * __restrict__ vp, int width, int lane, const T * __restrict__ x,
		int g0 = 0, int gs = 1)
{
	const int off = ld_stream<NT>(reinterpret_cast<const int *>(ip) + lane);
	const unsigned long long mask = *reinterpret_cast<const unsigned long long *>(ip + 4 * WAVE);          // uniform: a scalar load
	ip += 4 * WAVE + 16;
	if (__popcll(mask) <= 4)
		return sell_delta_slice5_body<T, NT, true>(ip, vp, width, lane, x, g0, gs, mask, off);
	return sell_delta_slice5_body<T, NT, false>(ip, vp, width, lane, x, g0, gs, mask, off);
}

// desc[2*s] = first value element of slice s, desc[2*s+1] = byte offset of its index block | mode (0 .. 5) in the low bits
template <typename T, bool NT>
__global__ __launch_bounds__(SELL_BLOCK) void
sell_delta_kernel(const int64_t * __restrict__ desc, const unsigned char * __restrict__ idx, const T * __restrict__ val,
		const int * __restrict__ row_of_sorted, const T * __restrict__ x, T * __restrict__ y,
		int m, int num_slices, int beta, XcdMap map)
{
	unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int lane = threadIdx.x % WAVE;
	const int slice = __builtin_amdgcn_readfirstlane((int) (tile * SELL_WAVES + threadIdx.x / WAVE));
	if (slice >= num_slices)
		return;
	const int64_t v_off = desc[2 * slice];
	const int64_t i_word = desc[2 * slice + 1];
	const int64_t v_next = desc[2 * slice + 2];
	const int mode = (int) (i_word & 7);
	const unsigned char * ip = idx + (i_word & ~(int64_t) 15);
	const T * vp = val + v_off + 2 * lane;
	const int groups = (int) ((v_next - v_off) / WAVE);       // = the slice's width in steps (name kept: passed as `width`)
	T s;
	if (mode == 0)
		s = sell_delta_slice<T, 0, NT>(ip, vp, groups, lane, x);
	else if (mode == 1)
		s = sell_delta_slice<T, 1, NT>(ip, vp, groups, lane, x);
	else if (mode == 2)
		s = sell_delta_slice<T, 2, NT>(ip, vp, groups, lane, x);
	else if (mode == 3)
		s = sell_delta_slice<T, 3, NT>(ip, vp, groups, lane, x);
	else if (mode == 5)
		s = sell_delta_slice5<T, NT>(ip, vp, groups, lane, x);
	else
		s = sell_delta_slice<T, 4, NT>(ip, vp, groups, lane, x);
	const long sorted_row = (long) slice * WAVE + lane;
	if (sorted_row < m)
	{
		T * yp = y + row_of_sorted[sorted_row];
		*yp = beta ? *yp + s : s;
	}
}

// Small matrices (a few thousand slices) cannot fill 256 CUs with one wave per slice: S waves share a slice, wave w takes
// the index groups w, w+S, ..., the S partial sums of a row meet in LDS and are added in wave order (deterministic;
// no longer the sequential order, so parity is to tolerance). One workgroup = 4/S slices.
template <typename T, int S, bool NT>
__global__ __launch_bounds__(SELL_BLOCK) void
sell_delta_split_kernel(const int64_t * __restrict__ desc, const unsigned char * __restrict__ idx, const T * __restrict__ val,
		const int * __restrict__ row_of_sorted, const T * __restrict__ x, T * __restrict__ y,
		int m, int num_slices, int beta, XcdMap map)
{
	constexpr int SPB = SELL_WAVES / S;      // slices per workgroup
	__shared__ T s_part[SELL_WAVES][WAVE];
	unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int lane = threadIdx.x % WAVE;
	const int wave = threadIdx.x / WAVE;
	const int w = __builtin_amdgcn_readfirstlane(wave % S);          // wave-uniform, and the compiler should know: group addresses stay scalar
	const int slice = __builtin_amdgcn_readfirstlane((int) (tile * SPB + wave / S));
	T s = 0;
	if (slice < num_slices)
	{
		const int64_t v_off = desc[2 * slice];
		const int64_t i_word = desc[2 * slice + 1];
		const int64_t v_next = desc[2 * slice + 2];
		const int mode = (int) (i_word & 7);
		const unsigned char * ip = idx + (i_word & ~(int64_t) 15);
		const T * vp = val + v_off + 2 * lane;
		const int groups = (int) ((v_next - v_off) / WAVE);       // = the slice's width in steps (name kept: passed as `width`)
		if (mode == 0)
			s = sell_delta_slice<T, 0, NT>(ip, vp, groups, lane, x, w, S);
		else if (mode == 1)
			s = sell_delta_slice<T, 1, NT>(ip, vp, groups, lane, x, w, S);
		else if (mode == 2)
			s = sell_delta_slice<T, 2, NT>(ip, vp, groups, lane, x, w, S);
		else if (mode == 3)
			s = sell_delta_slice<T, 3, NT>(ip, vp, groups, lane, x, w, S);
		else if (mode == 5)
			s = sell_delta_slice5<T, NT>(ip, vp, groups, lane, x, w, S);
		else
			s = sell_delta_slice<T, 4, NT>(ip, vp, groups, lane, x, w, S);
	}
	s_part[wave][lane] = s;
	__syncthreads();
	if (w == 0 && slice < num_slices)
	{
		T t = s_part[wave][lane];
		#pragma unroll
		for (int u = 1; u < S; u++)
			t += s_part[wave + u][lane];
		const long sorted_row = (long) slice * WAVE + lane;
		if (sorted_row < m)
		{
			T * yp = y + row_of_sorted[sorted_row];
			*yp = beta ? *yp + t : t;
		}
	}
}

template <typename T>
static int
sell_delta_launch(int S, const int64_t * desc, const unsigned char * idx, const void * val, const int * row_of_sorted, const void * x, void * y,
		int m, int num_slices, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	#define SELLD_LAUNCH(K) hipLaunchKernelGGL(K, dim3(grid), dim3(SELL_BLOCK), 0, stream, desc, idx, (const T *) val, \
			row_of_sorted, (const T *) x, (T *) y, m, num_slices, cfg.beta, cfg.map)
	if (S == 1)
	{
		if (cfg.nt) SELLD_LAUNCH((sell_delta_kernel<T, true>));
		else        SELLD_LAUNCH((sell_delta_kernel<T, false>));
	}
	else if (S == 2)
	{
		if (cfg.nt) SELLD_LAUNCH((sell_delta_split_kernel<T, 2, true>));
		else        SELLD_LAUNCH((sell_delta_split_kernel<T, 2, false>));
	}
	else if (S == 4)
	{
		if (cfg.nt) SELLD_LAUNCH((sell_delta_split_kernel<T, 4, true>));
		else        SELLD_LAUNCH((sell_delta_split_kernel<T, 4, false>));
	}
	else
	{
		set_error("sell_delta: waves per slice must be 1, 2 or 4 (got %d)", S);
		return 1;
	}
	#undef SELLD_LAUNCH
	HIP_TRY(hipGetLastError());
	return 0;
}

int
launch_sell_delta(bool f32, int waves_per_slice, const int64_t * desc, const unsigned char * idx, const void * val, const int * row_of_sorted,
		const void * x, void * y, int m, int num_slices, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? sell_delta_launch<float>(waves_per_slice, desc, idx, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out)
	           : sell_delta_launch<double>(waves_per_slice, desc, idx, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out);
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
	if (C == SELL_WIDE_C)
	{
		const unsigned grid = xcd_grid(cfg.map);
		if (grid_out)
			*grid_out = grid;
		if (grid == 0)
			return 0;
		if (cfg.nt)
			hipLaunchKernelGGL((sell_wide_kernel<T, true>), dim3(grid), dim3(SELL_WIDE_C), 0, stream, slice_ptr, col, (const T *) val, row_of_sorted,
					(const T *) x, (T *) y, m, num_slices, cfg.beta, cfg.map);
		else
			hipLaunchKernelGGL((sell_wide_kernel<T, false>), dim3(grid), dim3(SELL_WIDE_C), 0, stream, slice_ptr, col, (const T *) val, row_of_sorted,
					(const T *) x, (T *) y, m, num_slices, cfg.beta, cfg.map);
		HIP_TRY(hipGetLastError());
		return 0;
	}
	switch (C)
	{
		case 16: return sell_launch_c<T, 16>(slice_ptr, col, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out);
		case 32: return sell_launch_c<T, 32>(slice_ptr, col, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out);
		case 64: return sell_launch_c<T, 64>(slice_ptr, col, val, row_of_sorted, x, y, m, num_slices, cfg, stream, grid_out);
	}
	set_error("sell: C must be 16, 32, 64 or 256 (got %d)", C);
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
