// Shared device/host helpers of the MI355X SpMV engine (gfx950 only; wave = 64 lanes, 8 XCDs x 32 CUs).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "host_threads.hpp"

namespace spmv {

// ------------------------------------------------------------------------------------------------ errors
void set_error(const char * fmt, ...);

#define HIP_TRY(expr)                                                                                     \
	do {                                                                                              \
		hipError_t e_ = (expr);                                                                   \
		if (e_ != hipSuccess)                                                                     \
		{                                                                                         \
			::spmv::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return 1;                                                                         \
		}                                                                                         \
	} while (0)

constexpr int WAVE = 64;
constexpr int NUM_XCD = 8;

// --------------------------------------------------------------------------------------- XCD-aware tiling
// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2; which XCD is not defined, only the
// grouping). With the map on, XCD group k walks the contiguous tile range [start[k], start[k+1]), so every L2 sees one
// compact window of x instead of eight interleaved ones. The ranges are balanced by WORK (non-zeros), not by tile
// count: contiguous eighths of a matrix whose halves differ 3x in row length left half of the chip idle (measured on
// the nlpkkt240 twin: 4.3 -> 5.0 TB/s from balance alone). Speed only: any placement gives the same result.
struct XcdMap {
	unsigned start[NUM_XCD + 1];   // remap == 1: tile range per XCD group (balanced by work)
	unsigned ntiles;
	unsigned remap;                // 0 = identity, 1 = contiguous work-balanced ranges, 2 = chunks of `chunk` tiles dealt
	                               //     round-robin to the XCD groups (locality inside a chunk, balance by statistics)
	unsigned chunk;                // tiles per chunk in mode 2 (power of two)
};

// measured on the nlpkkt240 twin (SELL delta kernel): chunks of 32..128 tiles within 1 %, 8 tiles -17 %, >= 512 tiles -5..-13 %
constexpr unsigned XCD_CHUNK_DEFAULT = 64;

constexpr unsigned NO_TILE = 0xffffffffu;

// tiles per chunk of the chunked order: SPMV_MI355X_XCD_CHUNK (a power of two) or the default; read at every create()
inline unsigned
xcd_chunk_setting()
{
	if (const char * e = getenv("SPMV_MI355X_XCD_CHUNK"))
	{
		const long v = atol(e);
		if (v >= 1 && v <= (1 << 20) && (v & (v - 1)) == 0)
			return (unsigned) v;
	}
	return XCD_CHUNK_DEFAULT;
}

__device__ __forceinline__ unsigned
xcd_tile(unsigned bid, const XcdMap & mp)
{
	if (!mp.remap)
		return bid < mp.ntiles ? bid : NO_TILE;
	const unsigned k = bid % NUM_XCD;
	const unsigned i = bid / NUM_XCD;          // i-th block of this XCD group
	if (mp.remap == 2)
	{
		const unsigned t = ((i / mp.chunk) * NUM_XCD + k) * mp.chunk + i % mp.chunk;
		return t < mp.ntiles ? t : NO_TILE;
	}
	const unsigned t = mp.start[k] + i;
	return t < mp.start[k + 1] ? t : NO_TILE;
}

// blocks to launch for a map
inline unsigned
xcd_grid(const XcdMap & mp)
{
	if (!mp.remap)
		return mp.ntiles;
	if (mp.remap == 2)
	{
		const unsigned span = NUM_XCD * mp.chunk;
		return (mp.ntiles + span - 1) / span * span;
	}
	unsigned mx = 0;
	for (int k = 0; k < NUM_XCD; k++)
		mx = mp.start[k + 1] - mp.start[k] > mx ? mp.start[k + 1] - mp.start[k] : mx;
	return mx * NUM_XCD;
}

// equal tile counts (tiles of equal work: merge path, COO)
inline XcdMap
xcd_map_uniform(unsigned ntiles, int remap)
{
	XcdMap mp;
	mp.ntiles = ntiles;
	mp.remap = (unsigned) remap;
	mp.chunk = xcd_chunk_setting();
	for (int k = 0; k <= NUM_XCD; k++)
		mp.start[k] = (unsigned) ((unsigned long long) ntiles * k / NUM_XCD);
	return mp;
}

// tiles of `units_per_tile` consecutive units (rows, slices) whose cumulative work is prefix[unit] (prefix has
// num_units+1 entries, e.g. row_ptr): boundary k = first tile whose starting prefix reaches k/8 of the total
template <typename P>
inline XcdMap
xcd_map_balanced(const P * prefix, long num_units, long units_per_tile, int remap)
{
	XcdMap mp;
	const long ntiles = (num_units + units_per_tile - 1) / units_per_tile;
	mp.ntiles = (unsigned) ntiles;
	mp.remap = (unsigned) remap;
	mp.chunk = xcd_chunk_setting();
	const double total = (double) (prefix[num_units] - prefix[0]);
	mp.start[0] = 0;
	for (int k = 1; k < NUM_XCD; k++)
	{
		const double target = total * k / NUM_XCD;
		long lo = 0, hi = ntiles;          // first tile t with work before it >= target
		while (lo < hi)
		{
			long mid = (lo + hi) / 2;
			long u = mid * units_per_tile;
			if (u > num_units)
				u = num_units;
			if ((double) (prefix[u] - prefix[0]) >= target)
				hi = mid;
			else
				lo = mid + 1;
		}
		mp.start[k] = (unsigned) lo;
	}
	mp.start[NUM_XCD] = (unsigned) ntiles;
	for (int k = 1; k <= NUM_XCD; k++)
		if (mp.start[k] < mp.start[k - 1])
			mp.start[k] = mp.start[k - 1];
	return mp;
}

// ------------------------------------------------------------------------------------------------- loads
// Streamed-once matrix data (values / column indices) can bypass the cache-retention policy so that the reused
// x vector keeps its lines in L2 / Infinity Cache. NT is a compile-time switch chosen from the footprint.
template <bool NT, typename U>
__device__ __forceinline__ U
ld_stream(const U * p)
{
	if constexpr (NT)
		return __builtin_nontemporal_load(p);
	else
		return *p;
}

// --------------------------------------------------------------------------------------- wave primitives
template <typename T>
__device__ __forceinline__ T
shfl_xor_t(T v, int lane_mask)
{
	return __shfl_xor(v, lane_mask, WAVE);
}

template <typename T>
__device__ __forceinline__ T
shfl_up_t(T v, unsigned delta)
{
	return __shfl_up(v, delta, WAVE);
}

// Sum over aligned groups of G consecutive lanes (G power of two <= 64): every lane of a group ends with the
// group total. Fixed butterfly order -> results are reproducible run to run.
template <typename T, int G>
__device__ __forceinline__ T
group_reduce_sum(T v)
{
	#pragma unroll
	for (int off = G / 2; off >= 1; off >>= 1)
		v += shfl_xor_t(v, off);
	return v;
}

template <typename T>
__device__ __forceinline__ T
fma_t(T a, T b, T c);
template <>
__device__ __forceinline__ double
fma_t<double>(double a, double b, double c)
{
	return fma(a, b, c);
}
template <>
__device__ __forceinline__ float
fma_t<float>(float a, float b, float c)
{
	return fmaf(a, b, c);
}

}  // namespace spmv
