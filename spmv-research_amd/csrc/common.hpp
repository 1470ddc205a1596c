// Shared device/host helpers of the MI355X SpMV engine (gfx950 only; wave = 64 lanes, 8 XCDs x 32 CUs).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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
// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2). With remap on, XCD k walks
// the k-th contiguous eighth of the tiles, so every L2 sees one compact window of x instead of all eight
// windows interleaved. Speed only: any placement gives the same result. The grid must be launched with
// xcd_grid(ntiles) blocks; tiles >= ntiles are skipped by the caller.
__host__ __device__ inline unsigned
xcd_tiles_per_xcd(unsigned ntiles)
{
	return (ntiles + NUM_XCD - 1) / NUM_XCD;
}

__host__ inline unsigned
xcd_grid(unsigned ntiles, bool remap)
{
	return remap ? xcd_tiles_per_xcd(ntiles) * NUM_XCD : ntiles;
}

__device__ __forceinline__ unsigned
xcd_tile(unsigned bid, unsigned ntiles, int remap)
{
	if (!remap)
		return bid;
	return (bid % NUM_XCD) * xcd_tiles_per_xcd(ntiles) + bid / NUM_XCD;
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
