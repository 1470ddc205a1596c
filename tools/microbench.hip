// Stand-alone probe of the streaming patterns the SpMV kernels use (not part of the product):
// how close do (8B value + 4B index) per-lane streams get to the HBM copy rate, with and without the x gather,
// for different unroll depths / load widths / cache policies? Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT, typename U> __device__ __forceinline__ U ld(const U * p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }

// MODE 0: stream only (sum of val * (double) col). MODE 1: gather x[col].
// each wave owns a contiguous "slice" of W*64 elements, lane-interleaved like SELL-64.
template <int UNROLL, bool NT, int MODE, bool PIPE>
__global__ __launch_bounds__(256) void k_sell_like(const double * __restrict__ val, const int * __restrict__ col, const double * __restrict__ x,
		double * __restrict__ y, long slices, int W)
{
	long slice = (long) blockIdx.x * 4 + threadIdx.x / 64;
	if (slice >= slices) return;
	int lane = threadIdx.x % 64;
	long p = slice * W * 64 + lane;
	long pe = p + (long) W * 64;
	double s = 0;
	if constexpr (!PIPE)
	{
		for (; p < pe; p += 64 * UNROLL)
		{
			int c[UNROLL]; double v[UNROLL];
			#pragma unroll
			for (int u = 0; u < UNROLL; u++) { c[u] = ld<NT>(col + p + 64 * u); v[u] = ld<NT>(val + p + 64 * u); }
			#pragma unroll
			for (int u = 0; u < UNROLL; u++) s = fma(v[u], MODE ? x[c[u]] : (double) c[u], s);
		}
	}
	else
	{
		int c[UNROLL]; double v[UNROLL];
		#pragma unroll
		for (int u = 0; u < UNROLL; u++) { c[u] = ld<NT>(col + p + 64 * u); v[u] = ld<NT>(val + p + 64 * u); }
		for (p += 64 * UNROLL; p < pe; p += 64 * UNROLL)
		{
			double xv[UNROLL];
			#pragma unroll
			for (int u = 0; u < UNROLL; u++) xv[u] = MODE ? x[c[u]] : (double) c[u];
			int c2[UNROLL]; double v2[UNROLL];
			#pragma unroll
			for (int u = 0; u < UNROLL; u++) { c2[u] = ld<NT>(col + p + 64 * u); v2[u] = ld<NT>(val + p + 64 * u); }
			#pragma unroll
			for (int u = 0; u < UNROLL; u++) s = fma(v[u], xv[u], s);
			#pragma unroll
			for (int u = 0; u < UNROLL; u++) { c[u] = c2[u]; v[u] = v2[u]; }
		}
		#pragma unroll
		for (int u = 0; u < UNROLL; u++) s = fma(v[u], MODE ? x[c[u]] : (double) c[u], s);
	}
	y[slice * 64 + lane] = s;
}

// 2 rows per lane: 16-byte value loads + 8-byte index loads (slice = 128 rows)
template <int UNROLL, bool NT, int MODE>
__global__ __launch_bounds__(256) void k_sell128(const d2 * __restrict__ val, const i2 * __restrict__ col, const double * __restrict__ x,
		double * __restrict__ y, long slices, int W)
{
	long slice = (long) blockIdx.x * 4 + threadIdx.x / 64;
	if (slice >= slices) return;
	int lane = threadIdx.x % 64;
	long p = slice * W * 64 + lane;
	long pe = p + (long) W * 64;
	double s0 = 0, s1 = 0;
	for (; p < pe; p += 64 * UNROLL)
	{
		i2 c[UNROLL]; d2 v[UNROLL];
		#pragma unroll
		for (int u = 0; u < UNROLL; u++) { c[u] = ld<NT>(col + p + 64 * u); v[u] = ld<NT>(val + p + 64 * u); }
		#pragma unroll
		for (int u = 0; u < UNROLL; u++)
		{
			s0 = fma(v[u].x, MODE ? x[c[u].x] : (double) c[u].x, s0);
			s1 = fma(v[u].y, MODE ? x[c[u].y] : (double) c[u].y, s1);
		}
	}
	y[slice * 128 + 2 * lane] = s0;
	y[slice * 128 + 2 * lane + 1] = s1;
}

template <typename F>
static double timeit(F f, int iters)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	f(); CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	for (int i = 0; i < iters; i++) f();
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	return ms / iters;
}

int main(int argc, char ** argv)
{
	const long rows = (argc > 1 ? atol(argv[1]) : 28L * 1000 * 1000) / 128 * 128;      // argv[1]: rows (28 per row, 12 B each)
	const int reps = argc > 2 ? atoi(argv[2]) : 10;
	const int W = 28;                       // elements per row
	const long nnz = rows * W;
	printf("rows %ld nnz %ld  matrix bytes %.2f GB\n", rows, nnz, nnz * 12 / 1e9);
	std::vector<double> hv(nnz); std::vector<int> hc(nnz);
	// SELL-64 layout: element (slice s, k, lane l) -> s*W*64 + k*64 + l ; row = s*64+l ; banded columns
	const long offs[7] = {0, 1, -1, 240, -240, 57600, -57600};
	#pragma omp parallel for
	for (long s = 0; s < rows / 64; s++)
		for (int k = 0; k < W; k++)
			for (int l = 0; l < 64; l++)
			{
				long r = s * 64 + l;
				long c = r + offs[k % 7] + (k / 7) * 3;
				if (c < 0) c = 0; if (c >= rows) c = rows - 1;
				hc[s * W * 64 + k * 64 + l] = (int) c;
				hv[s * W * 64 + k * 64 + l] = 1.0 / (1 + k);
			}
	double * dv, * dx, * dy; int * dc;
	CK(hipMalloc(&dv, nnz * 8)); CK(hipMalloc(&dc, nnz * 4)); CK(hipMalloc(&dx, rows * 8)); CK(hipMalloc(&dy, rows * 8));
	CK(hipMemcpy(dv, hv.data(), nnz * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), nnz * 4, hipMemcpyHostToDevice));
	CK(hipMemset(dx, 0, rows * 8));
	const double bytes = nnz * 12.0 + rows * 16.0;
	const long slices = rows / 64;
	const unsigned grid = (unsigned) ((slices + 3) / 4);
	#define RUN(name, ...) { timeit([&] { hipLaunchKernelGGL(__VA_ARGS__); }, reps); double ms = timeit([&] { hipLaunchKernelGGL(__VA_ARGS__); }, reps); printf("%-40s %8.3f ms %8.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); }
	RUN("stream u4",          (k_sell_like<4, false, 0, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("stream u4 nt",       (k_sell_like<4, true, 0, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("stream u7 nt",       (k_sell_like<7, true, 0, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("stream u14 nt",      (k_sell_like<14, true, 0, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("stream u28 nt",      (k_sell_like<28, true, 0, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("gather u4",          (k_sell_like<4, false, 1, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("gather u4 nt",       (k_sell_like<4, true, 1, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("gather u7 nt",       (k_sell_like<7, true, 1, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("gather u14 nt",      (k_sell_like<14, true, 1, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("gather u28 nt",      (k_sell_like<28, true, 1, false>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("gather u4 nt pipe",  (k_sell_like<4, true, 1, true>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("gather u7 nt pipe",  (k_sell_like<7, true, 1, true>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	RUN("gather u7 pipe",     (k_sell_like<7, false, 1, true>), dim3(grid), dim3(256), 0, 0, dv, dc, dx, dy, slices, W);
	const long slices2 = rows / 128;
	const unsigned grid2 = (unsigned) ((slices2 + 3) / 4);
	RUN("stream 16B u2 nt",   (k_sell128<2, true, 0>), dim3(grid2), dim3(256), 0, 0, (const d2 *) dv, (const i2 *) dc, dx, dy, slices2, W);
	RUN("stream 16B u4 nt",   (k_sell128<4, true, 0>), dim3(grid2), dim3(256), 0, 0, (const d2 *) dv, (const i2 *) dc, dx, dy, slices2, W);
	RUN("stream 16B u7 nt",   (k_sell128<7, true, 0>), dim3(grid2), dim3(256), 0, 0, (const d2 *) dv, (const i2 *) dc, dx, dy, slices2, W);
	RUN("gather 16B u4 nt",   (k_sell128<4, true, 1>), dim3(grid2), dim3(256), 0, 0, (const d2 *) dv, (const i2 *) dc, dx, dy, slices2, W);
	RUN("gather 16B u7 nt",   (k_sell128<7, true, 1>), dim3(grid2), dim3(256), 0, 0, (const d2 *) dv, (const i2 *) dc, dx, dy, slices2, W);
	RUN("gather 16B u7",      (k_sell128<7, false, 1>), dim3(grid2), dim3(256), 0, 0, (const d2 *) dv, (const i2 *) dc, dx, dy, slices2, W);
	return 0;
}
