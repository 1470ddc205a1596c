// How fast are fp64 global atomic adds on MI355X? Decides whether a symmetric-storage SpMV (row f4: every stored
// off-diagonal entry also scatters a*x[i] into y[col]) can beat the expanded general kernel.
//   pattern 0: lane l adds to y[(w*64 + l + it*64*17) % range]         coalesced, conflict-free (best case)
//   pattern 1: y[hash(w, l, it) % range]                               random within `range` doubles
//   pattern 2: banded: y[(w*64 + l) - (hash % bw)] clamped             what a banded lower triangle scatters
// Reports G atomics/s for range = 1 MiB .. 256 MiB of y, plus the same loops with plain stores as the ceiling.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned a)
{
	a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
	return a;
}

template <int PATTERN, bool ATOMIC>
__global__ __launch_bounds__(256) void
scatter_kernel(double * __restrict__ y, long range, int iters, long bw)
{
	const long gid = (long) blockIdx.x * 256 + threadIdx.x;
	const long total = (long) gridDim.x * 256;
	double v = 1.0 + (double) threadIdx.x * 1e-9;
	for (int it = 0; it < iters; it++)
	{
		long idx;
		if (PATTERN == 0)
			idx = (gid + (long) it * total) % range;
		else if (PATTERN == 1)
			idx = (long) (hash32((unsigned) gid * 2654435761u + (unsigned) it * 40503u) % (unsigned long) range);
		else
		{
			long center = (gid + (long) it * total) % range;
			idx = center - (long) (hash32((unsigned) gid + (unsigned) it * 7919u) % (unsigned long) bw);
			if (idx < 0)
				idx = 0;
		}
		if (ATOMIC)
			unsafeAtomicAdd(y + idx, v);
		else
			__builtin_nontemporal_store(v, y + idx);
	}
}

template <int PATTERN, bool ATOMIC>
static double
run(double * y, long range, long bw)
{
	const int iters = 64;
	const unsigned grid = 256 * 16;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((scatter_kernel<PATTERN, ATOMIC>), dim3(grid), dim3(256), 0, 0, y, range, iters, bw);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	const int reps = 5;
	for (int r = 0; r < reps; r++)
		hipLaunchKernelGGL((scatter_kernel<PATTERN, ATOMIC>), dim3(grid), dim3(256), 0, 0, y, range, iters, bw);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	return (double) grid * 256 * iters * reps / (ms * 1e-3) / 1e9;
}

int
main()
{
	const long max_range = 32L << 20;             // doubles: 256 MiB
	double * y;
	CHECK(hipMalloc(&y, max_range * 8));
	CHECK(hipMemset(y, 0, max_range * 8));
	printf("%-10s %12s %12s %12s | %12s %12s %12s   (G ops/s)\n", "y range", "atomic seq", "atomic rand", "atomic band", "store seq", "store rand", "store band");
	for (long mb : {1L, 8L, 32L, 128L, 256L})
	{
		const long range = mb << 17;
		const long bw = 65536;
		printf("%6ld MiB %12.1f %12.1f %12.1f | %12.1f %12.1f %12.1f\n", mb,
				run<0, true>(y, range, bw), run<1, true>(y, range, bw), run<2, true>(y, range, bw),
				run<0, false>(y, range, bw), run<1, false>(y, range, bw), run<2, false>(y, range, bw));
		fflush(stdout);
	}
	return 0;
}
