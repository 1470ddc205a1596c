// Stand-alone probe (not part of the product): what bounds the x gathers of the column-blocked layout (kernels_coo.hip)?
// Same launch shape as coo_blocked_kernel: 256 workgroups of 1024 threads, one per CU (LDS request > 80 KiB), the 32 workgroups of
// an XCD sweep the same blocks of x in step (barrier per block), K entries per lane and block, u16 column inside the block.
//   pattern "share s": groups of s consecutive entries fall into ONE 64-byte sector of x (different words) -> s = 1 is the random
//   twin, s = 2, 4, 8 say whether the bound is per LANE (no change) or per SECTOR / L1 fill (time ~ 1/s);
//   widths: 4 / 8 / 16 bytes per lane; sc1 = L1 bypass; scalar = s_load gathers (the scalar data cache's own path to L2);
//   mix = vector gathers with every 4th batch taken by the scalar path; lds = + ds_add_f64 into a random LDS slot (the real kernel).
// Second part: write an 8-byte stream of S MB with one kernel, read it back with the next (propagation blocking's intermediate):
// does the Infinity Cache absorb it?
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o tools/bin/gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int THREADS = 1024;
constexpr int WGS = 256;
constexpr int K = 4;                 // entries per lane and block
constexpr int E = K * THREADS;       // entries per workgroup and block

typedef double d2 __attribute__((ext_vector_type(2)));

enum { M_V8 = 0, M_V8_LDS, M_V4, M_V16, M_V8_SC1, M_SCALAR, M_MIX, M_V8_NOBAR, M_SCALAR_LDS };

template <int MODE>
__global__ __launch_bounds__(THREADS) void
gather_k(const unsigned short * __restrict__ idx, const unsigned short * __restrict__ slot, const double * __restrict__ x,
		double * __restrict__ out, int nblocks, int block_cols)
{
	extern __shared__ __align__(16) unsigned char smem[];
	double * ys = reinterpret_cast<double *>(smem);
	const int tid = threadIdx.x;
	const int lane = tid % 64;
	if (MODE == M_V8_LDS || MODE == M_SCALAR_LDS)
	{
		for (int l = tid; l < 10240; l += THREADS)
			ys[l] = 0;
		__syncthreads();
	}
	const int xcd = blockIdx.x % 8;
	const unsigned short * my = idx + (size_t) blockIdx.x * nblocks * E;
	const unsigned short * ms = slot + (size_t) blockIdx.x * nblocks * E;
	double acc = 0;
	unsigned short c[K], nc[K];
	#pragma unroll
	for (int u = 0; u < K; u++)
		c[u] = my[u * THREADS + tid];
	for (int b = 0; b < nblocks; b++)
	{
		// every XCD starts its sweep somewhere else (the real ranges walk different windows of x)
		const double * xb = x + (size_t) ((b + xcd * 11) % nblocks) * block_cols;
		const int nb = b + 1 < nblocks ? b + 1 : b;
		if constexpr (MODE == M_SCALAR || MODE == M_SCALAR_LDS)
		{
			// the wave's 64 * K entries one by one through the scalar data cache: column from a lane, uniform address, s_load
			#pragma unroll
			for (int u = 0; u < K; u++)
			{
				const int cv = c[u];
				#pragma unroll
				for (int l0 = 0; l0 < 64; l0 += 16)
				{
					double t[16];
					#pragma unroll
					for (int l = 0; l < 16; l++)
						t[l] = xb[__builtin_amdgcn_readlane(cv, l0 + l)];
					if constexpr (MODE == M_SCALAR)
					{
						#pragma unroll
						for (int l = 0; l < 16; l++)
							acc += t[l];
					}
					else
					{
						// hand the 16 scalars to 16 lanes and add them into LDS with one vector instruction
						double mine = 0;
						#pragma unroll
						for (int l = 0; l < 16; l++)
							mine = (lane == l0 + l) ? t[l] : mine;
						if (lane >= l0 && lane < l0 + 16)
							unsafeAtomicAdd(&ys[ms[(size_t) b * E + u * THREADS + tid] % 10240], mine);
					}
				}
			}
		}
		else
		{
			double xv[K];
			#pragma unroll
			for (int u = 0; u < K; u++)
			{
				if constexpr (MODE == M_V4)
					xv[u] = reinterpret_cast<const float *>(xb)[c[u]];
				else if constexpr (MODE == M_V16)
				{
					d2 t = reinterpret_cast<const d2 *>(xb)[c[u] >> 1];
					xv[u] = t.x + t.y;
				}
				else if constexpr (MODE == M_V8_SC1)
					xv[u] = __hip_atomic_load(xb + c[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				else if constexpr (MODE == M_MIX)
				{
					if (u == K - 1)
					{
						double mine = 0;
						const int cv = c[u];
						#pragma unroll
						for (int l0 = 0; l0 < 64; l0 += 16)
						{
							double t[16];
							#pragma unroll
							for (int l = 0; l < 16; l++)
								t[l] = xb[__builtin_amdgcn_readlane(cv, l0 + l)];
							#pragma unroll
							for (int l = 0; l < 16; l++)
								mine = (lane == l0 + l) ? t[l] : mine;
						}
						xv[u] = mine;
					}
					else
						xv[u] = xb[c[u]];
				}
				else
					xv[u] = xb[c[u]];
			}
			#pragma unroll
			for (int u = 0; u < K; u++)
				nc[u] = my[(size_t) nb * E + u * THREADS + tid];
			if constexpr (MODE == M_V8_LDS)
			{
				#pragma unroll
				for (int u = 0; u < K; u++)
					unsafeAtomicAdd(&ys[ms[(size_t) b * E + u * THREADS + tid] % 10240], xv[u]);
			}
			else
			{
				#pragma unroll
				for (int u = 0; u < K; u++)
					acc += xv[u];
			}
		}
		if constexpr (MODE == M_SCALAR || MODE == M_SCALAR_LDS)
		{
			#pragma unroll
			for (int u = 0; u < K; u++)
				nc[u] = my[(size_t) nb * E + u * THREADS + tid];
		}
		#pragma unroll
		for (int u = 0; u < K; u++)
			c[u] = nc[u];
		if constexpr (MODE != M_V8_NOBAR)
			__syncthreads();
	}
	if (MODE == M_V8_LDS || MODE == M_SCALAR_LDS)
	{
		__syncthreads();
		for (int l = tid; l < 10240; l += THREADS)
			acc += ys[l];
	}
	out[(size_t) blockIdx.x * THREADS + tid] = acc;
}

__global__ __launch_bounds__(256) void
stream_write_k(d2 * __restrict__ p, long n2, double v)
{
	for (long i = (long) blockIdx.x * 256 + threadIdx.x; i < n2; i += (long) gridDim.x * 256)
		p[i] = d2{v, v + 1};
}

__global__ __launch_bounds__(256) void
stream_read_k(const d2 * __restrict__ p, long n2, double * __restrict__ out)
{
	double acc = 0;
	for (long i = (long) blockIdx.x * 256 + threadIdx.x; i < n2; i += (long) gridDim.x * 256)
	{
		d2 t = p[i];
		acc += t.x + t.y;
	}
	if (acc == 12345.678)
		out[0] = acc;
}

template <typename F>
static double
timeit(F f, int iters)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	f();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	for (int i = 0; i < iters; i++)
		f();
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	return ms / iters;
}

int
main(int argc, char ** argv)
{
	const int nblocks = argc > 1 ? atoi(argv[1]) : 82;
	const int block_cols = 59136;                       // 4.85 M columns / 82, a multiple of 64
	const int reps = 20;
	const size_t per_wg = (size_t) nblocks * E;
	const size_t total = per_wg * WGS;
	printf("gather bench: %d workgroups x %d threads, %d blocks of %d columns (%.0f KiB), %d entries per lane and block: %.1f M gathers per launch\n",
			WGS, THREADS, nblocks, block_cols, block_cols * 8 / 1024.0, K, total / 1e6);
	double * dx, * dout;
	unsigned short * didx, * dslot;
	CK(hipMalloc(&dx, (size_t) nblocks * block_cols * 8 + 4096));
	CK(hipMemset(dx, 0, (size_t) nblocks * block_cols * 8 + 4096));
	CK(hipMalloc(&dout, (size_t) WGS * THREADS * 8));
	CK(hipMalloc(&didx, total * 2 + 4096));
	CK(hipMalloc(&dslot, total * 2 + 4096));
	std::vector<unsigned short> h(total), hs(total);
	const int lds = 84 * 1024;
	#define GRANT(M) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_k<M>), hipFuncAttributeMaxDynamicSharedMemorySize, lds))
	GRANT(M_V8); GRANT(M_V8_LDS); GRANT(M_V4); GRANT(M_V16); GRANT(M_V8_SC1); GRANT(M_SCALAR); GRANT(M_MIX); GRANT(M_V8_NOBAR); GRANT(M_SCALAR_LDS);
	#define RUN(name, M) { double ms = timeit([&] { hipLaunchKernelGGL((gather_k<M>), dim3(WGS), dim3(THREADS), lds, 0, didx, dslot, dx, dout, nblocks, block_cols); }, reps); \
		printf("  %-34s %8.1f us  %7.1f G lanes/s  %6.3f lanes/clk/CU @2.1GHz\n", name, ms * 1e3, total / ms / 1e6, total / 256.0 / (ms * 1e-3 * 2.1e9)); fflush(stdout); }
	for (int s : {1, 2, 4, 8, 64})
	{
		// s consecutive entries share one 64-byte sector (s <= 8), s = 64: a whole wave instruction reads 8 adjacent sectors
		#pragma omp parallel for
		for (long wb = 0; wb < (long) WGS * nblocks; wb++)
		{
			std::mt19937_64 g(wb * 7919 + s);
			std::vector<unsigned> cols(E);
			const int sectors = block_cols / 8;
			if (s <= 8)
			{
				for (int e = 0; e < E; e += s)
				{
					unsigned sec = g() % sectors;
					for (int q = 0; q < s; q++)
						cols[e + q] = sec * 8 + q;
				}
			}
			else
			{
				for (int e = 0; e < E; e += 64)
				{
					unsigned sec = g() % (sectors - 8);
					for (int q = 0; q < 64; q++)
						cols[e + q] = sec * 8 + q;
				}
			}
			std::sort(cols.begin(), cols.end());
			for (int e = 0; e < E; e++)
			{
				h[wb * E + e] = (unsigned short) cols[e];
				hs[wb * E + e] = (unsigned short) (g() % 10240);
			}
		}
		CK(hipMemcpy(didx, h.data(), total * 2, hipMemcpyHostToDevice));
		CK(hipMemcpy(dslot, hs.data(), total * 2, hipMemcpyHostToDevice));
		printf("pattern: %d sorted entries share a 64-byte sector\n", s);
		RUN("vector 8 B", M_V8);
		RUN("vector 8 B, no barrier", M_V8_NOBAR);
		RUN("vector 8 B + ds_add_f64", M_V8_LDS);
		RUN("vector 4 B", M_V4);
		RUN("vector 16 B", M_V16);
		RUN("vector 8 B sc1 (L1 bypass)", M_V8_SC1);
		if (s == 1 || s == 8)
		{
			RUN("scalar s_load 8 B", M_SCALAR);
			RUN("scalar s_load 8 B + ds_add_f64", M_SCALAR_LDS);
			RUN("mix: 3 vector + 1 scalar batch", M_MIX);
		}
	}
	// ---- an 8-byte stream written by one kernel and read by the next
	printf("write-then-read of an 8-byte stream (two kernels back to back):\n");
	for (long mb : {32L, 64L, 128L, 192L, 256L, 512L, 1024L})
	{
		d2 * p;
		const long n2 = mb * 1024 * 1024 / 16;
		CK(hipMalloc(&p, (size_t) n2 * 16));
		double tw = timeit([&] { hipLaunchKernelGGL(stream_write_k, dim3(2048), dim3(256), 0, 0, p, n2, 1.0); }, 10);
		double tr = timeit([&] { hipLaunchKernelGGL(stream_read_k, dim3(2048), dim3(256), 0, 0, p, n2, dout); }, 10);
		double tp = timeit([&] { hipLaunchKernelGGL(stream_write_k, dim3(2048), dim3(256), 0, 0, p, n2, 1.0);
		                         hipLaunchKernelGGL(stream_read_k, dim3(2048), dim3(256), 0, 0, p, n2, dout); }, 10);
		printf("  %5ld MB: write %7.1f us (%5.2f TB/s)  read %7.1f us (%5.2f TB/s)  write+read %7.1f us (%5.2f TB/s over both)\n", mb,
				tw * 1e3, mb * 1.048576e6 / tw / 1e9, tr * 1e3, mb * 1.048576e6 / tr / 1e9, tp * 1e3, 2 * mb * 1.048576e6 / tp / 1e9);
		CK(hipFree(p));
	}
	return 0;
}
