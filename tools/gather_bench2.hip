// Stand-alone probe #2 (not part of the product): the two costs of coo_blocked_kernel and whether they overlap.
// Launch shape of the real kernel (256 workgroups x 1024 threads, one per CU, barrier per column block, one dword per entry =
// u16 column in block | u16 LDS slot). Statistics of the soc-LiveJournal1 twin: 0.44 distinct 128-byte lines of x per entry.
//   gather        x gathers only (sum in registers)
//   add           LDS ds_add_f64 only
//   serial        gathers of block b, then its adds, barrier              (the round-2 kernel's order)
//   pipelined     gathers of block b+1 issued BEFORE the adds of block b  (the LDS pipeline works while the gathers fly)
// slot order inside one wave instruction (free for the layout builder: the 64 entries of an instruction can be permuted at will):
//   random / conflict-free within groups of 16 lanes / within 32 lanes (bank pair = slot % 32)
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_bench2.hip -o tools/bin/gather_bench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int THREADS = 1024;
constexpr int WGS = 256;
constexpr int SLOTS = 19200;

enum { M_GATHER = 0, M_ADD, M_SERIAL, M_PIPE, M_ADD_F32, M_STORE };

template <int MODE, int K>
__global__ __launch_bounds__(THREADS) void
k(const unsigned * __restrict__ ent, const double * __restrict__ x, double * __restrict__ out, int nblocks, int block_cols)
{
	extern __shared__ __align__(16) unsigned char smem[];
	double * ys = reinterpret_cast<double *>(smem);
	float * yf = reinterpret_cast<float *>(smem);
	constexpr int E = K * THREADS;
	const int tid = threadIdx.x;
	for (int l = tid; l < SLOTS; l += THREADS)
		ys[l] = 0;
	__syncthreads();
	const int xcd = blockIdx.x % 8;
	const unsigned * my = ent + (size_t) blockIdx.x * nblocks * E;
	double acc = 0;
	unsigned c0[K], c1[K], c2[K];
	double xv[K], xn[K];
	auto xblk = [&](int b) { return x + (size_t) ((b + xcd * 19) % nblocks) * block_cols; };
	#pragma unroll
	for (int u = 0; u < K; u++)
	{
		c0[u] = __builtin_nontemporal_load(my + u * THREADS + tid);
		c1[u] = __builtin_nontemporal_load(my + (size_t) (nblocks > 1 ? 1 : 0) * E + u * THREADS + tid);
	}
	if constexpr (MODE == M_PIPE)
	{
		#pragma unroll
		for (int u = 0; u < K; u++)
			xv[u] = xblk(0)[c0[u] >> 16];
	}
	if constexpr (MODE == M_PIPE)
	{
		// hand-rotated: 3 entry buffers x 2 gather buffers = period 6 (nblocks is a multiple of 6), no register copies between steps
		#define PSTEP(CA, CB, CC, XA, XB, bb)                                                                        \
		{                                                                                                            \
			const double * xb1 = xblk((bb) + 1 < nblocks ? (bb) + 1 : (bb));                                      \
			_Pragma("unroll") for (int u = 0; u < K; u++) XB[u] = xb1[CB[u] >> 16];                               \
			const int b2 = (bb) + 2 < nblocks ? (bb) + 2 : nblocks - 1;                                           \
			_Pragma("unroll") for (int u = 0; u < K; u++) CC[u] = __builtin_nontemporal_load(my + (size_t) b2 * E + u * THREADS + tid); \
			_Pragma("unroll") for (int u = 0; u < K; u++) unsafeAtomicAdd(&ys[CA[u] & 0xffffu], XA[u]);          \
			__syncthreads();                                                                                      \
		}
		for (int b = 0; b < nblocks; b += 6)
		{
			PSTEP(c0, c1, c2, xv, xn, b)
			PSTEP(c1, c2, c0, xn, xv, b + 1)
			PSTEP(c2, c0, c1, xv, xn, b + 2)
			PSTEP(c0, c1, c2, xn, xv, b + 3)
			PSTEP(c1, c2, c0, xv, xn, b + 4)
			PSTEP(c2, c0, c1, xn, xv, b + 5)
		}
	}
	else
	for (int b = 0; b < nblocks; b++)
	{
		const int b2 = b + 2 < nblocks ? b + 2 : nblocks - 1;
		{
			const double * xb = xblk(b);
			if constexpr (MODE == M_GATHER || MODE == M_SERIAL)
			{
				#pragma unroll
				for (int u = 0; u < K; u++)
					xv[u] = xb[c0[u] >> 16];
			}
			else
			{
				#pragma unroll
				for (int u = 0; u < K; u++)
					xv[u] = (double) (c0[u] >> 16);
			}
			#pragma unroll
			for (int u = 0; u < K; u++)
				c2[u] = __builtin_nontemporal_load(my + (size_t) b2 * E + u * THREADS + tid);
			if constexpr (MODE == M_GATHER)
			{
				#pragma unroll
				for (int u = 0; u < K; u++)
					acc += xv[u];
			}
			else if constexpr (MODE == M_ADD_F32)
			{
				#pragma unroll
				for (int u = 0; u < K; u++)
					unsafeAtomicAdd(&yf[c0[u] & 0xffffu], (float) xv[u]);
			}
			else if constexpr (MODE == M_STORE)
			{
				#pragma unroll
				for (int u = 0; u < K; u++)
					ys[c0[u] & 0xffffu] = xv[u];
			}
			else
			{
				#pragma unroll
				for (int u = 0; u < K; u++)
					unsafeAtomicAdd(&ys[c0[u] & 0xffffu], xv[u]);
			}
		}
		#pragma unroll
		for (int u = 0; u < K; u++)
		{
			c0[u] = c1[u];
			c1[u] = c2[u];
		}
		__syncthreads();
	}
	for (int l = tid; l < SLOTS; l += THREADS)
		acc += ys[l];
	out[(size_t) blockIdx.x * THREADS + tid] = acc;
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

// permute each aligned group of 64 entries so that inside every group of G lanes the bank pairs (slot % 32) are distinct where possible
static void
order_conflict_free(unsigned * e, int G)
{
	std::vector<unsigned> pool(e, e + 64), outv;
	outv.reserve(64);
	for (int g0 = 0; g0 < 64; g0 += G)
	{
		bool used[32] = {false};
		std::vector<unsigned> grp;
		for (size_t i = 0; i < pool.size() && (int) grp.size() < G;)
		{
			const int bank = (pool[i] & 0xffffu) % 32;
			if (!used[bank])
			{
				used[bank] = true;
				grp.push_back(pool[i]);
				pool.erase(pool.begin() + i);
			}
			else
				i++;
		}
		while ((int) grp.size() < G && !pool.empty())
		{
			grp.push_back(pool.back());
			pool.pop_back();
		}
		outv.insert(outv.end(), grp.begin(), grp.end());
	}
	std::copy(outv.begin(), outv.end(), e);
}

template <int K>
static void
run_all(int nblocks, int block_cols)
{
	constexpr int E = K * THREADS;
	const size_t total = (size_t) nblocks * E * WGS;
	printf("K = %d entries per lane and block, %d blocks of %d columns, %.1f M entries per launch\n", K, nblocks, block_cols, total / 1e6);
	double * dx, * dout;
	unsigned * dent;
	CK(hipMalloc(&dx, (size_t) nblocks * block_cols * 8 + 4096));
	CK(hipMemset(dx, 0, (size_t) nblocks * block_cols * 8 + 4096));
	CK(hipMalloc(&dout, (size_t) WGS * THREADS * 8));
	CK(hipMalloc(&dent, total * 4 + 4096));
	std::vector<unsigned> h(total);
	const int lds = SLOTS * 8;
	#define GRANT(M) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<M, K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds))
	GRANT(M_GATHER); GRANT(M_ADD); GRANT(M_SERIAL); GRANT(M_PIPE); GRANT(M_ADD_F32); GRANT(M_STORE);
	#define RUN(name, M) { double ms = timeit([&] { hipLaunchKernelGGL((k<M, K>), dim3(WGS), dim3(THREADS), lds, 0, dent, dx, dout, nblocks, block_cols); }, 10); \
		printf("    %-28s %8.1f us  %6.3f clk per entry and CU @2.1GHz  (69.04 M entries: %6.1f us)\n", name, ms * 1e3, ms * 1e-3 * 2.1e9 / (total / 256.0), ms * 1e3 * 69.04e6 / total); fflush(stdout); }
	for (int order : {0, 16, 32})
	{
		#pragma omp parallel for
		for (long wb = 0; wb < (long) WGS * nblocks; wb++)
		{
			std::mt19937_64 g(wb * 7919 + 1);
			std::vector<unsigned> cols(E);
			for (int e = 0; e < E; e++)
				cols[e] = g() % block_cols;
			std::sort(cols.begin(), cols.end());
			for (int e = 0; e < E; e++)
				h[wb * E + e] = (cols[e] << 16) | (unsigned) (g() % SLOTS);
			if (order)
				for (int u = 0; u < K; u++)
					for (int w = 0; w < THREADS / 64; w++)
						order_conflict_free(&h[wb * E + u * THREADS + w * 64], order);
		}
		CK(hipMemcpy(dent, h.data(), total * 4, hipMemcpyHostToDevice));
		printf("  slot order inside a wave instruction: %s\n", order == 0 ? "random" : order == 16 ? "distinct bank pairs inside 16 lanes" : "distinct bank pairs inside 32 lanes");
		if (order == 0)
			RUN("gather only", M_GATHER);
		RUN("ds_add_f64 only", M_ADD);
		if (order == 0)
		{
			RUN("ds_add_f32 only", M_ADD_F32);
			RUN("ds_write_b64 only", M_STORE);
		}
		RUN("serial: gather, add", M_SERIAL);
		RUN("pipelined: gather b+1, add b", M_PIPE);
	}
	CK(hipFree(dx)); CK(hipFree(dout)); CK(hipFree(dent));
}

int
main()
{
	// 4.85 M columns: 148 blocks of 32768 columns = 2048 lines of 128 B; 4096 random entries per workgroup and block touch
	// (1 - exp(-2)) * 2048 = 1771 lines = 0.43 lines per entry (the twin's layout: 0.444)
	run_all<4>(150, 32768);
	run_all<8>(78, 65536);
	return 0;
}
