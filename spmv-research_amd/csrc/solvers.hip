// Device-resident Krylov callers of the SpMV path: Jacobi-preconditioned CG and BiCGSTAB.
//
// What they replace in the reference (benchmark_code/BENCH/src):
//   spmv_mi355x_pcg        preconditioned_cg()        bench_cg.cpp:93-322
//   spmv_mi355x_pbicgstab  preconditioned_bicgstab()  bench_bicg.cpp:149-459
// The reference keeps every vector on the host and calls MF->spmv(host x, host y) once or twice per iteration, so a GPU
// backend pays an upload of x and a download of y around every launch. Here all vectors live in HBM and
// the loop never waits for the host:
//   * the scalars (alpha, beta, omega, the error norms, the loop counter) live in a small device struct; the kernels of
//     iteration k read state[k&1] and block 0 of the last kernel writes state[(k+1)&1] (no read/write race, no sync);
//   * dots are two-stage: every block writes one partial per quantity, and every block of the CONSUMING kernel re-reduces
//     the <= 1024 partials in a fixed order — deterministic, and no separate "finish the reduction" launch;
//   * the `err < eps` break (bench_cg.cpp:238) becomes a device flag that predicates every later vector kernel off, so
//     x, x_best, the counter and the history are frozen exactly where the reference breaks; the last kernel of every
//     iteration also posts (iterations finished, done) into host-mapped pinned memory, which the host READS (no HIP call:
//     hipEventSynchronize in this loop was measured to stall the queue for up to 100 ms at a time) to stop enqueueing
//     after a break and to stay at most 2*POLL iterations ahead of the device;
//   * the vector updates around the SpMV are fused: CG = SpMV + 3 passes (p.Ap | x,r update + z.r, r.r | p update),
//     BiCGSTAB = 2 SpMV + 5 passes. z = r/K and h = x + s_a*y are never materialised.
// Same iteration semantics as the reference: Jacobi K = first stored diagonal entry (error on a zero), x0 = 0,
// eps = 1e-15*|b|, explicit residual every 100 iterations with x_best tracking and (CG only) the restart rule.
// Dot products accumulate in double for both precisions (the reference accumulates in ValueType with an OpenMP
// thread-count-dependent order, so its last bits are not reproducible either).

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "../../include/spmv_mi355x.h"

namespace spmv {


constexpr int VB = 256;           // threads per block of the vector kernels
constexpr int MAX_PART = 1024;    // partial sums per quantity
constexpr int RESTART_K = 100;    // bench_cg.cpp:178
constexpr int POLL = 32;          // the host looks at the progress word every POLL iterations

struct SolverState {
	double err, err_explicit, err_best, eps, eps_counter;
	double zr;                    // CG: (z, r)          BiCGSTAB: s_pk_p = (r0_, rk)
	long k;                       // completed loop bodies = num_loops_out
	long restarts;
	int done;                     // the `err < eps` break was reached
	int pad;
};

enum { P_A = 0, P_B, P_C, P_D, P_E, P_F, NUM_SLOTS };

__device__ __forceinline__ double
block_sum(double v)
{
	__shared__ double sh[VB / WAVE];
	__shared__ double total;
	for (int o = WAVE / 2; o > 0; o >>= 1)
		v += __shfl_down(v, o, WAVE);
	__syncthreads();                       // protects sh/total against the previous call
	if (threadIdx.x % WAVE == 0)
		sh[threadIdx.x / WAVE] = v;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		double s = 0;
		for (int w = 0; w < VB / WAVE; w++)
			s += sh[w];
		total = s;
	}
	__syncthreads();
	return total;
}

// Every block reduces the nb partials of one slot in the same order: all blocks get the same bits.
__device__ __forceinline__ double
sum_partials(const double * __restrict__ part, int slot, int nb)
{
	const double * p = part + (long) slot * MAX_PART;
	double v = 0;
	for (int i = threadIdx.x; i < nb; i += VB)
		v += p[i];
	return block_sum(v);
}

__device__ __forceinline__ void
store_partial(double * __restrict__ part, int slot, double v)
{
	v = block_sum(v);
	if (threadIdx.x == 0)
		part[(long) slot * MAX_PART + blockIdx.x] = v;
}

// (iterations the device has finished, loop count at the break or -1) for the host, in host-mapped pinned memory
__device__ __forceinline__ void
post_progress(volatile long * host_progress, long finished, long broke_at)
{
	host_progress[1] = broke_at;
	__threadfence_system();
	host_progress[0] = finished;
	__threadfence_system();
}

// Distributed solves: the per-block partials of up to 3 slots are summed into red[], all-reduced over the ranks by the
// caller's collective, and written back as the single partial of their slot (consumers then run with nb = 1).
struct SlotList {
	int n;
	int s[3];
};

__global__ __launch_bounds__(VB) void
reduce_slots_kernel(const double * __restrict__ part, int nb, SlotList sl, double * __restrict__ red)
{
	const double v = sum_partials(part, sl.s[blockIdx.x], nb);
	if (threadIdx.x == 0)
		red[blockIdx.x] = v;
}

__global__ void
scatter_slots_kernel(double * __restrict__ part, SlotList sl, const double * __restrict__ red)
{
	if ((int) threadIdx.x < sl.n)
		part[(long) sl.s[threadIdx.x] * MAX_PART] = red[threadIdx.x];
}

#define GRID_STRIDE(i, m) for (long i = (long) blockIdx.x * VB + threadIdx.x; i < (m); i += (long) gridDim.x * VB)

// ------------------------------------------------------------------------------------------------ shared kernels

// r = b - Ax ; partials: A = r.r, B = b.b     (bench_cg.cpp:146-150,163-166)
template <typename T>
__global__ __launch_bounds__(VB) void
residual_kernel(const T * __restrict__ b, const T * __restrict__ Ax, T * __restrict__ r, long m, double * __restrict__ part)
{
	double rr = 0, bb = 0;
	GRID_STRIDE(i, m)
	{
		const T bi = b[i];
		const T ri = bi + (T) -1 * Ax[i];
		r[i] = ri;
		rr += (double) ri * (double) ri;
		bb += (double) bi * (double) bi;
	}
	store_partial(part, P_A, rr);
	store_partial(part, P_B, bb);
}

// Explicit-residual bookkeeping (bench_cg.cpp:186-236, bench_bicg.cpp:277-302): partial A holds |b - A x|^2.
// promote: x_best = x when err_explicit < err_best. restart (CG, allow_restart): r = r_explicit, p = z = r/K and
// partial C = z.r. Decisions are recomputed identically by explicit_fin_kernel, which then updates the state in place.
__device__ __forceinline__ void
explicit_decide(const SolverState & st, double err_explicit, int allow_restart, bool & promote, bool & restart)
{
	promote = err_explicit < st.err_best;
	const double err_best = promote ? err_explicit : st.err_best;
	restart = allow_restart && (err_best > st.eps_counter) && (err_explicit / st.err > 1e3);
}

template <typename T>
__global__ __launch_bounds__(VB) void
explicit_kernel(const SolverState * __restrict__ st_p, const T * __restrict__ x, T * __restrict__ x_best,
		const T * __restrict__ r_explicit, T * __restrict__ r, T * __restrict__ p, const T * __restrict__ K, long m,
		int nb, int allow_restart, int ignore_done, double * __restrict__ part)
{
	const SolverState st = *st_p;
	if (st.done && !ignore_done)
		return;
	const double err_explicit = sqrt(sum_partials(part, P_A, nb));
	bool promote, restart;
	explicit_decide(st, err_explicit, allow_restart, promote, restart);
	double zr = 0;
	if (promote || restart)
	{
		GRID_STRIDE(i, m)
		{
			if (promote)
				x_best[i] = x[i];
			if (restart)
			{
				const T ri = r_explicit[i];
				const T zi = ri / K[i];
				r[i] = ri;
				p[i] = zi;
				zr += (double) zi * (double) ri;
			}
		}
	}
	if (restart)
		store_partial(part, P_C, zr);
}

__global__ __launch_bounds__(VB) void
explicit_fin_kernel(SolverState * __restrict__ st_p, int nb, int allow_restart, int ignore_done, const double * __restrict__ part)
{
	SolverState st = *st_p;
	if (st.done && !ignore_done)
		return;
	const double err_explicit = sqrt(sum_partials(part, P_A, nb));
	bool promote, restart;
	explicit_decide(st, err_explicit, allow_restart, promote, restart);
	double zr = 0;
	if (restart)
		zr = sum_partials(part, P_C, nb);
	if (threadIdx.x == 0)
	{
		st.err_explicit = err_explicit;
		if (promote)
			st.err_best = err_explicit;
		if (restart)
		{
			st.zr = zr;
			st.restarts++;
		}
		*st_p = st;
	}
}

// ------------------------------------------------------------------------------------------------ CG

// z0 = r0/K, p0 = z0 (bench_cg.cpp:153-157); partial C = z.r
template <typename T>
__global__ __launch_bounds__(VB) void
cg_init_kernel(const T * __restrict__ r, const T * __restrict__ K, T * __restrict__ p, long m, double * __restrict__ part)
{
	double zr = 0;
	GRID_STRIDE(i, m)
	{
		const T ri = r[i];
		const T zi = ri / K[i];
		p[i] = zi;
		zr += (double) zi * (double) ri;
	}
	store_partial(part, P_C, zr);
}

// eps / eps_counter / first error (bench_cg.cpp:159-182); 1 block. mode 0 = CG (zr = z.r from partial C),
// mode 1 = BiCGSTAB (zr = s_pk_p = (r0_, rk) = r.r since r0_ = rk, bench_bicg.cpp:232-241).
__global__ __launch_bounds__(VB) void
init_state_kernel(SolverState * __restrict__ st_p, int nb, int mode, const double * __restrict__ part)
{
	const double rr = sum_partials(part, P_A, nb);
	const double bb = sum_partials(part, P_B, nb);
	const double zr = mode == 0 ? sum_partials(part, P_C, nb) : rr;
	if (threadIdx.x == 0)
	{
		SolverState st;
		const double b_norm = sqrt(bb);
		st.err = sqrt(rr);
		st.eps = 1.0e-15 * b_norm;
		st.eps_counter = 1.0e-7 * b_norm;
		st.err_explicit = st.err;
		st.err_best = st.err;
		st.zr = zr;
		st.k = 0;
		st.restarts = 0;
		st.done = mode == 0 && st.err < st.eps;      // the first `if (err < eps) break` (k = 0); BiCGSTAB never breaks
		st.pad = 0;
		st_p[0] = st;
		st_p[1] = st;
	}
}

// partial A = p.Ap ; block 0 records the per-iteration report line (bench_cg.cpp:249)
template <typename T>
__global__ __launch_bounds__(VB) void
cg_dot_kernel(const SolverState * __restrict__ st_p, const T * __restrict__ p, const T * __restrict__ Ap, long m,
		double * __restrict__ history, long it, double * __restrict__ part)
{
	const SolverState st = *st_p;
	if (st.done)
		return;
	if (history && blockIdx.x == 0 && threadIdx.x == 0)
	{
		history[3 * it + 0] = st.err;
		history[3 * it + 1] = st.err_explicit;
		history[3 * it + 2] = st.err_best;
	}
	double s = 0;
	GRID_STRIDE(i, m)
		s += (double) p[i] * (double) Ap[i];
	store_partial(part, P_A, s);
}

// ak = (z.r)/(p.Ap); x += ak p; r -= ak Ap; z = r/K (not stored); partials D = z.r, E = r.r   (bench_cg.cpp:259-274)
template <typename T>
__global__ __launch_bounds__(VB) void
cg_update_kernel(const SolverState * __restrict__ st_p, T * __restrict__ x, T * __restrict__ r, const T * __restrict__ p,
		const T * __restrict__ Ap, const T * __restrict__ K, long m, int nb, double * __restrict__ part)
{
	const SolverState st = *st_p;
	if (st.done)
		return;
	const T ak = (T) (st.zr / sum_partials(part, P_A, nb));
	double zr = 0, rr = 0;
	GRID_STRIDE(i, m)
	{
		x[i] = x[i] + ak * p[i];
		const T ri = r[i] + (-ak) * Ap[i];
		r[i] = ri;
		const T zi = ri / K[i];
		zr += (double) zi * (double) ri;
		rr += (double) ri * (double) ri;
	}
	store_partial(part, P_D, zr);
	store_partial(part, P_E, rr);
}

// bk = (z.r)_new / (z.r)_old ; p = z + bk p (bench_cg.cpp:278-283); block 0 writes the next state: k+1, err = |r|
// and the `err < eps` break of the next loop top (bench_cg.cpp:209-214,238-239).
template <typename T>
__global__ __launch_bounds__(VB) void
cg_direction_kernel(const SolverState * __restrict__ st_p, SolverState * __restrict__ st_next, const T * __restrict__ r,
		T * __restrict__ p, const T * __restrict__ K, long m, int nb, const double * __restrict__ part, long it,
		volatile long * host_progress)
{
	const SolverState st = *st_p;
	if (st.done)
	{
		if (blockIdx.x == 0 && threadIdx.x == 0)
		{
			*st_next = st;
			post_progress(host_progress, it + 1, st.k);
		}
		return;
	}
	const double zr_new = sum_partials(part, P_D, nb);
	const T bk = (T) (zr_new / st.zr);
	GRID_STRIDE(i, m)
		p[i] = r[i] / K[i] + bk * p[i];
	if (blockIdx.x == 0)
	{
		const double rr = sum_partials(part, P_E, nb);
		if (threadIdx.x == 0)
		{
			SolverState nx = st;
			nx.zr = zr_new;
			nx.err = sqrt(rr);
			nx.k = st.k + 1;
			nx.done = nx.err < nx.eps;
			*st_next = nx;
			post_progress(host_progress, it + 1, nx.done ? nx.k : -1);
		}
	}
}

// ------------------------------------------------------------------------------------------------ BiCGSTAB

// r0_ = r, p = r, y = p/K (bench_bicg.cpp:232-246,328-332)
template <typename T>
__global__ __launch_bounds__(VB) void
bicg_init_kernel(const T * __restrict__ r, const T * __restrict__ K, T * __restrict__ r0, T * __restrict__ p, T * __restrict__ y, long m)
{
	GRID_STRIDE(i, m)
	{
		const T ri = r[i];
		r0[i] = ri;
		p[i] = ri;
		y[i] = ri / K[i];
	}
}

// partial A = r0_.v ; block 0 records the report line (bench_bicg.cpp:323)
template <typename T>
__global__ __launch_bounds__(VB) void
bicg_dot_kernel(const SolverState * __restrict__ st_p, const T * __restrict__ r0, const T * __restrict__ v, long m,
		double * __restrict__ history, long it, double * __restrict__ part)
{
	const SolverState st = *st_p;
	if (history && blockIdx.x == 0 && threadIdx.x == 0)
	{
		history[3 * it + 0] = st.err;
		history[3 * it + 1] = st.err_explicit;
		history[3 * it + 2] = st.err_best;
	}
	double s = 0;
	GRID_STRIDE(i, m)
		s += (double) r0[i] * (double) v[i];
	store_partial(part, P_A, s);
}

// s_a = s_pk_p / (r0_.v); s = r - s_a v; z = s/K (bench_bicg.cpp:343-360)
template <typename T>
__global__ __launch_bounds__(VB) void
bicg_s_kernel(const SolverState * __restrict__ st_p, const T * __restrict__ r, const T * __restrict__ v, const T * __restrict__ K,
		T * __restrict__ s, T * __restrict__ z, long m, int nb, const double * __restrict__ part)
{
	const SolverState st = *st_p;
	const T s_a = (T) ((T) st.zr / (T) sum_partials(part, P_A, nb));
	GRID_STRIDE(i, m)
	{
		const T si = r[i] + (-s_a) * v[i];
		s[i] = si;
		z[i] = si / K[i];
	}
}

// partials B = sum (t/K)(s/K), C = sum (t/K)^2 (bench_bicg.cpp:374-391)
template <typename T>
__global__ __launch_bounds__(VB) void
bicg_omega_kernel(const T * __restrict__ t, const T * __restrict__ s, const T * __restrict__ K, long m, double * __restrict__ part)
{
	double ts = 0, tt = 0;
	GRID_STRIDE(i, m)
	{
		const T ki = K[i];
		const T v1 = t[i] / ki;
		const T v2 = s[i] / ki;
		ts += (double) v1 * (double) v2;
		tt += (double) v1 * (double) v1;
	}
	store_partial(part, P_B, ts);
	store_partial(part, P_C, tt);
}

// s_w; r = s - s_w t; x = (x + s_a y) + s_w z; partials D = r0_.r, E = r.r (bench_bicg.cpp:350,394-402)
template <typename T>
__global__ __launch_bounds__(VB) void
bicg_update_kernel(const SolverState * __restrict__ st_p, const T * __restrict__ s, const T * __restrict__ t, const T * __restrict__ y,
		const T * __restrict__ z, const T * __restrict__ r0, T * __restrict__ r, T * __restrict__ x, long m, int nb,
		double * __restrict__ part)
{
	const SolverState st = *st_p;
	const T s_a = (T) ((T) st.zr / (T) sum_partials(part, P_A, nb));
	const T s_w = (T) ((T) sum_partials(part, P_B, nb) / (T) sum_partials(part, P_C, nb));
	double r0r = 0, rr = 0;
	GRID_STRIDE(i, m)
	{
		const T ri = s[i] + (-s_w) * t[i];
		r[i] = ri;
		const T hi = x[i] + s_a * y[i];
		x[i] = hi + s_w * z[i];
		r0r += (double) r0[i] * (double) ri;
		rr += (double) ri * (double) ri;
	}
	store_partial(part, P_D, r0r);
	store_partial(part, P_E, rr);
}

// s_b = (s_pk/s_pk_p)(s_a/s_w); p = r + s_b (p - s_w v); y = p/K for the next iteration; block 0 writes the next state
// (bench_bicg.cpp:402-419, 304-311, 328-332)
template <typename T>
__global__ __launch_bounds__(VB) void
bicg_direction_kernel(const SolverState * __restrict__ st_p, SolverState * __restrict__ st_next, const T * __restrict__ r,
		const T * __restrict__ v, const T * __restrict__ K, T * __restrict__ p, T * __restrict__ y, long m, int nb,
		const double * __restrict__ part, long it, volatile long * host_progress)
{
	const SolverState st = *st_p;
	const T s_pk_p = (T) st.zr;
	const T s_a = (T) (s_pk_p / (T) sum_partials(part, P_A, nb));
	const T s_w = (T) ((T) sum_partials(part, P_B, nb) / (T) sum_partials(part, P_C, nb));
	const double s_pk_d = sum_partials(part, P_D, nb);
	const T s_pk = (T) s_pk_d;
	const T s_b = (s_pk / s_pk_p) * (s_a / s_w);
	GRID_STRIDE(i, m)
	{
		const T pi = r[i] + s_b * (p[i] - s_w * v[i]);
		p[i] = pi;
		y[i] = pi / K[i];
	}
	if (blockIdx.x == 0)
	{
		const double rr = sum_partials(part, P_E, nb);
		if (threadIdx.x == 0)
		{
			SolverState nx = st;
			nx.zr = (double) s_pk;
			nx.err = sqrt(rr);
			nx.k = st.k + 1;
			*st_next = nx;
			post_progress(host_progress, it + 1, -1);
		}
	}
}

// ------------------------------------------------------------------------------------------------ host side

struct DeviceBuffers {
	std::vector<void *> ptrs;
	void * pinned = nullptr;
	~DeviceBuffers()
	{
		for (void * p : ptrs)
			(void) hipFree(p);
		if (pinned)
			(void) hipHostFree(pinned);
	}
	template <typename P>
	int alloc(P ** out, size_t bytes)
	{
		void * p = nullptr;
		HIP_TRY(hipMalloc(&p, bytes ? bytes : 8));
		ptrs.push_back(p);
		*out = (P *) p;
		return 0;
	}
};

// Jacobi preconditioner: the first stored entry of row i whose column is i (bench_cg.cpp:114-134).
template <typename T>
static long
jacobi_diagonal(const int32_t * row_ptr, const int32_t * col, const double * val, long m, long row_offset, T * K)
{
	long bad = -1;
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static)
	for (long i = 0; i < m; i++)
	{
		T k = 0;
		for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++)
			if (col[j] == i + row_offset)
			{
				k = (T) val[j];
				break;
			}
		K[i] = k;
		if (k == 0)
		{
			#pragma omp critical
			if (bad < 0 || i < bad)
				bad = i;
		}
	}
	return bad;
}

#define ABI_TRY(expr)       \
	do {                    \
		if ((expr))         \
			return 1;       \
	} while (0)

template <typename T>
static int
solve(int method, spmv_mi355x_matrix * A, const spmv_mi355x_dist_ops * dist, long m_arg, const int32_t * row_ptr, const int32_t * col,
		const double * val, const void * b_host, void * x_host, long max_iterations, double * history_host,
		spmv_mi355x_solver_info * info)
{
	const auto t_start = std::chrono::steady_clock::now();
	const long m = dist ? m_arg : spmv_mi355x_rows(A);
	hipStream_t stream = nullptr;
	DeviceBuffers buf;
	const size_t vb = (size_t) m * sizeof(T);

	std::vector<T> K_host((size_t) std::max<long>(m, 1));
	const long bad = jacobi_diagonal<T>(row_ptr, col, val, m, dist ? dist->row_offset : 0, K_host.data());
	if (bad >= 0)
	{
		set_error("bad K, zero in diagonal (row %ld)", bad);
		return 1;
	}

	T * b, * K, * x, * x_best, * r, * r_explicit, * p, * Ap;
	T * r0 = nullptr, * y = nullptr, * z = nullptr, * s = nullptr, * v = nullptr;
	// Plain allocations, the SpMV outputs (Ap, v) included: the engine's placement search (placement.hip) costs ~1 s and leaves the
	// driver clearing 160 GiB for seconds afterwards — more than a solve of a few hundred iterations takes (measured: 0.27 instead of
	// 0.24 ms per CG iteration on the 160^3 stencil right after it). A caller who solves many systems with one handle can hand in
	// vectors from spmv_mi355x_output_alloc through the device-pointer entry points instead.
	for (T ** q : {&b, &K, &x, &x_best, &r, &r_explicit, &p, &Ap})
		ABI_TRY(buf.alloc(q, vb));
	if (method == 1)
		for (T ** q : {&r0, &y, &z, &s, &v})
			ABI_TRY(buf.alloc(q, vb));
	double * part, * history = nullptr;
	SolverState * st;
	ABI_TRY(buf.alloc(&part, sizeof(double) * NUM_SLOTS * MAX_PART));
	ABI_TRY(buf.alloc(&st, 2 * sizeof(SolverState)));
	if (history_host && max_iterations > 0)
	{
		ABI_TRY(buf.alloc(&history, sizeof(double) * 3 * (size_t) max_iterations));
		HIP_TRY(hipMemsetAsync(history, 0, sizeof(double) * 3 * (size_t) max_iterations, stream));
	}
	HIP_TRY(hipHostMalloc(&buf.pinned, 2 * sizeof(long), hipHostMallocMapped | hipHostMallocCoherent));
	volatile long * progress = (volatile long *) buf.pinned;          // [0] iterations finished, [1] break flag
	progress[0] = 0;
	progress[1] = -1;
	long * progress_dev = nullptr;
	HIP_TRY(hipHostGetDevicePointer((void **) &progress_dev, buf.pinned, 0));

	HIP_TRY(hipMemcpyAsync(b, b_host, vb, hipMemcpyHostToDevice, stream));
	HIP_TRY(hipMemcpyAsync(K, K_host.data(), vb, hipMemcpyHostToDevice, stream));
	HIP_TRY(hipMemsetAsync(x, 0, vb, stream));                    // x0 = 0 (bench_cg.cpp:139-145)
	HIP_TRY(hipMemsetAsync(x_best, 0, vb, stream));
	HIP_TRY(hipMemsetAsync(part, 0, sizeof(double) * NUM_SLOTS * MAX_PART, stream));

	const int nb = (int) std::min<long>(MAX_PART, std::max<long>(1, (m + 4 * VB - 1) / (4 * VB)));
	const dim3 grid(nb), block(VB), one(1);
	long spmv_calls = 0;
	auto spmv = [&](const T * in, T * out) {
		spmv_calls++;
		if (dist)
		{
			if (dist->spmv(dist->ctx, in, out))
			{
				set_error("solver: the caller's distributed spmv callback failed");
				return 1;
			}
			return 0;
		}
		return spmv_mi355x_spmv_device_async(A, in, out, 0, stream);
	};
	// single GPU: consumers re-reduce the nb per-block partials themselves. Distributed: the listed slots are reduced,
	// summed over the ranks by the caller's collective and put back as ONE partial; consumers then read nbc = 1 partial.
	const int nbc = dist ? 1 : nb;
	auto global_reduce = [&](SlotList sl) {
		if (!dist)
			return 0;
		hipLaunchKernelGGL(reduce_slots_kernel, dim3(sl.n), block, 0, stream, part, nb, sl, dist->reduce_buf_dev);
		if (dist->allreduce_sum(dist->ctx, dist->reduce_buf_dev, sl.n))
		{
			set_error("solver: the caller's all-reduce callback failed");
			return 1;
		}
		hipLaunchKernelGGL(scatter_slots_kernel, one, dim3(WAVE), 0, stream, part, sl, dist->reduce_buf_dev);
		return 0;
	};
	// |b - A x|^2 into partial A, r_explicit = b - A x
	auto explicit_residual = [&](const T * xx) {
		if (spmv(xx, Ap))
			return 1;
		hipLaunchKernelGGL((residual_kernel<T>), grid, block, 0, stream, b, Ap, r_explicit, m, part);
		return global_reduce({1, {P_A, 0, 0}});
	};

	// r0 = b - A x0
	ABI_TRY(spmv(x, Ap));
	hipLaunchKernelGGL((residual_kernel<T>), grid, block, 0, stream, b, Ap, r, m, part);
	if (method == 0)
		hipLaunchKernelGGL((cg_init_kernel<T>), grid, block, 0, stream, r, K, p, m, part);
	else
		hipLaunchKernelGGL((bicg_init_kernel<T>), grid, block, 0, stream, r, K, r0, p, y, m);
	ABI_TRY(global_reduce({3, {P_A, P_B, P_C}}));
	hipLaunchKernelGGL(init_state_kernel, one, block, 0, stream, st, nbc, method, part);
	HIP_TRY(hipGetLastError());

	const bool debug = getenv("SPMV_MI355X_SOLVER_DEBUG") != nullptr;
	double t_spin = 0;
	const auto t_loop = std::chrono::steady_clock::now();
	long it = 0;
	for (; it < max_iterations; it++)
	{
		if (it % POLL == 0 && it >= 2 * POLL)
		{
			// stay at most 2*POLL iterations ahead; plain reads of the mapped word, no HIP call
			const auto t_wait = std::chrono::steady_clock::now();
			long spins = 0;
			while (progress[0] < it - POLL)
			{
				if ((++spins & 0xfff) == 0)
				{
					HIP_TRY(hipGetLastError());
					if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count() > 120.0)
					{
						set_error("solver: the device made no progress for 120 s at iteration %ld", it);
						(void) hipStreamSynchronize(stream);
						return 1;
					}
				}
				__builtin_ia32_pause();
			}
			t_spin += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count();
			// stop rule that every rank of a distributed solve evaluates identically: only what the device had posted by
			// iteration it - POLL counts (the wait above guarantees it is visible), never "whatever is visible now"
			const long broke_at = progress[1];
			if (broke_at >= 0 && broke_at <= it - POLL)
				break;
		}
		SolverState * cur = st + (it & 1), * nxt = st + ((it + 1) & 1);
		if (it > 0 && it % RESTART_K == 0)
		{
			ABI_TRY(explicit_residual(x));
			hipLaunchKernelGGL((explicit_kernel<T>), grid, block, 0, stream, cur, x, x_best, r_explicit, r, p, K, m, nbc,
					method == 0, 0, part);
			if (method == 0)
				ABI_TRY(global_reduce({1, {P_C, 0, 0}}));             // z.r of a restart (stale and unread otherwise)
			hipLaunchKernelGGL(explicit_fin_kernel, one, block, 0, stream, cur, nbc, method == 0, 0, part);
		}
		if (method == 0)
		{
			ABI_TRY(spmv(p, Ap));
			hipLaunchKernelGGL((cg_dot_kernel<T>), grid, block, 0, stream, cur, p, Ap, m, history, it, part);
			ABI_TRY(global_reduce({1, {P_A, 0, 0}}));
			hipLaunchKernelGGL((cg_update_kernel<T>), grid, block, 0, stream, cur, x, r, p, Ap, K, m, nbc, part);
			ABI_TRY(global_reduce({2, {P_D, P_E, 0}}));
			hipLaunchKernelGGL((cg_direction_kernel<T>), grid, block, 0, stream, cur, nxt, r, p, K, m, nbc, part, it, progress_dev);
		}
		else
		{
			ABI_TRY(spmv(y, v));
			hipLaunchKernelGGL((bicg_dot_kernel<T>), grid, block, 0, stream, cur, r0, v, m, history, it, part);
			ABI_TRY(global_reduce({1, {P_A, 0, 0}}));
			hipLaunchKernelGGL((bicg_s_kernel<T>), grid, block, 0, stream, cur, r, v, K, s, z, m, nbc, part);
			ABI_TRY(spmv(z, Ap));                                // t = A z
			hipLaunchKernelGGL((bicg_omega_kernel<T>), grid, block, 0, stream, Ap, s, K, m, part);
			ABI_TRY(global_reduce({2, {P_B, P_C, 0}}));
			hipLaunchKernelGGL((bicg_update_kernel<T>), grid, block, 0, stream, cur, s, Ap, y, z, r0, r, x, m, nbc, part);
			ABI_TRY(global_reduce({2, {P_D, P_E, 0}}));
			hipLaunchKernelGGL((bicg_direction_kernel<T>), grid, block, 0, stream, cur, nxt, r, v, K, p, y, m, nbc, part, it,
					progress_dev);
		}
	}
	HIP_TRY(hipGetLastError());
	const auto t_loop_end = std::chrono::steady_clock::now();
	if (debug)
	{
		HIP_TRY(hipStreamSynchronize(stream));
		const auto t_sync = std::chrono::steady_clock::now();
		fprintf(stderr, "[solver] setup %.3f ms, enqueue loop %.3f ms (spin %.3f ms), drain %.3f ms, %ld iterations launched\n",
				std::chrono::duration<double>(t_loop - t_start).count() * 1e3,
				std::chrono::duration<double>(t_loop_end - t_loop).count() * 1e3, t_spin * 1e3,
				std::chrono::duration<double>(t_sync - t_loop_end).count() * 1e3, it);
	}

	// final explicit residual of x, promotion of x_best (bench_cg.cpp:288-306); runs after a break too
	SolverState * fin = st + (it & 1);
	ABI_TRY(explicit_residual(x));
	hipLaunchKernelGGL((explicit_kernel<T>), grid, block, 0, stream, fin, x, x_best, r_explicit, r, p, K, m, nbc, 0, 1, part);
	hipLaunchKernelGGL(explicit_fin_kernel, one, block, 0, stream, fin, nbc, 0, 1, part);
	// the harness's own check of the returned vector: error = |b - A x_best| (bench_cg.cpp:412-418)
	ABI_TRY(explicit_residual(x_best));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(x_host, x_best, vb, hipMemcpyDeviceToHost, stream));
	SolverState st_host;
	std::vector<double> part_host((size_t) nbc);
	HIP_TRY(hipMemcpyAsync(&st_host, fin, sizeof(SolverState), hipMemcpyDeviceToHost, stream));
	HIP_TRY(hipMemcpyAsync(part_host.data(), part + (long) P_A * MAX_PART, sizeof(double) * nbc, hipMemcpyDeviceToHost, stream));
	if (history)
		HIP_TRY(hipMemcpyAsync(history_host, history, sizeof(double) * 3 * (size_t) max_iterations, hipMemcpyDeviceToHost, stream));
	HIP_TRY(hipStreamSynchronize(stream));
	if (info)
	{
		spmv_mi355x_solver_info out;
		memset(&out, 0, sizeof(out));
		double ee = 0;
		for (double v : part_host)
			ee += v;
		out.iterations = st_host.k;
		out.error = std::sqrt(ee);
		out.error_best = st_host.err_best;
		out.eps = st_host.eps;
		out.eps_counter = st_host.eps_counter;
		out.restarts = st_host.restarts;
		out.spmv_calls = spmv_calls - 1;                          // the last one is the harness's check, not the solver's
		out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
		if (debug)
			fprintf(stderr, "[solver] tail (final residuals, downloads) %.3f ms, total %.3f ms\n",
					std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop_end).count() * 1e3, out.seconds * 1e3);
		const unsigned want = info->struct_size;
		out.struct_size = sizeof(out);
		memcpy(info, &out, std::min<size_t>(want, sizeof(out)));
		info->struct_size = (unsigned) std::min<size_t>(want, sizeof(out));
	}
	return 0;
}

static int
solve_entry(int method, spmv_mi355x_matrix * A, const int32_t * row_ptr, const int32_t * col, const double * val, const void * b_host,
		void * x_host, long max_iterations, double * history_host, spmv_mi355x_solver_info * info)
{
	if (!A || !row_ptr || !b_host || !x_host)
	{
		set_error("solver: NULL argument");
		return 1;
	}
	if (info && info->struct_size < 8)
	{
		set_error("solver: info->struct_size not set");
		return 1;
	}
	if (spmv_mi355x_rows(A) != spmv_mi355x_cols(A))
	{
		set_error("the matrix must be square");                  // bench_cg.cpp:487-488
		return 1;
	}
	if (max_iterations < 0)
	{
		set_error("solver: max_iterations < 0");
		return 1;
	}
	if (spmv_mi355x_nnz(A) > 0 && (!col || !val))
	{
		set_error("solver: NULL CSR arrays");
		return 1;
	}
	HIP_TRY(hipSetDevice(spmv_mi355x_device(A)));
	if (spmv_mi355x_precision(A) == SPMV_MI355X_F32)
		return solve<float>(method, A, nullptr, 0, row_ptr, col, val, b_host, x_host, max_iterations, history_host, info);
	return solve<double>(method, A, nullptr, 0, row_ptr, col, val, b_host, x_host, max_iterations, history_host, info);
}

static int
solve_dist_entry(int method, const spmv_mi355x_dist_ops * ops, int precision, long m_local, const int32_t * row_ptr, const int32_t * col,
		const double * val, const void * b_host, void * x_host, long max_iterations, double * history_host,
		spmv_mi355x_solver_info * info)
{
	if (!ops || ops->struct_size < sizeof(spmv_mi355x_dist_ops) || !ops->spmv || !ops->allreduce_sum || !ops->reduce_buf_dev)
	{
		set_error("distributed solver: ops incomplete (struct_size, spmv, allreduce_sum and reduce_buf_dev are required)");
		return 1;
	}
	if (!row_ptr || !b_host || !x_host || m_local < 0 || max_iterations < 0 || (row_ptr[m_local] > 0 && (!col || !val)))
	{
		set_error("distributed solver: bad argument");
		return 1;
	}
	if (info && info->struct_size < 8)
	{
		set_error("solver: info->struct_size not set");
		return 1;
	}
	if (precision == SPMV_MI355X_F32)
		return solve<float>(method, nullptr, ops, m_local, row_ptr, col, val, b_host, x_host, max_iterations, history_host, info);
	if (precision == SPMV_MI355X_F64)
		return solve<double>(method, nullptr, ops, m_local, row_ptr, col, val, b_host, x_host, max_iterations, history_host, info);
	set_error("unknown precision %d", precision);
	return 1;
}

}  // namespace spmv

extern "C" {

int
spmv_mi355x_pcg_dist(const spmv_mi355x_dist_ops * ops, int precision, long m_local, const int32_t * row_ptr_local,
		const int32_t * col_idx_global, const double * values_fp64, const void * b_local_host, void * x_local_host,
		long max_iterations, double * history_out, spmv_mi355x_solver_info * info)
{
	return spmv::solve_dist_entry(0, ops, precision, m_local, row_ptr_local, col_idx_global, values_fp64, b_local_host, x_local_host,
			max_iterations, history_out, info);
}

int
spmv_mi355x_pbicgstab_dist(const spmv_mi355x_dist_ops * ops, int precision, long m_local, const int32_t * row_ptr_local,
		const int32_t * col_idx_global, const double * values_fp64, const void * b_local_host, void * x_local_host,
		long max_iterations, double * history_out, spmv_mi355x_solver_info * info)
{
	return spmv::solve_dist_entry(1, ops, precision, m_local, row_ptr_local, col_idx_global, values_fp64, b_local_host, x_local_host,
			max_iterations, history_out, info);
}

int
spmv_mi355x_pcg(spmv_mi355x_matrix * A, const int32_t * row_ptr, const int32_t * col_idx, const double * values_fp64,
		const void * b_host, void * x_host, long max_iterations, double * history_out, spmv_mi355x_solver_info * info)
{
	return spmv::solve_entry(0, A, row_ptr, col_idx, values_fp64, b_host, x_host, max_iterations, history_out, info);
}

int
spmv_mi355x_pbicgstab(spmv_mi355x_matrix * A, const int32_t * row_ptr, const int32_t * col_idx, const double * values_fp64,
		const void * b_host, void * x_host, long max_iterations, double * history_out, spmv_mi355x_solver_info * info)
{
	return spmv::solve_entry(1, A, row_ptr, col_idx, values_fp64, b_host, x_host, max_iterations, history_out, info);
}

}
