// Row-sorted COO SpMV for gfx950.
//
// In the reference the COO multiply is third-party (mkl_cspblas_?coogemv at benchmark_code/BENCH/src/spmv_kernels/
// mkl_coo.cpp:102-104, rocsparse_dcoomv at GPU_clean/rocsparse_coo.cpp:294); only the CSR -> COO row expansion
// (mkl_coo.cpp:79-90) is in-repo, and it yields entries sorted by row. This kernel relies on that order.
//
// Each wavefront owns 64*K consecutive entries. All K (row, col, val) triples per lane are loaded up front (coalesced
// 4+4+sizeof(V) byte streams), then for each of the K wave-steps:
//   - p = val * x[col]; lane 0 adds the partial sum carried from the previous step if it continues the same row;
//   - segmented inclusive scan by row over the wave (6 shuffle steps: rows are sorted, so "same row as lane-off"
//     means the whole span is one segment);
//   - a lane whose successor has a different row ends a segment and stores y[row] (plain store: a row that ends
//     inside the wave's chunk has all of its in-chunk entries in that sum).
// The segment still open at the end of the chunk goes to (carry_row, carry_val)[wave]; coo_fixup_kernel adds the carries
// in wave order. y is cleared first (rows without entries must read 0; a row ending exactly at a chunk end is written
// only by the fix-up) unless beta == 1. No atomics: results are reproducible run to run.

#include <algorithm>

#include "launch.hpp"

namespace spmv {

constexpr int COO_BLOCK = 256;
constexpr int COO_WAVES = COO_BLOCK / WAVE;

template <typename T, int K, bool NT>
__global__ __launch_bounds__(COO_BLOCK) void
coo_kernel(const int * __restrict__ rowind, const int * __restrict__ col, const T * __restrict__ val,
		const T * __restrict__ x, T * __restrict__ y, long nnz, int num_waves,
		int * __restrict__ carry_row, T * __restrict__ carry_val, int beta, XcdMap map)
{
	unsigned tile = xcd_tile(blockIdx.x, map);
	if (tile == NO_TILE)
		return;
	const int lane = threadIdx.x % WAVE;
	const int w = tile * COO_WAVES + threadIdx.x / WAVE;
	if (w >= num_waves)
		return;
	const long start = (long) w * (WAVE * K);

	int r[K], c[K];
	T v[K];
	#pragma unroll
	for (int k = 0; k < K; k++)
	{
		long idx = start + (long) k * WAVE + lane;
		bool ok = idx < nnz;
		long j = ok ? idx : 0;
		r[k] = ok ? ld_stream<NT>(rowind + j) : -1;     // -1 = past the end (never equals a real row)
		c[k] = ok ? ld_stream<NT>(col + j) : 0;
		v[k] = ok ? ld_stream<NT>(val + j) : T(0);
	}
	T p[K];
	#pragma unroll
	for (int k = 0; k < K; k++)
		p[k] = (r[k] >= 0) ? v[k] * x[c[k]] : T(0);

	int open_row = -1;      // row of lane 63 after the previous step (wave-uniform)
	T open_val = 0;         // its inclusive segment sum
	#pragma unroll
	for (int k = 0; k < K; k++)
	{
		T s = p[k];
		const int row = r[k];
		if (lane == 0 && row >= 0 && row == open_row)
			s += open_val;
		#pragma unroll
		for (int off = 1; off < WAVE; off <<= 1)
		{
			T so = shfl_up_t(s, off);
			int ro = __shfl_up(row, off, WAVE);
			if (lane >= off && ro == row)
				s += so;
		}
		// successor row: the next lane, or for lane 63 the first lane of the next step (-2 after the last step)
		int next_row = __shfl_down(row, 1, WAVE);
		const int next_first = (k + 1 < K) ? __shfl(r[(k + 1 < K) ? k + 1 : k], 0, WAVE) : -2;
		if (lane == WAVE - 1)
			next_row = next_first;
		if (row >= 0 && next_row != row)
		{
			// a segment ends at this lane. If nothing valid follows inside the chunk it may continue in the next
			// wave's chunk: exactly one lane per wave takes this branch and hands the sum to the fix-up.
			const bool chunk_final = (next_row == -1) || (k + 1 == K && lane == WAVE - 1);
			if (chunk_final)
			{
				carry_row[w] = row;
				carry_val[w] = s;
			}
			else
			{
				T * yp = y + row;
				*yp = beta ? *yp + s : s;
			}
		}
		open_row = __shfl(row, WAVE - 1, WAVE);
		open_val = __shfl(s, WAVE - 1, WAVE);
	}
}

template <typename T>
__global__ __launch_bounds__(COO_BLOCK) void
coo_fixup_kernel(const int * __restrict__ carry_row, const T * __restrict__ carry_val, int ncarry, int m, T * __restrict__ y)
{
	int t = blockIdx.x * COO_BLOCK + threadIdx.x;
	if (t >= ncarry)
		return;
	int row = carry_row[t];
	if (row < 0 || row >= m)
		return;
	if (t > 0 && carry_row[t - 1] == row)
		return;
	T sum = carry_val[t];
	for (int u = t + 1; u < ncarry && carry_row[u] == row; u++)
		sum += carry_val[u];
	y[row] += sum;
}

// ------------------------------------------------------------------------------------------------ launchers

static int
coo_default_k(int items_per_lane)
{
	if (items_per_lane == 2 || items_per_lane == 4 || items_per_lane == 8)
		return items_per_lane;
	return 4;
}

int
coo_wave_items(int items_per_lane)
{
	return WAVE * coo_default_k(items_per_lane);
}

template <typename T, int K>
static int
coo_launch_k(const int * rowind, const int * col, const void * val, const void * x, void * y, int m, long nnz, int num_waves,
		int * carry_row, void * carry_val, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	unsigned grid = xcd_grid(cfg.map);
	if (grid_out)
		*grid_out = grid;
	if (!cfg.beta && m > 0)
		HIP_TRY(hipMemsetAsync(y, 0, (size_t) m * sizeof(T), stream));
	if (grid == 0)
		return 0;
	if (cfg.nt)
		hipLaunchKernelGGL((coo_kernel<T, K, true>), dim3(grid), dim3(COO_BLOCK), 0, stream, rowind, col, (const T *) val,
				(const T *) x, (T *) y, nnz, num_waves, carry_row, (T *) carry_val, cfg.beta, cfg.map);
	else
		hipLaunchKernelGGL((coo_kernel<T, K, false>), dim3(grid), dim3(COO_BLOCK), 0, stream, rowind, col, (const T *) val,
				(const T *) x, (T *) y, nnz, num_waves, carry_row, (T *) carry_val, cfg.beta, cfg.map);
	HIP_TRY(hipGetLastError());
	unsigned fgrid = (unsigned) ((num_waves + COO_BLOCK - 1) / COO_BLOCK);
	hipLaunchKernelGGL((coo_fixup_kernel<T>), dim3(fgrid), dim3(COO_BLOCK), 0, stream, carry_row, (const T *) carry_val,
			num_waves, m, (T *) y);
	HIP_TRY(hipGetLastError());
	return 0;
}

template <typename T>
static int
coo_dispatch(int K, const int * rowind, const int * col, const void * val, const void * x, void * y, int m, long nnz, int num_waves,
		int * carry_row, void * carry_val, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	switch (K)
	{
		case 2: return coo_launch_k<T, 2>(rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out);
		case 4: return coo_launch_k<T, 4>(rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out);
		case 8: return coo_launch_k<T, 8>(rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out);
	}
	set_error("coo: unsupported items per lane %d", K);
	return 1;
}

int
launch_coo(bool f32, int items_per_lane, const int * rowind, const int * col, const void * val, const void * x, void * y,
		int m, long nnz, int num_waves, int * carry_row, void * carry_val,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	int K = coo_default_k(items_per_lane);
	return f32 ? coo_dispatch<float>(K, rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out)
	           : coo_dispatch<double>(K, rowind, col, val, x, y, m, nnz, num_waves, carry_row, carry_val, cfg, stream, grid_out);
}

// ------------------------------------------------------------------------------------------------ column-blocked COO
//
// For graph matrices (soc-LiveJournal1: 14 entries per row scattered over a 39 MB x) the row-sorted kernels sit on the
// fabric's random-sector rate: every gather misses the 4 MiB L2 of its XCD and most of the bytes it brings are unused.
// MI355X has 160 KiB of LDS on each of its 256 CUs — 40 MiB in all, as much as the whole y of such a matrix — and eight L2s
// that each serve 32 CUs. The layout below is built around both:
//   * the rows are cut into 8*P contiguous RANGES (P = 1 while y fits the chip's LDS); XCD k owns ranges k*P .. k*P+P-1 and
//     works through them one after the other;
//   * a range is dealt to 32 workgroups (one per CU, 1024 threads) in chunks of 16 rows so that every workgroup gets the same
//     number of entries (longest chunk first to the emptiest workgroup): every workgroup of the range sees the same column
//     structure, and its y rows (<= 160 KiB) live in LDS for the whole launch, written back once as full 128-byte lines;
//   * a workgroup's entries are sorted by COLUMN and cut into BATCHES of K x 1024 (K entries per lane: 8 for pattern matrices,
//     4 with a value stream). The 64 entries one wave instruction gathers have a base column (a wave-uniform scalar: 16 x K
//     per batch); an entry is ONE dword: column - base (17 bits) | LDS slot of its row (15 bits), + the value unless all values
//     are equal (Matrix-Market `pattern` matrices): 4 bytes per non-zero of stream instead of CSR's 12. Every batch is full (the
//     last one is padded with entries that add into spare LDS slots; so is an instruction whose 64 sorted entries would span more
//     than 2^17 columns), so the loop below has no branch around any load: the compiler counts the outstanding memory operations
//     exactly. The 32 workgroups of an XCD sweep the same columns at about the same time, so that stretch of x enters the XCD's
//     L2 once; neighbouring lanes gather from neighbouring lines;
//   * v*x[col] is added into the LDS copy of y with ds_add_f64 (LDS atomics): the order of a row's additions is run-dependent,
//     parity is to the tolerance (1e-12 / 1e-5), not bit-wise;
//   * a row too long for one workgroup's share (a hub of a power-law graph) is SPLIT over the 32 workgroups of its range, each
//     summing a contiguous piece into an extra LDS slot; the 32 partial sums go to a carry array and a fix-up kernel adds
//     them to y in workgroup order — the merge-path remedy (merge.cpp:302-318: partial rows + carry fix-up) inside this layout.
//
// What bounds it (tools/gather_bench*.hip, profiles/r03_gather_bench*.txt): a CU takes in one 128-byte line of x per ~2 clocks
// (64 B/clk, L2-served) whatever part of the line is used, + ~0.2 clk per gathering lane; the twin touches 0.44 distinct lines per
// entry and workgroup -> ~1.28 clk per entry = 165 us for 69 M entries; ds_add_f64 costs 0.44 clk per entry and runs in the LDS
// pipeline BESIDE the gathers if the gathers of batch i+1 are issued before the adds of batch i: the loop is rotated by hand over
// three entry buffers and three gather buffers (period 3) so that no register copy puts a wait between them. (Round 2's loop had a
// branch around every gather and the compiler put `s_waitcnt vmcnt(0)` after each: 250 us.)
constexpr int COOB_THREADS = 1024;
constexpr int COOB_WGS = 32;               // workgroups per range = CUs per XCD
constexpr int COOB_CHUNK = 16;             // rows per chunk of the round-robin deal
constexpr int COOB_SLOT_BITS = 15;         // LDS slot of the entry's row (a workgroup holds at most 20 480 fp64 slots)
constexpr int COOB_SPARE = 64;             // spare LDS slots the padding entries add into (one per lane of a wave: no conflicts)
constexpr unsigned COOB_SLOT_MASK = (1u << COOB_SLOT_BITS) - 1;

template <typename T, bool UNIT, int K>
struct CoobBatch {
	unsigned e[K];
	T v[UNIT ? 1 : K];
};

// the entry streams are read once: nontemporal, so that they do not push x out of L2
template <typename T, bool UNIT, int K>
__device__ __forceinline__ void
coob_load(CoobBatch<T, UNIT, K> & b, const unsigned * __restrict__ ent, const T * __restrict__ val, size_t first)
{
	#pragma unroll
	for (int u = 0; u < K; u++)
	{
		b.e[u] = ld_stream<true>(ent + first + (size_t) u * COOB_THREADS);
		if constexpr (!UNIT)
			b.v[u] = ld_stream<true>(val + first + (size_t) u * COOB_THREADS);
	}
}

// base: the K base columns of this wave's K instructions of the batch (wave-uniform: scalar loads, scalar x pointers)
template <typename T, bool UNIT, int K>
__device__ __forceinline__ void
coob_gather(T (&xv)[K], const CoobBatch<T, UNIT, K> & b, const T * __restrict__ x, const int * __restrict__ base)
{
	#pragma unroll
	for (int u = 0; u < K; u++)
		xv[u] = (x + base[u])[b.e[u] >> COOB_SLOT_BITS];
}

// The LDS copy of y is fp64 for both precisions: ds_add_f32 runs at a sixth of the rate of ds_add_f64 on this part (2.7 against
// 0.44 clocks per entry and CU, profiles/r03_gather_bench2.txt), and the sums are more accurate.
template <typename T, bool UNIT, int K>
__device__ __forceinline__ void
coob_add(const T (&xv)[K], const CoobBatch<T, UNIT, K> & b, double * __restrict__ ys, T unit)
{
	#pragma unroll
	for (int u = 0; u < K; u++)
		unsafeAtomicAdd(&ys[b.e[u] & COOB_SLOT_MASK], (double) (UNIT ? unit : b.v[u]) * (double) xv[u]);
}

template <typename T, bool UNIT, int K>
__global__ __launch_bounds__(COOB_THREADS) void
coo_blocked_kernel(const int * __restrict__ wg_rows, const int * __restrict__ range_row, const int * __restrict__ chunk_ptr,
		const int * __restrict__ chunk_row, const int * __restrict__ batch_ptr,
		const int * __restrict__ batch_base, const int * __restrict__ range_long, const unsigned * __restrict__ ent,
		const T * __restrict__ val, const T * __restrict__ x, T * __restrict__ y, T * __restrict__ carry,
		int ranges_per_xcd, int chunk_rows, int sync_mask, T unit, int beta)
{
	extern __shared__ __align__(16) unsigned char coob_smem[];
	double * ys = reinterpret_cast<double *>(coob_smem);
	constexpr size_t BATCH = (size_t) K * COOB_THREADS;
	// workgroups are dealt round-robin to the XCDs: the i-th workgroup of XCD k is the (i % 32)-th of its (i / 32)-th range
	const int xcd = (int) blockIdx.x % NUM_XCD;
	const int i = (int) blockIdx.x / NUM_XCD;
	const int range = xcd * ranges_per_xcd + i / COOB_WGS;
	const int j = i % COOB_WGS;
	const int t = range * COOB_WGS + j;
	const int nloc = wg_rows[t];
	if (nloc == 0)
		return;                            // whole workgroup: no rows, hence no entries
	const int tid = (int) threadIdx.x;
	for (int l = tid; l < nloc + COOB_SPARE; l += COOB_THREADS)
		ys[l] = 0;
	const int nb = batch_ptr[t + 1] - batch_ptr[t];
	// base columns: [batch][wave][K]; the wave number is uniform, said so to the compiler for scalar loads
	const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
	const int * __restrict__ bb = batch_base + ((size_t) batch_ptr[t] * (COOB_THREADS / WAVE) + wave) * K;
	constexpr int BSTRIDE = (COOB_THREADS / WAVE) * K;      // ints of base columns per batch
	const size_t e0 = (size_t) batch_ptr[t] * BATCH + (size_t) tid;
	__syncthreads();
	if (nb > 0)
	{
		// nb is a multiple of 3 (the builder appends all-padding batches), and two more batches of entries / base columns lie behind
		// every workgroup's last (the next workgroup's, or slack at the very end): no index below is clamped and no step is conditional
		CoobBatch<T, UNIT, K> q0, q1, q2;
		T xa[K], xb[K], xc[K];
		coob_load<T, UNIT, K>(q0, ent, val, e0);
		coob_load<T, UNIT, K>(q1, ent, val, e0 + BATCH);
		coob_gather<T, UNIT, K>(xa, q0, x, bb);
		// one step = one batch: entry loads of the batch after the next, gathers of the NEXT batch, then this batch's adds into LDS.
		// Vector memory operations complete in issue order, so the order of issue decides what a wait drains: the next step's
		// gathers need the entries loaded here, and with those loads issued BEFORE this step's gathers that wait leaves the
		// gathers in flight (the other order drained the queue at every step). The barrier keeps the 16 waves on one stretch of x.
		#define COOB_STEP(QA, QB, QC, XA, XB, s, k3)                                                                \
		{                                                                                                         \
			coob_load<T, UNIT, K>(QC, ent, val, e0 + (size_t) ((s) + 2) * BATCH);                              \
			__builtin_amdgcn_sched_barrier(0);   /* entry loads BEFORE the gathers: see below */               \
			coob_gather<T, UNIT, K>(XB, QB, x, bb + (size_t) ((s) + 1) * BSTRIDE);                             \
			__builtin_amdgcn_sched_barrier(0);   /* nothing that consumes XA may move above the gathers */     \
			coob_add<T, UNIT, K>(XA, QA, ys, unit);                                                            \
			if (sync_mask & (1 << (k3)))                                                                       \
				__syncthreads();                                                                           \
		}
		for (int s = 0; s < nb; s += 3)
		{
			COOB_STEP(q0, q1, q2, xa, xb, s, 0)
			COOB_STEP(q1, q2, q0, xb, xc, s + 1, 1)
			COOB_STEP(q2, q0, q1, xc, xa, s + 2, 2)
		}
		#undef COOB_STEP
	}
	__syncthreads();
	// local row l = row l % 16 of the workgroup's chunk l / 16 (chunk_row: first row of every chunk the deal gave it, in row order)
	const int r1 = range_row[range + 1];
	const int * __restrict__ cr = chunk_row + chunk_ptr[t];
	const int k0 = range_long[range], nlong = range_long[range + 1] - k0;     // split rows of this range: the last `nlong` LDS slots
	const int nnorm = nloc - nlong;
	for (int l = tid; l < nnorm; l += COOB_THREADS)
	{
		const int row = cr[l / chunk_rows] + l % chunk_rows;
		if (row < r1)
			y[row] = (T) (beta ? (double) y[row] + ys[l] : ys[l]);
	}
	for (int l = tid; l < nlong; l += COOB_THREADS)
		carry[(size_t) (k0 + l) * COOB_WGS + j] = (T) ys[nnorm + l];
}

// adds the 32 partial sums of every split row to y, in workgroup order (deterministic given the partial sums)
template <typename T>
__global__ __launch_bounds__(256) void
coo_blocked_fixup_kernel(const int * __restrict__ long_row, const T * __restrict__ carry, int num_long, T * __restrict__ y)
{
	const int k = (int) (blockIdx.x * 256 + threadIdx.x);
	if (k >= num_long)
		return;
	T sum = 0;
	for (int j = 0; j < COOB_WGS; j++)
		sum += carry[(size_t) k * COOB_WGS + j];
	y[long_row[k]] += sum;
}

int coo_blocked_wgs_per_range() { return COOB_WGS; }
int coo_blocked_chunk_rows() { return COOB_CHUNK; }
int coo_blocked_max_long_rows() { return 64; }     // split rows per range (LDS slots set aside in every workgroup)
int coo_blocked_spare_slots() { return COOB_SPARE; }
int coo_blocked_slot_bits() { return COOB_SLOT_BITS; }
// entries per lane and batch: with a value stream three batches of (entry, value) per lane are live, 4 keep that in registers
int coo_blocked_batch_entries(bool unit) { return (unit ? 8 : 4) * COOB_THREADS; }

int
coo_blocked_rows_cap(bool f32)
{
	// rows of y one workgroup keeps in LDS: the CU's 160 KiB less a little for the runtime, at most what the slot bits index
	(void) f32;                                                    // the LDS copy of y is fp64 for both precisions
	const int bytes = 160 * 1024 - 512;
	int rows = bytes / 8 / COOB_CHUNK * COOB_CHUNK;
	if (const char * e = getenv("SPMV_MI355X_COOB_ROWS"))          // tests: a small cap makes small matrices take several passes
		if (atoi(e) >= COOB_CHUNK && atoi(e) <= rows)
			rows = atoi(e) / COOB_CHUNK * COOB_CHUNK + coo_blocked_max_long_rows() + COOB_SPARE;
	return std::min(rows, (1 << COOB_SLOT_BITS) / COOB_CHUNK * COOB_CHUNK - COOB_CHUNK) - coo_blocked_max_long_rows() - COOB_SPARE;
}

template <typename T, bool UNIT>
static int
coo_blocked_launch(const int * wg_rows, const int * range_row, const int * chunk_ptr, const int * chunk_row, const int * batch_ptr, const int * batch_base, const int * range_long,
		const int * long_row, int num_long, const unsigned * ent, const void * val, const void * x, void * y, void * carry, int num_ranges,
		int chunk_rows, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	constexpr int K = UNIT ? 8 : 4;
	// The layout assumes ONE workgroup per CU (32 per XCD, in step): with less than half of the CU's 160 KiB each, two could share
	// a CU and leave another idle — ask for more than half.
	lds_bytes = std::max(lds_bytes, 84 * 1024);
	// more than 64 KiB of dynamic LDS has to be granted per kernel function, once per device
	static int granted[64] = {0};
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	if (dev >= 0 && dev < 64 && lds_bytes > granted[dev])
	{
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&coo_blocked_kernel<T, UNIT, K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		granted[dev] = lds_bytes;
	}
	const unsigned grid = (unsigned) num_ranges * COOB_WGS;
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	const int per_xcd = num_ranges / NUM_XCD;
	// steps of the period-3 loop that end with a workgroup barrier (bit k = step k): 7 = every batch
	static const int sync_mask = [] { const char * e = getenv("SPMV_MI355X_COOB_SYNC"); return e ? atoi(e) & 7 : 7; }();
	hipLaunchKernelGGL((coo_blocked_kernel<T, UNIT, K>), dim3(grid), dim3(COOB_THREADS), lds_bytes, stream, wg_rows, range_row, chunk_ptr, chunk_row, batch_ptr, batch_base,
			range_long, ent, (const T *) val, (const T *) x, (T *) y, (T *) carry, per_xcd, chunk_rows, sync_mask, (T) (UNIT ? cfg.unit_value : 0), cfg.beta);
	HIP_TRY(hipGetLastError());
	if (num_long > 0)
	{
		hipLaunchKernelGGL((coo_blocked_fixup_kernel<T>), dim3((unsigned) ((num_long + 255) / 256)), dim3(256), 0, stream, long_row, (const T *) carry,
				num_long, (T *) y);
		HIP_TRY(hipGetLastError());
	}
	return 0;
}

int
launch_coo_blocked(bool f32, const int * wg_rows, const int * range_row, const int * chunk_ptr, const int * chunk_row, const int * batch_ptr, const int * batch_base, const int * range_long,
		const int * long_row, int num_long, const unsigned * ent, const void * val, const void * x, void * y, void * carry, int num_ranges,
		int chunk_rows, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	#define COOB_ARGS wg_rows, range_row, chunk_ptr, chunk_row, batch_ptr, batch_base, range_long, long_row, num_long, ent, val, x, y, carry, num_ranges, chunk_rows, lds_bytes, cfg, stream, grid_out
	if (cfg.unit)
		return f32 ? coo_blocked_launch<float, true>(COOB_ARGS) : coo_blocked_launch<double, true>(COOB_ARGS);
	return f32 ? coo_blocked_launch<float, false>(COOB_ARGS) : coo_blocked_launch<double, false>(COOB_ARGS);
	#undef COOB_ARGS
}

}  // namespace spmv
