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
// fabric's random-sector rate: every gather misses the 4 MiB L2 of its XCD and most of the 64 bytes it brings are unused.
// MI355X has 160 KiB of LDS on each of its 256 CUs — 40 MiB in all, as much as the whole y of such a matrix — and eight L2s
// that each serve 32 CUs. The layout below is built around both:
//   * the rows are cut into 8*P contiguous RANGES (P = 1 while y fits the chip's LDS); XCD k owns ranges k*P .. k*P+P-1 and
//     works through them one after the other;
//   * a range is dealt to 32 workgroups (one per CU, 1024 threads) in chunks of 16 rows, round-robin: every workgroup of the
//     range sees the same column structure and the same number of entries per column block, and its y rows (<= 160 KiB)
//     live in LDS for the whole launch, written back once as full 128-byte lines;
//   * a workgroup's entries are ordered by column block (<= 65 536 columns of x, default 384 KiB) and, inside a block, by
//     column: the 32 workgroups of an XCD sweep the same block at the same time, so the block is fetched into that L2 once,
//     and neighbouring lanes gather from neighbouring lines (several entries per 128-byte line instead of one);
//   * an entry is ONE dword: column inside the block (16 bits) | row inside the workgroup (16 bits), + the value unless all
//     values are equal (Matrix-Market `pattern` matrices): 4 bytes per non-zero of stream instead of CSR's 12;
//   * v*x[col] is added into the LDS copy of y with ds_add (LDS atomics): the order of a row's additions is run-dependent,
//     parity is to the tolerance (1e-12 / 1e-5), not bit-wise.
//   * a row too long for one workgroup's share (a hub of a power-law graph) is SPLIT over the 32 workgroups of its range, each
//     summing a contiguous piece into an extra LDS slot; the 32 partial sums go to a carry array and a fix-up kernel adds
//     them to y in workgroup order — the merge-path remedy (merge.cpp:302-318: partial rows + carry fix-up) inside this layout.
// The next block's first batch of entries is in flight while the current block is consumed; a workgroup barrier after
// every block keeps the 16 waves on the same block of x.
constexpr int COOB_THREADS = 1024;
constexpr int COOB_U = 8;                  // entries per lane and batch
constexpr int COOB_WGS = 32;               // workgroups per range = CUs per XCD
constexpr int COOB_CHUNK = 16;             // rows per chunk of the round-robin deal

template <typename T, bool UNIT>
struct CoobBatch {
	unsigned e[COOB_U];
	T v[UNIT ? 1 : COOB_U];
};

// Branch-free: the entry arrays carry COOB_SLACK spare entries, so a lane past the end of its block loads something valid and
// discards it. Straight-line loads let the compiler count the outstanding memory operations exactly (s_waitcnt vmcnt(N));
// with a branch around every load it fell back to vmcnt(0) before each gather and the prefetch below was waited for at once.
template <typename T, bool UNIT>
__device__ __forceinline__ void
coob_load(CoobBatch<T, UNIT> & b, const unsigned * __restrict__ ent, const T * __restrict__ val, int first)
{
	#pragma unroll
	for (int u = 0; u < COOB_U; u++)
	{
		// the entry streams are read once: nontemporal, so that they do not push the x block out of L2
		b.e[u] = ld_stream<true>(ent + first + u * COOB_THREADS);
		if constexpr (!UNIT)
			b.v[u] = ld_stream<true>(val + first + u * COOB_THREADS);
	}
}

// Lanes past the end of the block issue NO gather: the vector memory pipeline handles a gather lane by lane (PMC: one L1
// access per lane), and dummy gathers for the idle lanes cost 316 instead of 244 us on the soc-LiveJournal1 twin.
template <typename T, bool UNIT>
__device__ __forceinline__ void
coob_gather(T (&xv)[COOB_U], const CoobBatch<T, UNIT> & b, const T * __restrict__ xb, int first, int end)
{
	#pragma unroll
	for (int u = 0; u < COOB_U; u++)
	{
		xv[u] = 0;
		if (first + u * COOB_THREADS < end)
			xv[u] = xb[b.e[u] >> 16];
	}
}

// The LDS copy of y is fp64 for both precisions: ds_add_f32 runs at half the rate of ds_add_f64 on this part (fp32 on the
// soc-LiveJournal1 twin: 487 us with float atomics, 259 us with plain stores in their place), and the sums are more accurate.
template <typename T, bool UNIT>
__device__ __forceinline__ void
coob_add(const T (&xv)[COOB_U], const CoobBatch<T, UNIT> & b, double * __restrict__ ys, T unit, int first, int end)
{
	#pragma unroll
	for (int u = 0; u < COOB_U; u++)
		if (first + u * COOB_THREADS < end)
			unsafeAtomicAdd(&ys[b.e[u] & 0xffffu], (double) (UNIT ? unit : b.v[u]) * (double) xv[u]);
}

// One column block of one workgroup: gathers of the current batch first, THEN the prefetch of the batch two blocks ahead
// (vector memory operations complete in issue order: everything issued before the gathers would have to land before they do),
// then the LDS adds, then whatever the block holds beyond one batch.
template <typename T, bool UNIT>
__device__ __forceinline__ void
coob_step(const CoobBatch<T, UNIT> & cur, CoobBatch<T, UNIT> & pf, const unsigned * __restrict__ ent, const T * __restrict__ val,
		const T * __restrict__ xb, double * __restrict__ ys, T unit, int e0, int e1, int pf_first, int tid)
{
	T xv[COOB_U];
	coob_gather<T, UNIT>(xv, cur, xb, e0 + tid, e1);
	coob_load<T, UNIT>(pf, ent, val, pf_first + tid);
	coob_add<T, UNIT>(xv, cur, ys, unit, e0 + tid, e1);
	for (int e = e0 + tid + COOB_U * COOB_THREADS; e < e1; e += COOB_U * COOB_THREADS)
	{
		CoobBatch<T, UNIT> more;
		coob_load<T, UNIT>(more, ent, val, e);
		coob_gather<T, UNIT>(xv, more, xb, e, e1);
		coob_add<T, UNIT>(xv, more, ys, unit, e, e1);
	}
}

template <typename T, bool UNIT>
__global__ __launch_bounds__(COOB_THREADS) void
coo_blocked_kernel(const int * __restrict__ wg_rows, const int * __restrict__ range_row, const int * __restrict__ seg_blk,
		const int * __restrict__ range_blk, const int * __restrict__ range_long, const unsigned * __restrict__ ent,
		const T * __restrict__ val, const T * __restrict__ x, T * __restrict__ y, T * __restrict__ carry,
		int ranges_per_xcd, int max_blocks, int chunk_rows, int sync_every, T unit, int beta)
{
	extern __shared__ __align__(16) unsigned char coob_smem[];
	double * ys = reinterpret_cast<double *>(coob_smem);
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
	for (int l = tid; l < nloc; l += COOB_THREADS)
		ys[l] = 0;
	const int * sb = seg_blk + (size_t) t * (max_blocks + 1);
	const int * bc = range_blk + (size_t) range * (max_blocks + 1);        // [0] = blocks of this range, [1 + b] = first column of block b
	const int b0 = 0, b1 = bc[0];
	auto xblk = [&](int b) { return x + bc[1 + (b < b1 ? b : b1 - 1)]; };
	// entry offset of block b, clamped to the last boundary (blocks past the range's last hold nothing)
	auto at = [&](int b) { return sb[b < b1 ? b : b1]; };
	CoobBatch<T, UNIT> q0, q1, q2;
	if (b0 < b1)
	{
		coob_load<T, UNIT>(q0, ent, val, at(b0) + tid);
		coob_load<T, UNIT>(q1, ent, val, at(b0 + 1) + tid);
	}
	__syncthreads();
	int since = 0;
	// three batches rotate through q0, q1, q2 (compile-time names: registers, not an indexed array)
	for (int b = b0; b < b1; b += 3)
	{
		coob_step<T, UNIT>(q0, q2, ent, val, xblk(b), ys, unit, at(b), at(b + 1), at(b + 2), tid);
		if (++since == sync_every) { since = 0; __syncthreads(); }    // keep the waves of the workgroup on the same column blocks
		if (b + 1 < b1)
		{
			coob_step<T, UNIT>(q1, q0, ent, val, xblk(b + 1), ys, unit, at(b + 1), at(b + 2), at(b + 3), tid);
			if (++since == sync_every) { since = 0; __syncthreads(); }
		}
		if (b + 2 < b1)
		{
			coob_step<T, UNIT>(q2, q1, ent, val, xblk(b + 2), ys, unit, at(b + 2), at(b + 3), at(b + 4), tid);
			if (++since == sync_every) { since = 0; __syncthreads(); }
		}
	}
	__syncthreads();
	// local row l = chunk (l / 16) of this workgroup, row l % 16 of the chunk; its chunks are j, j + 32, j + 64, ... of the range
	const int r0 = range_row[range], r1 = range_row[range + 1];
	const int k0 = range_long[range], nlong = range_long[range + 1] - k0;     // split rows of this range: the last `nlong` LDS slots
	const int nnorm = nloc - nlong;
	for (int l = tid; l < nnorm; l += COOB_THREADS)
	{
		const int row = r0 + ((l / chunk_rows) * COOB_WGS + j) * chunk_rows + l % chunk_rows;
		if (row < r1)
			y[row] = (T) (beta ? (double) y[row] + ys[l] : ys[l]);
	}
	if (tid < nlong)
		carry[(size_t) (k0 + tid) * COOB_WGS + j] = (T) ys[nnorm + tid];
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
int coo_blocked_batch_entries() { return COOB_U * COOB_THREADS; }         // entries one workgroup takes per batch
int coo_blocked_entry_slack() { return COOB_U * COOB_THREADS + 64; }      // spare entries behind the entry arrays (branch-free loads)

int
coo_blocked_rows_cap(bool f32)
{
	// rows of y one workgroup keeps in LDS: the CU's 160 KiB less a little for the runtime, at most what 16 bits index
	(void) f32;                                                    // the LDS copy of y is fp64 for both precisions
	const int bytes = 160 * 1024 - 512;
	int rows = bytes / 8 / COOB_CHUNK * COOB_CHUNK;
	if (const char * e = getenv("SPMV_MI355X_COOB_ROWS"))          // tests: a small cap makes small matrices take several passes
		if (atoi(e) >= COOB_CHUNK && atoi(e) <= rows)
			rows = atoi(e) / COOB_CHUNK * COOB_CHUNK + coo_blocked_max_long_rows();
	return std::min(rows, 65536 / COOB_CHUNK * COOB_CHUNK - COOB_CHUNK) - coo_blocked_max_long_rows();
}

template <typename T>
static int
coo_blocked_launch(const int * wg_rows, const int * range_row, const int * seg_blk, const int * range_blk, const int * range_long,
		const int * long_row, int num_long, const unsigned * ent, const void * val, const void * x, void * y, void * carry, int num_ranges,
		int num_blocks, int block_cols, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	// The layout assumes ONE workgroup per CU (32 per XCD, in step): with less than half of the CU's 160 KiB each, two could share
	// a CU and leave another idle — ask for more than half.
	lds_bytes = std::max(lds_bytes, 84 * 1024);
	// more than 64 KiB of dynamic LDS has to be granted per kernel function, once per device
	static int granted[64] = {0};
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	if (dev >= 0 && dev < 64 && lds_bytes > granted[dev])
	{
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&coo_blocked_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&coo_blocked_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		granted[dev] = lds_bytes;
	}
	const unsigned grid = (unsigned) num_ranges * COOB_WGS;
	if (grid_out)
		*grid_out = grid;
	if (grid == 0)
		return 0;
	const int per_xcd = num_ranges / NUM_XCD;
	// column blocks between workgroup barriers (measured on the soc-LiveJournal1 twin: 1, 2 and 4 within 2 %, never: +5 %)
	static const int sync_every = [] { const char * e = getenv("SPMV_MI355X_COOB_SYNC"); return e ? atoi(e) : 1; }();
	if (cfg.unit)
		hipLaunchKernelGGL((coo_blocked_kernel<T, true>), dim3(grid), dim3(COOB_THREADS), lds_bytes, stream, wg_rows, range_row, seg_blk, range_blk,
				range_long, ent, (const T *) nullptr, (const T *) x, (T *) y, (T *) carry, per_xcd, num_blocks, block_cols, sync_every, (T) cfg.unit_value, cfg.beta);
	else
		hipLaunchKernelGGL((coo_blocked_kernel<T, false>), dim3(grid), dim3(COOB_THREADS), lds_bytes, stream, wg_rows, range_row, seg_blk, range_blk,
				range_long, ent, (const T *) val, (const T *) x, (T *) y, (T *) carry, per_xcd, num_blocks, block_cols, sync_every, (T) 0, cfg.beta);
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
launch_coo_blocked(bool f32, const int * wg_rows, const int * range_row, const int * seg_blk, const int * range_blk, const int * range_long,
		const int * long_row, int num_long, const unsigned * ent, const void * val, const void * x, void * y, void * carry, int num_ranges,
		int num_blocks, int block_cols, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out)
{
	return f32 ? coo_blocked_launch<float>(wg_rows, range_row, seg_blk, range_blk, range_long, long_row, num_long, ent, val, x, y, carry, num_ranges,
	                                       num_blocks, block_cols, lds_bytes, cfg, stream, grid_out)
	           : coo_blocked_launch<double>(wg_rows, range_row, seg_blk, range_blk, range_long, long_row, num_long, ent, val, x, y, carry, num_ranges,
	                                        num_blocks, block_cols, lds_bytes, cfg, stream, grid_out);
}

}  // namespace spmv
