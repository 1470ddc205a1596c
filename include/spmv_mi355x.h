/* spmv_mi355x.h — C ABI of the MI355X (gfx950 / CDNA4) SpMV engine.
 *
 * This is the drop-in boundary for the SpMV hot path of LiHaoxu/SpMV-Research: the shared object
 * libspmv_mi355x.so (hand-written HIP, hipcc --offload-arch=gfx950) exports exactly what a backend TU of the
 * reference harness needs behind its plug-in API
 *
 *     struct Matrix_Format * csr_to_format(INT_T * row_ptr, INT_T * col_ind, ValueTypeReference * values,
 *                                          long m, long n, long nnz, long symmetric, long symmetry_expanded);
 *     virtual void Matrix_Format::spmv(ValueType * x, ValueType * y);
 *         (benchmark_code/BENCH/src/spmv_kernels/spmv_kernel.h:8-29)
 *
 * with plain pointers, sizes and opaque handles, so the host TU is compiled by g++ with no HIP header
 * (precedent for the split: GPU_clean/csr_rocm_vector.cpp:215,239 `extern "C" launch_kernel_wrapper`).
 * The adapter TU that binds this ABI to Matrix_Format is spmv-research_amd/host/spmv_kernel_mi355x.cpp;
 * INTEGRATION.md shows the Makefile_in rule a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; spmv_mi355x_last_error() gives the message
 *     (the Matrix_Format adapter turns a failure into the reference's error()+exit behaviour, lib/debug.h:83-135);
 *   - single caller thread per handle, blocking on return unless the name ends in _async / takes a stream;
 *   - indices are int32 (INT_T, make.sh:166), values arrive as fp64 regardless of precision (Q3, bench.cpp:601)
 *     and are narrowed on upload when precision == SPMV_MI355X_F32 (csr.cpp:72 does the same);
 *   - inputs are deep-copied: the caller may free them right after create (bench.cpp:605-629);
 *   - there is NO CPU fallback: without a usable gfx950 device create() fails.
 */
#ifndef SPMV_MI355X_H
#define SPMV_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmv_mi355x_matrix spmv_mi355x_matrix;   /* opaque */

/* storage format + kernel of y = A*x */
enum {
	SPMV_MI355X_CSR_SCALAR   = 0,  /* one lane per row; bit-identical to the reference CPU kernel (csr.cpp:334-350)      */
	SPMV_MI355X_CSR_VECTOR   = 1,  /* one group of 2..64 lanes per row (64 = one wavefront per row); replaces
	                                  GPU_clean/spmv_subkernel_csr_rocm_vector.cpp:5-54 / CPU analogue csr_vec.cpp:182-213 */
	SPMV_MI355X_CSR_MERGE    = 2,  /* merge-path CSR; replaces merge.cpp:256-319 / GPU_clean/merge_cuda.cu:249-261        */
	SPMV_MI355X_SELL_C_SIGMA = 3,  /* sliced ELL, C rows per slice, sigma-window sort; replaces sell_sorted.cpp:112-419 /
	                                  sell_c_s.cpp:39-131                                                                  */
	SPMV_MI355X_COO          = 4,  /* row-sorted COO, segmented reduction; replaces mkl_coo.cpp:58-106 /
	                                  GPU_clean/rocsparse_coo.cpp:88-104,294                                               */
	SPMV_MI355X_CSR_STREAM   = 5,  /* one wavefront per block of R consecutive rows, products staged in LDS (CSR-Stream);
	                                  replaces the row-block CSR kernels GPU_clean/spmv_subkernel_csr_rocm_adaptive.cpp:76-153 */
	SPMV_MI355X_NUM_FORMATS  = 6
};

enum { SPMV_MI355X_F64 = 0, SPMV_MI355X_F32 = 1 };

/* Tunables. Zero-initialise, set struct_size = sizeof(spmv_mi355x_opts); 0 / unset = engine default. */
typedef struct {
	int  struct_size;
	int  device;            /* HIP device ordinal; -1 = the current device                                        */
	int  lanes_per_row;     /* CSR_VECTOR: 2,4,8,16,32,64; 0 = chosen from mean nnz/row                            */
	                        /* CSR_STREAM: the same field holds ROWS PER WAVEFRONT (4,8,16,32,64); 0 = auto          */
	int  sell_split;        /* SELL delta format: wavefronts sharing one 64-row slice (1, 2 or 4); 0 = auto by slice count */
	int  sell_c;            /* SELL: rows per slice (16, 32, 64, or 256 = the BSC library's, sell_c_s.cpp:58-60); 0 = 64         */
	int  sell_sigma;        /* SELL: sort window in rows (multiple of sell_c); 0 = 16384 (sell_c_s.cpp:58-60)      */
	int  merge_items;       /* MERGE: merge items per thread (5,7,9,11,13); COO: entries per lane (2,4,8); 0 = default */
	int  xcd_remap;         /* tile order over the 8 XCDs: 0 = auto, 1 = contiguous work-balanced ranges, 2 = off,
	                           3 = chunks of 64 tiles dealt round-robin to the XCDs                                */
	int  nontemporal;       /* matrix streams loaded with the nt policy: 0 = auto (by footprint), 1 = on, 2 = off  */
	int  stream_mode;       /* CSR_STREAM: 0 = auto (3); 1 = products staged in LDS (row-major x gather); 2 = (value,column)
	                           pairs staged in LDS through registers, lanes walk rows (coalesced x gathers); 3 = the same with
	                           the global->LDS copy done by LDS-DMA (global_load_lds). With 64 rows per wave 2 and 3 are
	                           bit-exact; 4 = nnz-balanced row blocks whose window of x is copied into LDS (up to 128 KiB) and
	                           gathered from there, lanes_per_row lanes per row (8..64), merge_items = blocks per CU (0 auto);
	                           for matrices whose row blocks touch a narrow column range (FEM / banded)                   */
	long row_begin;         /* row block [row_begin,row_end) of the GLOBAL CSR to keep on this device (row-partitioned */
	long row_end;           /*   multi-GPU, §8e); 0,0 = all rows. x stays full length n; y has row_end-row_begin rows. */
	long col_begin;         /* optional column filter [col_begin,col_end) used for the local/remote split that lets    */
	long col_end;           /*   the local part run while allgather(x) is in flight; 0,0 = no filter                   */
	int  col_filter_mode;   /* 0 = keep all, 1 = keep columns inside [col_begin,col_end), 2 = keep columns outside     */
	int  sell_delta;        /* SELL with 64-row slices: column indices stored as one base per step + 8/16-bit deltas per lane
	                           where they fit (lossless, bit-identical results): 0 = auto (on when sell_c = 64), 1 = on, 2 = off */
	int  convert_on;        /* where the layouts with a GPU builder (SELL plain / delta / LDS-window: csrc/convert_sell.hip; the entry arrays of
	                           the column-blocked layout: csrc/convert_coo.hip) are built from the CSR: 0 = auto (GPU), 1 = GPU,
	                           2 = host (OpenMP; kept as the checker — both produce the same bytes)                    */
	int  symmetric_input;   /* 1 = the CSR arrays hold ONE triangle of a symmetric matrix (KEEP_SYMMETRY builds of the harness:
	                           csr_to_format(..., symmetric = 1, symmetry_expanded = 0), csr_sym.cpp:118-123); the product is
	                           y = (T + T^t - diag T) x; rows()/nnz() report the expanded matrix. SELL_C_SIGMA on a banded matrix
	                           (every slice group's window of rows + columns fits LDS: (w + 1) * (sizeof(V) + 8) <= 152 KiB) keeps the
	                           triangle and multiplies without expanding it — half the matrix stream, the mirrored additions as LDS
	                           atomics (csrc/kernels_sell_window.hip) — on its own when the expanded stream exceeds the 256 MiB
	                           Infinity Cache, always with sell_window = 1. Everything else expands the triangle at create().        */
	int  rows_per_group;    /* CSR_VECTOR: consecutive rows a lane group keeps in flight together (1, 2 or 4; 2 and 4 need
	                           lanes_per_row >= 8); 0 = auto                                                        */
	int  col_blocks;        /* COO: 0 = row-sorted COO (the reference's layout); -1 = column-blocked layout for graph matrices: the rows are
	                           dealt to workgroups that keep their y in LDS, a workgroup's entries are sorted by column and stored as one
	                           dword each in full batches (csrc/kernels_coo.hip); > 0 = the same with a wave instruction's 64 entries kept
	                           inside ceil(n / col_blocks) columns (tests). LDS atomics: sums to the tolerance, not bit-reproducible.
	                           CSR_MERGE: 0 / -2 = the CSR-order merge path (deterministic); -1 / > 0 = the same column-blocked layout
	                           with merge-path-balanced row ranges — never chosen unless asked for                                  */
	int  sell_window;       /* SELL, 64-row slices: a workgroup owns a group of consecutive slices, copies the group's column window of
	                           x into LDS and gathers from there; column indices are 16-bit offsets into the window (for banded /
	                           FEM matrices; csrc/kernels_sell_window.hip). 0 = auto (when sell_sigma, sell_delta and convert_on are at their defaults
	                           and every group's window fits; rows are then sorted inside a slice group: sigma = 64 * sell_group), 1 = on, 2 = off.
	                           spmv_mi355x_sell_layout() does not decode this layout: name a sell_sigma (or sell_window = 2) for layout parity */
	int  kahan;             /* CSR_SCALAR: 1 = Kahan-compensated row sums, the reference's CUSTOM_KAHAN build (csr.cpp:353-373); same
	                           operations in the same order -> bit-identical to it                                             */
	int  sell_group;        /* sell_window: slices per workgroup (1, 2, 4, 8 or 16; times sell_split at most 16 wavefronts); 0 = auto */
	int  placement;         /* where the handle's vectors live relative to its value array ("vectors placed by the engine" below):
	                           0 = off (default: plain allocations), 1 = on (slices of the device's vector pools; the first handle
	                           of a process that asks makes ONE walk through the device's free memory to find them), 2 = off,
	                           3 = 1 + a search over the handle's matrix arrays (worth 1-5 %, ~500 launches).
	                           SPMV_MI355X_PLACEMENT in the environment overrides: 0 off, 1 on, 2 on + log on stderr, 3, 4 (diagnostic) */
	int  placement_budget_gib;  /* transient memory the walk may hold, GiB (0 = 160; it never takes the device's last 8 GiB)             */
} spmv_mi355x_opts;

/* ---- library / device ------------------------------------------------------------------------------------ */
const char * spmv_mi355x_last_error(void);
int  spmv_mi355x_device_count(int * count_out);
int  spmv_mi355x_device_info(int device, char * name_out, long name_n, int * compute_units_out, long * hbm_bytes_out);

/* ---- construction = csr_to_format() (csr.cpp:221-240 and the per-format constructors) ------------------------ */
int  spmv_mi355x_create(spmv_mi355x_matrix ** out, int format, int precision,
		long m, long n, long nnz,
		const int32_t * row_ptr, const int32_t * col_idx, const double * values_fp64,
		const spmv_mi355x_opts * opts /* may be NULL */);
int  spmv_mi355x_destroy(spmv_mi355x_matrix * A);

/* Matrix_Format fields (spmv_kernel.h:10-15,23) */
const char * spmv_mi355x_format_name(const spmv_mi355x_matrix * A);
double spmv_mi355x_mem_footprint(const spmv_mi355x_matrix * A);      /* bytes of the device-side format            */
double spmv_mi355x_csr_mem_footprint(const spmv_mi355x_matrix * A);  /* nnz*(sizeof(V)+4)+(m+1)*4                  */
long   spmv_mi355x_rows(const spmv_mi355x_matrix * A);               /* local rows (row block)                      */
long   spmv_mi355x_cols(const spmv_mi355x_matrix * A);
long   spmv_mi355x_nnz(const spmv_mi355x_matrix * A);                /* local non-zeros after row/column filtering  */
int    spmv_mi355x_precision(const spmv_mi355x_matrix * A);          /* SPMV_MI355X_F64 / SPMV_MI355X_F32           */
int    spmv_mi355x_device(const spmv_mi355x_matrix * A);             /* HIP device ordinal the handle lives on      */

/* ---- a handle from a CSR that arrives in pieces ---------------------------------------------------------------------- */
/* spmv_mi355x_create() needs the whole CSR in host memory at once. A caller that generates or reads its rows piece by piece (a rank
 * of a multi-GPU run: bench.py --gpus N) appends them to a CSR kept in DEVICE memory — every piece is checked like create() checks a
 * matrix (row_ptr from 0 and monotone, columns in [0, n)) — and gets ONE handle converted on the GPU from the resident CSR; the host
 * never holds more than a piece. nnz_capacity: an upper bound of the non-zeros to come (device arrays of that size live until
 * create_from_stream). Pieces are consecutive rows in order; a piece's row_ptr has rows + 1 entries starting at 0.
 * create_from_stream consumes the stream (also on failure). Formats: SPMV_MI355X_SELL_C_SIGMA with 64-row slices and the delta
 * layout (what create() picks for large matrices; same arrays, same results); opts fields of other layouts are rejected.
 * No reference counterpart: csr_to_format() receives complete arrays (spmv_kernel.h:8-29). */
typedef struct spmv_mi355x_csr_stream spmv_mi355x_csr_stream;
int  spmv_mi355x_csr_stream_begin(spmv_mi355x_csr_stream ** out, int device /* -1: current */, long m, long n, long nnz_capacity);
int  spmv_mi355x_csr_stream_append(spmv_mi355x_csr_stream * s, long rows, const int32_t * row_ptr, const int32_t * col_idx,
		const double * values);
int  spmv_mi355x_create_from_stream(spmv_mi355x_matrix ** out, spmv_mi355x_csr_stream * s, int format, int precision,
		const spmv_mi355x_opts * opts);
int  spmv_mi355x_csr_stream_discard(spmv_mi355x_csr_stream * s);

/* ---- Matrix_Format::spmv(x, y) with HOST buffers --------------------------------------------------------- */
/* Reference GPU-backend semantics (GPU_clean/csr_rocm_vector.cpp:224-257): x is uploaded when the host
 * pointer is new (or always_copy is set), one launch + device sync, y is downloaded on the first call (or when
 * always_copy is set). x: n values, y: rows values, both of the handle's precision. */
int  spmv_mi355x_spmv(spmv_mi355x_matrix * A, const void * x_host, void * y_host);
int  spmv_mi355x_set_always_copy(spmv_mi355x_matrix * A, int on);   /* for callers whose x changes (bench_cg.cpp)  */
int  spmv_mi355x_upload_x(spmv_mi355x_matrix * A, const void * x_host);
int  spmv_mi355x_download_y(spmv_mi355x_matrix * A, void * y_host);

/* ---- device-pointer entry points (solvers, multi-GPU, benchmarks) ---------------------------------------- */
/* y_dev = A * x_dev (beta == 0) or y_dev += A * x_dev (beta == 1), enqueued on hip_stream (a hipStream_t passed
 * as void*, NULL = the default stream); returns after enqueue. */
int  spmv_mi355x_spmv_device_async(spmv_mi355x_matrix * A, const void * x_dev, void * y_dev, int beta, void * hip_stream);
/* Time `iters` back-to-back launches with HIP events recorded on the stream the kernels run on; ms per iteration. */
int  spmv_mi355x_time_device(spmv_mi355x_matrix * A, const void * x_dev, void * y_dev, int iters,
		void * hip_stream, double * ms_per_iter_out);
/* Name and launch shape of the dominant kernel (for matching rocprofv3 kernel-trace rows). */
int  spmv_mi355x_kernel_info(const spmv_mi355x_matrix * A, char * name_out, long name_n, long * grid_out, int * block_out);

/* dst_dev[0..bytes) = src_dev[0..bytes), enqueued on hip_stream: lets a caller without HIP headers (the ctypes / cgo side of
 * the distributed solver callbacks) move a vector slice into its exchange buffer. */
int  spmv_mi355x_copy_device_async(void * dst_dev, const void * src_dev, long bytes, void * hip_stream);

/* Device buffers owned by the handle (allocated lazily by the host-buffer entry points): x has n values, y has rows + 64
 * (the reference driver's slack, bench_spmv.cpp:606-609). With opts.placement = 1 both are placed by the engine (below). */
void * spmv_mi355x_x_device(spmv_mi355x_matrix * A);
void * spmv_mi355x_y_device(spmv_mi355x_matrix * A);
int  spmv_mi355x_upload_y(spmv_mi355x_matrix * A, const void * y_host);      /* rows values into the handle's y (for y += A x) */

/* ---- vectors placed by the engine --------------------------------------------------------------------------------- */
/* The 288 GiB of an MI355X behave as 32 GiB blocks that fall into classes, and the same kernel on the same matrix and x takes 1.28 or
 * 1.46 ms (nlpkkt240 twin) depending only on whether y lives in a block of the same class as the value array; which memory an
 * allocation gets is the driver's choice (profiles/r02_placement.md). With opts.placement = 1 (or SPMV_MI355X_PLACEMENT
 * >= 1) the engine keeps, per process and device, two to four VECTOR POOLS (1-4 GiB each) in blocks of different class: the first
 * handle that needs a vector of 8 MiB or more walks the device's free memory once — candidates 16 GiB of ballast apart, timed with the
 * handle's own kernel; one that differs by 1.5 % from every candidate kept so far is kept; the walk ends three candidates after the
 * last new class; at most opts.placement_budget_gib (160) of ballast, returned when the walk ends — and every later vector of any
 * handle is a slice of the pool in which that handle's kernel runs fastest (one trial of six launches per pool). The handle's own pair (spmv_mi355x_x_device / y_device, used by spmv_mi355x_spmv) is placed this way; output_alloc /
 * input_alloc give callers of the device-pointer entry points the same for vectors the handle's SpMV writes / reads (bytes >= (rows +
 * 64) resp. cols values; smaller, under 8 MiB or with placement off: a plain allocation). Zero-filled. Free with output_free.
 * OFF by default: the walk holds tens of GiB for a fraction of a second, its free-memory check is racy against other processes on the
 * same GPU, and the driver clears the returned ballast in the background for a few seconds, during which any process's kernels on that
 * GPU run up to 5 % slower (profiles/r02_placement.md §6). placement_release frees a device's pools (no vector of them may be live).
 * No reference counterpart (the reference's GPU backends hipMalloc their vectors in the constructor, GPU_clean/csr_rocm_vector.cpp:77-86). */
int  spmv_mi355x_output_alloc(spmv_mi355x_matrix * A, size_t bytes, void ** out);
int  spmv_mi355x_input_alloc(spmv_mi355x_matrix * A, size_t bytes, void ** out);
int  spmv_mi355x_output_free(void * p);
/* The search over the handle's MATRIX arrays that opts.placement = 3 runs for the handle's own vector pair, for a caller's pair: every
 * array of 16 MiB .. 8 GiB is tried at up to ten sites 16 GiB of ballast apart (a device copy and six launches of y = A x per trial; y is
 * overwritten) and stays where the kernel ran fastest if that beats where it was by 2 %. Same budget and the same caveats as the walk. */
int  spmv_mi355x_place_arrays(spmv_mi355x_matrix * A, const void * x_dev, void * y_dev);
int  spmv_mi355x_placement_release(int device /* -1: every device */);
/* what the walk of a device found: state 0 = none made yet, 1 = pools of different block class kept, 2 = no contrast inside the budget
 * (plain allocations); candidates timed, GiB of ballast held at its deepest, the number of pools, the walking handle's kernel time (us)
 * with y in each */
int  spmv_mi355x_placement_info(int device, int * state_out, int * candidates_out, long * walked_gib_out, int * pools_out, double us_out[4]);

/* ---- solver callers of spmv() ---------------------------------------------------------------- */
/* Device-resident replacements for the reference's two Krylov drivers, which call MF->spmv() with a vector that changes
 * every iteration:  spmv_mi355x_pcg       = preconditioned_cg()        benchmark_code/BENCH/src/bench_cg.cpp:93-322
 *                   spmv_mi355x_pbicgstab = preconditioned_bicgstab()  benchmark_code/BENCH/src/bench_bicg.cpp:149-459
 * Arguments as in the reference: the handle (MF), the host CSR arrays the Jacobi preconditioner K = diag(A) is read from
 * (values in fp64 = ValueTypeReference, like spmv_mi355x_create), b and x_res_out as HOST arrays of the handle's
 * precision, max_iterations (CG_MAX_NUM_ITERS). Same semantics: x0 = 0, eps = 1e-15*|b|, explicit residual every 100
 * iterations with best-x tracking, CG restart rule and `err < eps` break, BiCGSTAB never breaks; errors "bad K, zero in
 * diagonal" and "the matrix must be square" are returned (rc 1 + last_error) instead of exit(1).
 * history_out (may be NULL): 3*max_iterations doubles; row k = error, error_explicit, error_best as the reference prints
 * them at iteration k (bench_cg.cpp:249); rows >= info->iterations stay 0. */
typedef struct {
	unsigned struct_size;      /* in: sizeof(spmv_mi355x_solver_info) */
	long   iterations;         /* num_loops_out (bench_cg.cpp:315) */
	double error;              /* |b - A*x_res_out|, the CSV "error" column (bench_cg.cpp:412-418) */
	double error_best;         /* err_best: smallest explicit residual seen = the one of x_res_out */
	double eps, eps_counter;   /* 1e-15*|b|, 1e-7*|b| (bench_cg.cpp:159-174) */
	long   restarts;           /* CG restarts taken (bench_cg.cpp:219-235) */
	long   spmv_calls;         /* SpMV launches the solver made */
	double seconds;            /* wall time of the whole call = the CSV "time" column */
} spmv_mi355x_solver_info;
int  spmv_mi355x_pcg(spmv_mi355x_matrix * A, const int32_t * row_ptr, const int32_t * col_idx, const double * values_fp64,
		const void * b_host, void * x_res_out_host, long max_iterations, double * history_out, spmv_mi355x_solver_info * info);
int  spmv_mi355x_pbicgstab(spmv_mi355x_matrix * A, const int32_t * row_ptr, const int32_t * col_idx, const double * values_fp64,
		const void * b_host, void * x_res_out_host, long max_iterations, double * history_out, spmv_mi355x_solver_info * info);

/* Row-partitioned (multi-GPU) form of the same two solvers: one process per GPU owns the row block [row_offset,
 * row_offset + m_local) of A, b and x. The solver keeps every vector device-resident and local; the two things that cross
 * ranks are handed to the caller, who has the communicator (torch.distributed / RCCL in bench-level code):
 *   spmv(ctx, in_dev, out_dev)           out_local = (A * in)_local where `in` is this rank's slice of the global vector
 *                                        (exchange of the slices + the local SpMV launches, e.g. §8e's allgather(x) scheme)
 *   allreduce_sum(ctx, reduce_buf_dev, count)   in-place sum over the ranks of `count` doubles in reduce_buf_dev
 * Both are called on the host between kernel launches and must ENQUEUE their work on the NULL stream (or order against
 * it); nothing waits for the host. Per iteration: CG 1 spmv + 2 all-reduces (1 and 2 doubles), BiCGSTAB 2 spmv + 3
 * all-reduces. All ranks compute identical scalars, take the `err < eps` break at the same iteration and return the same
 * history/info; the values equal the single-GPU solver's up to the summation order of the dots.
 * row_ptr_local has m_local+1 entries starting at 0; col_idx_global holds GLOBAL column indices (the Jacobi diagonal of
 * local row i is the first entry with column row_offset + i); b / x are the local slices (host, handle precision). */
typedef struct {
	unsigned struct_size;      /* sizeof(spmv_mi355x_dist_ops) */
	long   row_offset;
	int  (*spmv)(void * ctx, const void * in_dev, void * out_dev);
	int  (*allreduce_sum)(void * ctx, double * reduce_buf_dev, int count);
	double * reduce_buf_dev;   /* device scratch of >= 4 doubles owned by the caller (so it can be a tensor of its framework) */
	void * ctx;
} spmv_mi355x_dist_ops;
int  spmv_mi355x_pcg_dist(const spmv_mi355x_dist_ops * ops, int precision, long m_local, const int32_t * row_ptr_local,
		const int32_t * col_idx_global, const double * values_fp64, const void * b_local_host, void * x_local_out_host,
		long max_iterations, double * history_out, spmv_mi355x_solver_info * info);
int  spmv_mi355x_pbicgstab_dist(const spmv_mi355x_dist_ops * ops, int precision, long m_local, const int32_t * row_ptr_local,
		const int32_t * col_idx_global, const double * values_fp64, const void * b_local_host, void * x_local_out_host,
		long max_iterations, double * history_out, spmv_mi355x_solver_info * info);

/* ---- several GPUs of one node behind ONE handle (SURVEY §8b "create_partitioned", §8e) ------------------------------------- */
/* What the reference's single-process driver can call: csr_to_format() hands over the whole matrix (bench.cpp:600-603), the
 * library cuts it into `nparts` nnz-balanced contiguous row blocks — loop_partitioner_balance_prefix_sums
 * (lib/parallel_util.h:156-184, what csr.cpp:140 does per thread) with one worker per GPU — puts block p on devices[p]
 * (NULL: device p modulo the device count) and keeps x on every device as nparts equal padded slices. Per SpMV the x slices
 * are exchanged (RCCL allgather over xGMI; peer / device copies where RCCL is unavailable or several parts share a device)
 * while each device computes the part of its block whose columns lie in its own slice; the remote-column part is accumulated
 * once the exchange has landed. y comes back in global row order. Square matrices only.
 * exchange: 0 = auto (RCCL when the devices are distinct and librccl.so.1 loads, else copies), 1 = RCCL, 2 = copies. */
typedef struct spmv_mi355x_partitioned spmv_mi355x_partitioned;   /* opaque */
int  spmv_mi355x_create_partitioned(spmv_mi355x_partitioned ** out, int nparts, const int * devices, int exchange, int format,
		int precision, long m, long n, long nnz, const int32_t * row_ptr, const int32_t * col_idx, const double * values_fp64,
		const spmv_mi355x_opts * opts /* may be NULL; device / row block / column filter fields are set per part */);
int  spmv_mi355x_destroy_partitioned(spmv_mi355x_partitioned * P);
/* Matrix_Format::spmv(x, y) with host buffers (n and m values of the handle's precision), same caching convention as
 * spmv_mi355x_spmv: x is uploaded when its pointer is new, y is downloaded on the first call after an upload. */
int  spmv_mi355x_spmv_partitioned(spmv_mi355x_partitioned * P, const void * x_host, void * y_host);
int  spmv_mi355x_partitioned_set_always_copy(spmv_mi355x_partitioned * P, int on);
/* `iters` SpMVs back to back on the resident x (exchange forced every time): wall-clock ms per SpMV between all-device syncs. */
int  spmv_mi355x_time_partitioned(spmv_mi355x_partitioned * P, int iters, double * ms_per_iter_out);
int  spmv_mi355x_partitioned_parts(const spmv_mi355x_partitioned * P);
int  spmv_mi355x_partitioned_offsets(const spmv_mi355x_partitioned * P, long * offsets_out /* [nparts+1] */);
const char * spmv_mi355x_partitioned_format_name(const spmv_mi355x_partitioned * P);
const char * spmv_mi355x_partitioned_exchange(const spmv_mi355x_partitioned * P);     /* "RCCL allgather" | "peer copies" | ... */
double spmv_mi355x_partitioned_mem_footprint(const spmv_mi355x_partitioned * P);

/* ---- format introspection for parity tests (host copies of the converted arrays) -------------------------- */
/* One stored array of the plain SELL layout ("val", "col", "slice_ptr", "row_of_sorted"), of the LDS-window SELL layout ("val", "idx",
 * "desc", "row_of_sorted", "groups") or of the column-blocked layout
 * ("entries", "val", "batch_base", "batch_ptr", "chunk_ptr", "chunk_row", "wg_rows", "range_row", "range_long", "long_row") exactly as it
 * lies in device memory: a malloc'ed copy (free with spmv_mi355x_free). For the tests that hold the host and the GPU builder of a
 * layout to the same bytes, and for diagnostics. */
int  spmv_mi355x_stored_array(const spmv_mi355x_matrix * A, const char * name, void ** out, size_t * bytes_out);

/* SELL-C-sigma layout: any out pointer may be NULL. Arrays are malloc'ed copies; free with spmv_mi355x_free().
 * For delta-compressed handles the column array is DECODED back to the plain column-major layout. */
int  spmv_mi355x_sell_layout(const spmv_mi355x_matrix * A, long * C_out, long * sigma_out, long * num_slices_out,
		long * nnz_ext_out, int64_t ** slice_ptr_out, int32_t ** col_out, double ** val_as_f64_out,
		int32_t ** row_of_sorted_out);
/* merge-path tile start coordinates (row, nnz) — num_tiles+1 pairs */
int  spmv_mi355x_merge_tiles(const spmv_mi355x_matrix * A, long * num_tiles_out, long * tile_items_out, int32_t ** coords_out);
void spmv_mi355x_free(void * p);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_MI355X_H */
