/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, gcc) of the reference algorithms on the SpMV hot path of
 * LiHaoxu/SpMV-Research, used ONLY as the checker by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py. Nothing under spmv-research_amd/ may include, link or call it.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference;
 * BENCH = benchmark_code/BENCH/src).
 *
 * PINNING STATUS (how each restatement is tied to the reference's actual behaviour):
 *   - orc_csr_*, orc_csr_kahan_*, orc_csr_vec_*, orc_sell_sorted_*, orc_mtx_read, orc_coo_to_csr,
 *     orc_partition_*: pinned BIT-EXACT against the genuine reference code compiled from its own sources
 *     (oracle/_ref, recipe oracle/Makefile) on the fixtures in tests/golden/ (generator:
 *     oracle/gen_golden.py) and live in tests/test_oracle_vs_ref.py when oracle/_ref is present.
 *   - orc_merge_*: the reference merge TU (BENCH/spmv_kernels/merge.cpp) is UNBUILDABLE here: it includes
 *     merge/sparse_matrix.h -> artificial_matrix_generation.h from the un-vendored, empty
 *     artificial-matrix-generator submodule. Pinned indirectly: with num_threads == 1 the algorithm
 *     degenerates to the sequential row loop and must bit-equal the reference csr result; with T > 1 only
 *     rows cut by a thread boundary may differ, within 1e-12 relative. The T>1 carry path is
 *     "parity unpinned" w.r.t. the reference binary.
 *   - orc_coo_*: the reference's COO arithmetic lives in Intel MKL (mkl_cspblas_dcoogemv, call site
 *     BENCH/spmv_kernels/mkl_coo.cpp:102-104; MKL version unpinned, library absent). Only the in-repo
 *     CSR->COO expansion (mkl_coo.cpp:79-90) is restated; the multiply is the published COO definition
 *     (y[r] += v*x[c] over entries in order). "parity unpinned" at the library boundary; results are
 *     checked against the reference csr result / quad gold instead.
 *   - orc_gold_*: restates check_accuracy()'s _Float128 Kahan gold (BENCH/bench_spmv.cpp:151-170); the
 *     reference never exports it, so it is cross-checked only through the error metrics being ~1 ulp.
 */
#ifndef SPMV_ORACLE_H
#define SPMV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- partitioners: lib/parallel_util.h:47-91 and :156-184 (binary search lib/macros/macrolib.h:537-590) */
void orc_partition_iterations(long num_workers, long worker_pos, long start, long end, long * s_out, long * e_out);
void orc_partition_prefix_sums(long num_workers, long worker_pos, const int32_t * sums, long N, long total_sum,
		long * s_out, long * e_out);

/* ---- Matrix Market + COO->CSR: lib/storage_formats/matrix_market/matrix_market.c:150-323,420-454,
 *      matrix_market_gen.c:65-202, lib/storage_formats/csr/csr_gen.c:99-213 */
typedef struct {
	long m, n, nnz, nnz_sym, nnz_diag, nnz_non_diag;
	int symmetric, skew, hermitian;
	char field[16];       /* field as written in the file (values are always converted to real) */
	int32_t * R;          /* [nnz] 0-based */
	int32_t * C;
	double * V;
} orc_coo_t;
int  orc_mtx_read(const char * filename, int num_threads, orc_coo_t * out, char * err, long err_n);
void orc_coo_free(orc_coo_t * coo);
void orc_coo_to_csr(const int32_t * R, const int32_t * C, const double * V, long m, long n, long nnz,
		int32_t * row_ptr, int32_t * col_idx, double * values);

/* ---- CSR scalar (BENCH/spmv_kernels/csr.cpp:334-350,381-404), Kahan (:353-373) */
void orc_csr_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y, int num_threads);
void orc_csr_spmv_f32(const int32_t * row_ptr, const int32_t * col_idx, const float * a, long m,
		const float * x, float * y, int num_threads);
void orc_csr_kahan_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y);

/* ---- CSR vector, SIMD within a row (BENCH/spmv_kernels/csr_vec.cpp:182-213) */
void orc_csr_vec_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y, int vec_len);
void orc_csr_vec_spmv_f32(const int32_t * row_ptr, const int32_t * col_idx, const float * a, long m,
		const float * x, float * y, int vec_len);

/* ---- merge-path CSR (BENCH/spmv_kernels/merge.cpp:226-319) */
void orc_merge_path_search(long diagonal, const int32_t * row_end_offsets, long a_len, long b_len, long * x_out, long * y_out);
void orc_merge_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m, long nnz,
		const double * x, double * y, int num_threads);
void orc_merge_spmv_f32(const int32_t * row_ptr, const int32_t * col_idx, const float * a, long m, long nnz,
		const float * x, float * y, int num_threads);

/* ---- SELL-C-sigma as the reference builds it (BENCH/spmv_kernels/sell_sorted.cpp:112-298, kernel :338-419):
 *      sigma = each thread's nnz-balanced row range, ascending stable degree sort inside it. */
typedef struct {
	long m, nnz, C, num_slices, nnz_ext;
	int32_t * slice_ptr;       /* [num_slices+1] element offsets (width*C per slice) */
	int32_t * ja;              /* [nnz_ext] column-major inside a slice */
	double * a;                /* [nnz_ext] */
	int32_t * permutation;     /* [m] original row -> sorted position */
	int32_t * rev_permutation; /* [m] sorted position -> original row */
	double mem_footprint;      /* with sizeof(ValueType) = value_bytes */
} orc_sell_t;
int  orc_sell_sorted_build(const int32_t * row_ptr, const int32_t * col_idx, const double * values, long m, long nnz,
		int C, int num_threads, int value_bytes, orc_sell_t * out);
void orc_sell_free(orc_sell_t * s);
/* the BSC SELL-C-sigma library's layout (sell_c_s.cpp:58-75, sell-C-s/RISC-V/sellcs_format.c:137-200, radix_sort.c:36-122):
 * pinned bit for bit to the reference's own format code compiled in place (oracle/_ref/.../libref_sellcs.so) */
long orc_sellcs_layout(const int32_t * row_ptr, const int32_t * col_idx, const double * values, long m, long C, long sigma,
		int32_t * row_order, int32_t * widths, int64_t * slice_ptr, int32_t * col, double * val);
void orc_sell_spmv_f64(const orc_sell_t * s, const double * x, double * y);
void orc_sell_spmv_f32(const orc_sell_t * s, const float * x, float * y);

/* ---- COO (expansion: BENCH/spmv_kernels/mkl_coo.cpp:79-90; multiply: published definition) */
void orc_csr_to_coo_rows(const int32_t * row_ptr, long m, int32_t * rowind);
void orc_coo_spmv_f64(const int32_t * rowind, const int32_t * colind, const double * val, long m, long nnz,
		const double * x, double * y);
void orc_coo_spmv_f32(const int32_t * rowind, const int32_t * colind, const float * val, long m, long nnz,
		const float * x, float * y);

/* ---- quad-precision Kahan gold + the 8 CSV error metrics (BENCH/bench_spmv.cpp:108-235,
 *      lib/array_metrics.c:1477-2149). y_test is given in double (the reference converts to _Float128 and
 *      back to double for the metrics). metrics_out[8] = mae,max_ae,mse,mape,smape,lnQ_error,mlare,gmare;
 *      returns the reference's maxDiff (max relative diff over rows with y_gold > eps). */
void   orc_gold_spmv(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y_gold_as_double);
double orc_check_accuracy(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, const double * y_test, int is_double, double * metrics_out);

/* ---- timing helper for bench.py's cpu_baseline leg (protocol of BENCH/bench_spmv.cpp:335-382) */
double orc_time_csr_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y, int num_threads, long min_loops, double min_runtime,
		long * loops_out, double * tmin_out, double * tmax_out);

/* ---- CSR with symmetric storage, one thread (BENCH/spmv_kernels/csr_sym.cpp:191-267); pinned against oracle/_ref */
void orc_csr_sym_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y);
void orc_csr_sym_spmv_f32(const int32_t * row_ptr, const int32_t * col_idx, const float * a, long m,
		const float * x, float * y);

/* ---- solver callers of spmv() — PARITY UNPINNED (bench_cg.cpp / bench_bicg.cpp do not compile here; see
 *      solver_oracle.c). history: 3 doubles per loop (error, error_explicit, error_best); info_out[4] =
 *      {eps, eps_counter, err_best, restarts}; return num_loops_out, -1 zero diagonal, -2 not square. */
long orc_pcg_f64(const int * row_ptr, const int * col, const double * val, long m, long n, const double * b,
		double * x_out, long max_iterations, double * history, double * info_out);
long orc_pbicgstab_f64(const int * row_ptr, const int * col, const double * val, long m, long n, const double * b,
		double * x_out, long max_iterations, double * history, double * info_out);

#ifdef __cplusplus
}
#endif
#endif
