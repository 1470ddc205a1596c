/* spmv_host.h — C ABI of the host side of the MI355X SpMV engine (libspmv_host.so, plain C++/OpenMP, no HIP).
 *
 * These are the "either side of the kernel" pieces of the reference hot path that a drop-in needs, with the
 * reference's exact semantics:
 *
 *   spmv_host_mtx_read        <- mtx_read + mtx_values_convert_to_real
 *                                (lib/storage_formats/matrix_market/matrix_market.c:150-323,420-454; matrix_market_gen.c:65-202)
 *   spmv_host_coo_to_csr      <- coo_to_csr(..., sort_columns=1, transpose=0)  (lib/storage_formats/csr/csr_gen.c:99-213)
 *   spmv_host_partition_*     <- loop_partitioner_balance_iterations / _prefix_sums (lib/parallel_util.h:47-91,156-184)
 *   spmv_host_gen_*           synthetic stand-ins for the SuiteSparse matrices of BASELINE.json (no .mtx file exists
 *                             in the reference tree and there is no network): "twins" driven by the parameter strings of
 *                             benchmark_code/BENCH/config.sh:402-455 (generator is OURS: the reference's generator is an
 *                             un-vendored submodule) and an analytic nlpkkt-like KKT matrix.
 *
 * No SpMV arithmetic lives here: y always comes from libspmv_mi355x.so.
 * Every function returns 0 on success; spmv_host_last_error() gives the message otherwise. Arrays handed out are
 * malloc'ed and released with spmv_host_free().
 */
#ifndef SPMV_HOST_H
#define SPMV_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char * spmv_host_last_error(void);
void spmv_host_free(void * p);

/* COO as the reference loader produces it: 0-based int32 indices, fp64 values (integer -> double, complex -> |z|,
 * pattern -> 1.0), symmetric files expanded (entries [0,nnz_sym) = file order, then the mirrored off-diagonals in
 * file order). */
typedef struct {
	long m, n, nnz, nnz_sym, nnz_diag, nnz_non_diag;
	int  symmetric, skew, hermitian, pad_;
	char field[16];
	int32_t * R;
	int32_t * C;
	double *  V;
} spmv_host_coo;

int  spmv_host_mtx_read(const char * filename, spmv_host_coo * out);
void spmv_host_coo_free(spmv_host_coo * coo);

/* rows ascending, columns ascending inside a row, duplicates kept (in input order). row_ptr[m+1], col_idx[nnz], values[nnz]
 * are caller-allocated. */
int  spmv_host_coo_to_csr(const int32_t * R, const int32_t * C, const double * V, long m, long n, long nnz,
		int32_t * row_ptr, int32_t * col_idx, double * values);

/* Matrix-Market writer (coordinate real general, 1-based, %.17g) for small synthetic inputs. */
int  spmv_host_mtx_write_csr(const char * filename, const int32_t * row_ptr, const int32_t * col_idx, const double * values,
		long m, long n);

/* worker w of W gets [*s, *e) */
int  spmv_host_partition_iterations(long num_workers, long worker_pos, long start, long end, long * s, long * e);
int  spmv_host_partition_prefix_sums(long num_workers, long worker_pos, const int32_t * sums, long N, long total_sum,
		long * s, long * e);

/* CSR produced by the generators */
typedef struct {
	long m, n, nnz;
	int32_t * row_ptr;
	int32_t * col_idx;
	double *  values;
} spmv_host_csr;
void spmv_host_csr_free(spmv_host_csr * csr);

/* Twin of a real matrix from the reference's feature vector (argument order of bench.cpp:569-579):
 * nr_rows nr_cols avg_nnz_per_row std_nnz_per_row distribution placement avg_bw_scaled skew avg_num_neighbours
 * cross_row_similarity seed. `pattern` != 0 -> all values 1.0 (pattern matrices), else Uniform(-1,1). */
int  spmv_host_gen_twin(long nr_rows, long nr_cols, double avg_nnz_per_row, double std_nnz_per_row,
		double avg_bw_scaled, double skew, double avg_num_neighbours, double cross_row_similarity,
		unsigned long seed, int pattern, spmv_host_csr * out);
/* Named configs of BASELINE.json: "cant", "scircuit", "pwtk", "soc-LiveJournal1", "nlpkkt240"; `scale` in (0,1] shrinks
 * the row count (nlpkkt: the grid edge) for tests. */
int  spmv_host_gen_named(const char * name, double scale, spmv_host_csr * out);
/* Symmetric indefinite KKT-like matrix [H A^T; A 0] over an N^3 grid (m = 2N^3 + 6N^2, ~27.5 nnz/row; N = 240 gives
 * the size of nlpkkt240). */
int  spmv_host_gen_kkt(long N, unsigned long seed, spmv_host_csr * out);

/* Row-partitioned generation (each rank of a multi-GPU run builds only its block): the global row_ptr alone
 * (row_ptr may be NULL to query m), and rows [row_begin,row_end) as a local CSR with global column indices. */
int  spmv_host_gen_kkt_row_ptr(long N, int32_t * row_ptr /* [m+1] or NULL */, long * m_out, long * nnz_out);
int  spmv_host_gen_kkt_block(long N, unsigned long seed, long row_begin, long row_end, spmv_host_csr * out);
/* An arbitrary ascending list of rows of the same matrix as a local CSR (a rank's rows under a graph partition). */
int  spmv_host_gen_kkt_rows(long N, unsigned long seed, const int32_t * rows, long count, spmv_host_csr * out);
/* The same into CALLER-allocated arrays — rows[count] (or the contiguous rows row_begin.. when rows is NULL), row_ptr[count+1],
 * col_idx[capacity], values[capacity] or NULL for the structure alone: no second copy of a rank's block is ever made. */
int  spmv_host_gen_kkt_rows_into(long N, unsigned long seed, const int32_t * rows, long row_begin, long count, int32_t * row_ptr,
		int32_t * col_idx, double * values, long capacity);
/* The same rows with a COLUMN FILTER applied while they are generated: keep_inside = 1 keeps the columns in [col_lo, col_hi), 0 the
 * others — the local-column / remote-column halves of a rank's row block (the overlap scheme of SURVEY §8e) without ever holding
 * the unfiltered block on the host. col_idx == NULL: only row_ptr (the filtered prefix) is computed, to size the arrays. */
int  spmv_host_gen_kkt_rows_filtered(long N, unsigned long seed, const int32_t * rows, long row_begin, long count, long col_lo, long col_hi,
		int keep_inside, int32_t * row_ptr, int32_t * col_idx, double * values, long capacity);
/* In place: in a fraction `frac` of the rows every off-diagonal column moves by a random offset in [-span, span] (rows stay
 * sorted and duplicate-free). Breaks the translation invariance of a generated matrix: bench.py --jitter measures how much
 * of the compressed-index SELL format's advantage rests on it. */
int  spmv_host_jitter_columns(long m, long n, const int32_t * row_ptr, int32_t * col_idx, double * values, double frac, long span,
		unsigned long seed);
/* The graph partition of that matrix WITHOUT building it (breadth-first sweep and exchange volumes over columns computed on the
 * fly): owner[m] in [0,parts) as spmv_host_bfs_order + spmv_host_owners_from_order give; volume[p] as spmv_host_partition_volume. */
int  spmv_host_kkt_bfs_owner(long N, long parts, int32_t * owner);
int  spmv_host_kkt_partition_volume(long N, const int32_t * owner, long parts, long * volume);
/* x kept as `parts` slices padded to `padded` entries (one equal-sized allgather): column c of part p becomes
 * p*padded + (c - offsets[p]); offsets has parts+1 entries. In place. */
int  spmv_host_remap_columns(int32_t * col_idx, long nnz, const long * offsets, long parts, long padded);

/* For columns already in the padded layout: the sub-range [lo[q],hi[q]) of each part's slice that is referenced at all
 * (hi[q] <= lo[q]: none). Used to trim the x exchange to what a row block really reads. */
int  spmv_host_column_ranges(const int32_t * col_idx, long nnz, long padded, long parts, long * lo, long * hi);

/* Communication-aware row partition for the multi-GPU SpMV (square matrices; host/graph_partition.cpp). The reference has
 * no counterpart: its threads share x (csr.cpp:140 partitions rows only for load balance, lib/parallel_util.h:156-184).
 *   bfs_order          order[m]: vertices in breadth-first order from a pseudo-peripheral vertex, every component
 *   owners_from_order  owner[v] in [0,parts): nnz-balanced contiguous cuts of that order
 *   partition_volume   volume[p]: distinct x entries part p reads from other parts under `owner`
 *   partition_layout   perm[m] (new -> old) and offsets[parts+1]: parts in order; inside a part first the vertices other
 *                      parts read (grouped by the lowest reader), then the interior, each group in original order
 *   permuted_block     rows [row_begin,row_end) of P A P^T (inv = old -> new) as a local CSR with columns ascending in
 *                      the new numbering; free with spmv_host_csr_free */
int  spmv_host_bfs_order(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, int32_t * order);
int  spmv_host_owners_from_order(const int32_t * row_ptr, long m, const int32_t * order, long parts, int32_t * owner);
int  spmv_host_partition_volume(const int32_t * row_ptr, const int32_t * col_idx, long m, const int32_t * owner, long parts,
		long * volume);
int  spmv_host_partition_layout(const int32_t * row_ptr, const int32_t * col_idx, long m, const int32_t * owner, long parts,
		int32_t * perm, long * offsets);
int  spmv_host_permuted_block(const int32_t * row_ptr, const int32_t * col_idx, const double * values, long m,
		const int32_t * perm, const int32_t * inv, long row_begin, long row_end, spmv_host_csr * out);

/* Halo index lists of `rank` under an owner map, in ORIGINAL vertex numbers, every list ascending: send (vertices of `rank`
 * that rows of part q read) and recv (vertices of part q that rows of `rank` read), concatenated over q with
 * offsets[parts+1]. send[q] on rank p and recv[p] on rank q are the same set, so packed buffers need no header. The two
 * lists are malloc'ed; free with spmv_host_free. */
int  spmv_host_halo_lists(const int32_t * row_ptr, const int32_t * col_idx, long m, const int32_t * owner, long parts, long rank,
		long * send_offsets, int32_t ** send_list, long * recv_offsets, int32_t ** recv_list);

/* Structural features the reference uses to describe a matrix (lib/storage_formats/csr_util/csr_util_gen.c:437-447,
 * 596-695,961): out[0..6] = avg nnz/row, std nnz/row, avg bandwidth scaled by n, skew = (max-avg)/avg,
 * avg_num_neighbours (window 1), cross_row_similarity (window 1), max nnz/row. */
int  spmv_host_csr_features(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, double * out7);
/* The matrix statistics of the reference's artificial-matrix CSV row (bench_spmv.cpp:489-563): out15 = density [%], mem_footprint
 * [MiB, fp64 CSR], avg/std nnz per row, avg/std bandwidth, the same scaled by n, avg/std scatter (degree / bandwidth), the same for
 * the scaled bandwidth, skew, avg_num_neighbours, cross_row_similarity; mem_range = the power-of-two class "[lo-hi]" in MiB. */
int  spmv_host_csr_am_stats(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, double * out15, char * mem_range,
		long mem_range_n);

#ifdef __cplusplus
}
#endif
#endif
