// TEST INFRASTRUCTURE — not part of the product.
//
// Thin extern "C" shim that is linked TOGETHER WITH the genuine reference translation units
// (compiled in place from /root/reference by oracle/Makefile, outputs only under oracle/_ref/).
// It lets tests / fixture generation drive the reference's own code path
//     mtx_read -> mtx_values_convert_to_real -> coo_to_csr(...,1,0) -> csr_to_format -> MF->spmv
// exactly the way the reference driver does (benchmark_code/BENCH/src/bench.cpp:180-224,600-603 and
// bench_spmv.cpp:598-609), and hands the arrays back through plain pointers.
//
// Nothing in this file restates reference code; it only calls it. One shared object is built per
// (backend TU, precision) pair because every backend defines the same csr_to_format symbol.

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <omp.h>

#include "macros/cpp_defines.h"
extern "C" {
#include "storage_formats/matrix_market/matrix_market.h"
#include "aux/csr_converter_reference.h"
}
#include "spmv_kernels/spmv_kernel.h"

static struct Matrix_Format * g_MF = NULL;

extern "C" {

int ref_sizeof_value(void) { return (int) sizeof(ValueType); }
int ref_max_threads(void) { return omp_get_max_threads(); }
void ref_set_threads(int t) { omp_set_num_threads(t); }

// Reads a Matrix-Market file with the reference loader and converts to CSR with the reference
// converter. Arrays are malloc'ed here; caller frees with ref_free().
int ref_mtx_to_csr(const char * filename, long * m, long * n, long * nnz, long * symmetric,
		long * nnz_diag, long * nnz_non_diag,
		int32_t ** row_ptr, int32_t ** col_idx, double ** values)
{
	struct Matrix_Market * MTX = mtx_read((char *) filename, 1, 1);
	long M = MTX->m, N = MTX->n, NNZ = MTX->nnz;
	*symmetric = MTX->symmetric;
	*nnz_diag = MTX->nnz_diag;
	*nnz_non_diag = MTX->nnz_non_diag;
	mtx_values_convert_to_real(MTX);
	int32_t * ia = (int32_t *) calloc(M + 1, sizeof(*ia));
	int32_t * ja = (int32_t *) calloc(NNZ > 0 ? NNZ : 1, sizeof(*ja));
	double * a = (double *) calloc(NNZ > 0 ? NNZ : 1, sizeof(*a));
	coo_to_csr(MTX->R, MTX->C, (double *) MTX->V, M, N, NNZ, ia, ja, a, 1, 0);
	mtx_destroy(&MTX);
	*m = M; *n = N; *nnz = NNZ;
	*row_ptr = ia; *col_idx = ja; *values = a;
	return 0;
}

// KEEP_SYMMETRY flavour of the same (bench.cpp:131-136,180-192): the file's own entries only (nnz_sym of them, one
// triangle for symmetric files), converted to CSR by the reference converter.
int ref_mtx_to_csr_keep_symmetry(const char * filename, long * m, long * n, long * nnz, long * symmetric,
		int32_t ** row_ptr, int32_t ** col_idx, double ** values)
{
	struct Matrix_Market * MTX = mtx_read((char *) filename, 0, 1);
	long M = MTX->m, N = MTX->n, NNZ = MTX->nnz_sym;
	*symmetric = MTX->symmetric;
	mtx_values_convert_to_real(MTX);
	int32_t * ia = (int32_t *) calloc(M + 1, sizeof(*ia));
	int32_t * ja = (int32_t *) calloc(NNZ > 0 ? NNZ : 1, sizeof(*ja));
	double * a = (double *) calloc(NNZ > 0 ? NNZ : 1, sizeof(*a));
	coo_to_csr(MTX->R, MTX->C, (double *) MTX->V, M, N, NNZ, ia, ja, a, 1, 0);
	mtx_destroy(&MTX);
	*m = M; *n = N; *nnz = NNZ;
	*row_ptr = ia; *col_idx = ja; *values = a;
	return 0;
}

void ref_free(void * p) { free(p); }

// COO (0-based) -> CSR through the reference converter (sorted columns, no transpose).
int ref_coo_to_csr(int32_t * R, int32_t * C, double * V, long m, long n, long nnz,
		int32_t * row_ptr, int32_t * col_idx, double * values)
{
	coo_to_csr(R, C, V, m, n, nnz, row_ptr, col_idx, values, 1, 0);
	return 0;
}

// csr_to_format of the linked backend TU. One live format per process (the reference keeps
// per-thread state in file statics).
int ref_csr_to_format(int32_t * row_ptr, int32_t * col_idx, double * values, long m, long n, long nnz)
{
	setenv("USE_PROCESSES", "0", 0);   // read without NULL check by the backends' constructors
	g_MF = csr_to_format(row_ptr, col_idx, values, m, n, nnz, 0, 1);
	return g_MF == NULL;
}

// the symmetric-storage backend (csr_sym.cpp) takes symmetric = 1, symmetry_expanded = 0 (csr_sym.cpp:118-123)
int ref_csr_to_format_symmetric(int32_t * row_ptr, int32_t * col_idx, double * values, long m, long n, long nnz)
{
	setenv("USE_PROCESSES", "0", 0);
	g_MF = csr_to_format(row_ptr, col_idx, values, m, n, nnz, 1, 0);
	return g_MF == NULL;
}

const char * ref_format_name(void) { return g_MF ? g_MF->format_name : ""; }
double ref_mem_footprint(void) { return g_MF ? g_MF->mem_footprint : 0; }
double ref_csr_mem_footprint(void) { return g_MF ? g_MF->csr_mem_footprint : 0; }

// x has n elements, y has m + 64 elements (driver convention, bench_spmv.cpp:606-609).
int ref_spmv(void * x, void * y)
{
	if (!g_MF)
		return 1;
	g_MF->spmv((ValueType *) x, (ValueType *) y);
	return 0;
}

// Timed loop in the reference driver's convention (bench_spmv.cpp:335-382): one warm-up call, then
// per-call CLOCK_MONOTONIC_RAW timing until >= min_loops and >= min_runtime; returns the median.
double ref_time_spmv(void * x, void * y, long min_loops, double min_runtime, long * loops_out,
		double * tmin_out, double * tmax_out)
{
	if (!g_MF)
		return -1;
	long cap = 1 << 20, n = 0;
	double * t = (double *) malloc(cap * sizeof(*t));
	double total = 0;
	g_MF->spmv((ValueType *) x, (ValueType *) y);
	while ((total < min_runtime || n < min_loops) && n < cap)
	{
		struct timespec a, b;
		clock_gettime(CLOCK_MONOTONIC_RAW, &a);
		g_MF->spmv((ValueType *) x, (ValueType *) y);
		clock_gettime(CLOCK_MONOTONIC_RAW, &b);
		t[n] = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
		total += t[n];
		n++;
	}
	// insertion-free median: simple qsort
	qsort(t, n, sizeof(*t), [](const void * p, const void * q) -> int {
		double u = *(const double *) p, v = *(const double *) q;
		return (u > v) - (u < v);
	});
	double med = t[n / 2];
	if (loops_out) *loops_out = n;
	if (tmin_out) *tmin_out = t[0];
	if (tmax_out) *tmax_out = t[n - 1];
	free(t);
	return med;
}

}
