/* TEST INFRASTRUCTURE — NOT PRODUCT CODE. See oracle.h for scope and pinning status.
 *
 * CPU restatement of the reference SpMV kernels and format builders. Written from the algorithm
 * descriptions in SURVEY.md §8(a) and a reading of the cited reference lines; arithmetic order is
 * what matters (results are compared bit-for-bit with the genuine reference build in oracle/_ref).
 *
 * Floating point: compiled with -ffp-contract=off; every place where the reference build
 * (gcc -O3 -march=native, default -ffp-contract=fast) fuses a multiply-add is written as an
 * explicit fma()/fmaf() so the result does not depend on this file's compiler flags.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <time.h>
#include <omp.h>
#include <quadmath.h>

#include "oracle.h"

/* ======================================================================================== partitioners */

/* lib/parallel_util.h:47-91 (increment +1 only, the only form used on the path). */
void
orc_partition_iterations(long num_workers, long worker_pos, long start, long end, long * s_out, long * e_out)
{
	long len = end - start;
	if (len < 1)
	{
		/* |incr| > len: worker 0 owns the (empty) range, the others get [end,end). */
		*s_out = (worker_pos == 0) ? start : end;
		*e_out = end;
		return;
	}
	long per = len / num_workers;
	long rem = len % num_workers;
	if (rem != 0 && worker_pos < rem)
	{
		per += 1;
		rem = 0;
	}
	long ls = start + per * worker_pos + rem;
	long le = ls + per;
	if (worker_pos == num_workers - 1)
		le = end;
	*s_out = ls;
	*e_out = le;
}

/* lib/macros/macrolib.h:537-590 — "closest value" binary search over A[lo..hi] (inclusive bounds). */
static long
closest_index(const int32_t * A, long lo, long hi, long target)
{
	long s = lo, e = hi, mid;
	if (target < A[s])
		return s;
	if (target > A[e])
		return e;
	while (1)
	{
		mid = (s + e) / 2;
		if (mid == s || mid == e)
			break;
		if (target > A[mid])
			s = mid;
		else
			e = mid;
	}
	if (target == A[s])
		return s;
	if (target == A[e])
		return e;
	long ds = labs(target - (long) A[s]);
	long de = labs(target - (long) A[e]);
	return (ds < de) ? s : e;
}

/* lib/parallel_util.h:156-184. 'sums[N]' is never read; targets are computed in long (the type of
 * total_sum at every call site). */
void
orc_partition_prefix_sums(long num_workers, long worker_pos, const int32_t * sums, long N, long total_sum,
		long * s_out, long * e_out)
{
	long target = sums[0] + (total_sum * worker_pos) / num_workers;
	long target_next = sums[0] + (total_sum * (worker_pos + 1)) / num_workers;
	long i_s, i_e;
	i_s = (worker_pos == 0) ? 0 : closest_index(sums, 0, N - 1, target);
	i_e = (worker_pos == num_workers - 1) ? N : closest_index(sums, 0, N - 1, target_next);
	*s_out = i_s;
	*e_out = i_e;
}

/* ========================================================================================== CSR scalar */

/* BENCH/spmv_kernels/csr.cpp:334-350: per row, sum = 0; sum += a[j]*x[ja[j]] left to right; the reference
 * build contracts the statement to one scalar FMA per non-zero (SURVEY.md §8 a3 [probe]). */
#define CSR_SCALAR_BODY(T, FMA)                                                     \
	long i, j, j_e;                                                             \
	j = row_ptr[i_s];                                                           \
	for (i = i_s; i < i_e; i++)                                                 \
	{                                                                           \
		T sum = 0;                                                          \
		j_e = row_ptr[i + 1];                                               \
		for (; j < j_e; j++)                                                \
			sum = FMA(a[j], x[col_idx[j]], sum);                        \
		y[i] = sum;                                                         \
	}

static void
csr_rows_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, const double * x, double * y, long i_s, long i_e)
{
	CSR_SCALAR_BODY(double, fma)
}

static void
csr_rows_f32(const int32_t * row_ptr, const int32_t * col_idx, const float * a, const float * x, float * y, long i_s, long i_e)
{
	CSR_SCALAR_BODY(float, fmaf)
}

/* BENCH/spmv_kernels/csr.cpp:381-404 with the thread ranges of csr.cpp:140 (nnz-balanced). */
void
orc_csr_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y, int num_threads)
{
	if (num_threads < 1)
		num_threads = 1;
	if (m < 1)
		return;
	long nnz = row_ptr[m] - row_ptr[0];
	#pragma omp parallel num_threads(num_threads)
	{
		long i_s, i_e;
		orc_partition_prefix_sums(omp_get_num_threads(), omp_get_thread_num(), row_ptr, m, nnz, &i_s, &i_e);
		csr_rows_f64(row_ptr, col_idx, a, x, y, i_s, i_e);
	}
}

void
orc_csr_spmv_f32(const int32_t * row_ptr, const int32_t * col_idx, const float * a, long m,
		const float * x, float * y, int num_threads)
{
	if (num_threads < 1)
		num_threads = 1;
	if (m < 1)
		return;
	long nnz = row_ptr[m] - row_ptr[0];
	#pragma omp parallel num_threads(num_threads)
	{
		long i_s, i_e;
		orc_partition_prefix_sums(omp_get_num_threads(), omp_get_thread_num(), row_ptr, m, nnz, &i_s, &i_e);
		csr_rows_f32(row_ptr, col_idx, a, x, y, i_s, i_e);
	}
}

/* BENCH/spmv_kernels/csr.cpp:353-373. 'a*x - compensation' is contracted by the reference build to
 * fma(a, x, -compensation). */
void
orc_csr_kahan_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y)
{
	long i, j;
	for (i = 0; i < m; i++)
	{
		double sum = 0, compensation = 0, val, tmp;
		for (j = row_ptr[i]; j < row_ptr[i + 1]; j++)
		{
			val = fma(a[j], x[col_idx[j]], -compensation);
			tmp = sum + val;
			compensation = (tmp - sum) - val;
			sum = tmp;
		}
		y[i] = sum;
	}
}

/* ================================================================================ CSR, symmetric storage */

/* BENCH/spmv_kernels/csr_sym.cpp:191-267 (subkernel_csr_sym_split + compute_csr) with ONE thread, so i_s = 0, i_e = m
 * and every column is "inside the thread's range": per stored entry (i, col, a):  sum_upper += a*x[col]  and, off the
 * diagonal,  y_upper[col] += a*x[i];  after the row  y_upper[i] = sum_upper  (an ASSIGNMENT: contributions that earlier
 * rows scattered into y_upper[i] would be lost — they do not exist when the stored triangle is the LOWER one, which is
 * what Matrix-Market symmetric files hold); finally y[i] = 0 + y_upper[i]. The reference build (gcc -O3, default
 * -ffp-contract=fast) contracts the row sum to an FMA but not the scatter (`prod` is rounded first, csr_sym.cpp:216):
 * established by trying the four combinations against oracle/_ref/libref_csr_sym_* and pinned bit for bit to it.
 * With T > 1 the reference scatters through compare-and-swap loops in a run-dependent order: not restated. */
#define CSR_SYM_BODY(T, FMA)                                                                  \
	long i, j;                                                                            \
	for (i = 0; i < m; i++)                                                               \
		y[i] = 0;                                                                     \
	for (i = 0; i < m; i++)                                                               \
	{                                                                                     \
		T sum_upper = 0;                                                              \
		for (j = row_ptr[i]; j < row_ptr[i + 1]; j++)                                 \
		{                                                                             \
			long col = col_idx[j];                                                \
			sum_upper = FMA(a[j], x[col], sum_upper);                             \
			if (i != col)                                                         \
			{                                                                     \
				T prod = a[j] * x[i];                                         \
				y[col] += prod;                                               \
			}                                                                     \
		}                                                                             \
		y[i] = sum_upper;                                                             \
	}

void
orc_csr_sym_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y)
{
	CSR_SYM_BODY(double, fma)
}

void
orc_csr_sym_spmv_f32(const int32_t * row_ptr, const int32_t * col_idx, const float * a, long m,
		const float * x, float * y)
{
	CSR_SYM_BODY(float, fmaf)
}

/* ========================================================================================== CSR vector */

/* BENCH/spmv_kernels/csr_vec.cpp:182-213: VEC_LEN lanes each FMA-accumulate every VEC_LEN-th element of the
 * vectorisable prefix of the row, horizontal add (halving tree, as _mm512_reduce_add_pd /
 * lib/vectorization/x86/avx256/...f64.h:169-177 do it), then the scalar remainder is FMA-added in order.
 * Empty rows get 0 (csr_vec.cpp:222-228). */
#define CSR_VEC_BODY(T, FMA)                                                                  \
	long i, j, k, w;                                                                      \
	T lanes[64];                                                                          \
	for (i = 0; i < m; i++)                                                               \
	{                                                                                     \
		long j_s = row_ptr[i], j_e = row_ptr[i + 1];                                  \
		long j_e_vec = j_s + ((j_e - j_s) & ~((long) vec_len - 1));                   \
		T sum;                                                                        \
		if (j_s == j_e) { y[i] = 0; continue; }                                       \
		for (k = 0; k < vec_len; k++)                                                 \
			lanes[k] = 0;                                                         \
		for (j = j_s; j < j_e_vec; j += vec_len)                                      \
			for (k = 0; k < vec_len; k++)                                         \
				lanes[k] = FMA(a[j + k], x[col_idx[j + k]], lanes[k]);        \
		for (w = vec_len / 2; w >= 1; w /= 2)                                         \
			for (k = 0; k < w; k++)                                               \
				lanes[k] = lanes[k] + lanes[k + w];                           \
		sum = lanes[0];                                                               \
		for (j = j_e_vec; j < j_e; j++)                                               \
			sum = FMA(a[j], x[col_idx[j]], sum);                                  \
		y[i] = sum;                                                                   \
	}

void
orc_csr_vec_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y, int vec_len)
{
	CSR_VEC_BODY(double, fma)
}

void
orc_csr_vec_spmv_f32(const int32_t * row_ptr, const int32_t * col_idx, const float * a, long m,
		const float * x, float * y, int vec_len)
{
	CSR_VEC_BODY(float, fmaf)
}

/* ========================================================================================== merge path */

/* BENCH/spmv_kernels/merge.cpp:226-252. List A = row end offsets (length a_len = m), list B = the
 * natural numbers 0..nnz-1 (counting iterator; kept integral here — the reference carries it in
 * ValueType, which loses exactness above 2^24 in fp32 builds). */
void
orc_merge_path_search(long diagonal, const int32_t * row_end_offsets, long a_len, long b_len, long * x_out, long * y_out)
{
	long x_min = diagonal - b_len;
	if (x_min < 0)
		x_min = 0;
	long x_max = (diagonal < a_len) ? diagonal : a_len;
	while (x_min < x_max)
	{
		long pivot = (x_min + x_max) >> 1;
		if (row_end_offsets[pivot] <= diagonal - pivot - 1)
			x_min = pivot + 1;
		else
			x_max = pivot;
	}
	*x_out = (x_min < a_len) ? x_min : a_len;
	*y_out = diagonal - x_min;
}

/* BENCH/spmv_kernels/merge.cpp:256-319: equal shares of the (m + nnz)-long merge path per thread, whole rows,
 * then the partial last row into a carry, serial fix-up in thread order. */
#define MERGE_BODY(T, FMA)                                                                              \
	const int32_t * row_end = row_ptr + 1;                                                          \
	long num_merge_items = m + nnz;                                                                 \
	long items_per_thread = (num_merge_items + num_threads - 1) / num_threads;                      \
	long * row_carry = (long *) malloc(num_threads * sizeof(*row_carry));                           \
	T * val_carry = (T *) malloc(num_threads * sizeof(*val_carry));                                 \
	for (long tid = 0; tid < num_threads; tid++)                                                    \
	{                                                                                               \
		long sx, sy, ex, ey;                                                                    \
		long d0 = items_per_thread * tid;                                                       \
		if (d0 > num_merge_items) d0 = num_merge_items;                                         \
		long d1 = d0 + items_per_thread;                                                        \
		if (d1 > num_merge_items) d1 = num_merge_items;                                         \
		orc_merge_path_search(d0, row_end, m, nnz, &sx, &sy);                                   \
		orc_merge_path_search(d1, row_end, m, nnz, &ex, &ey);                                   \
		for (; sx < ex; sx++)                                                                   \
		{                                                                                       \
			T running = 0;                                                                  \
			for (; sy < row_end[sx]; sy++)                                                  \
				running = FMA(a[sy], x[col_idx[sy]], running);                          \
			y[sx] = running;                                                                \
		}                                                                                       \
		T running = 0;                                                                          \
		for (; sy < ey; sy++)                                                                   \
			running = FMA(a[sy], x[col_idx[sy]], running);                                  \
		row_carry[tid] = ex;                                                                    \
		val_carry[tid] = running;                                                               \
	}                                                                                               \
	for (long tid = 0; tid < num_threads - 1; tid++)                                                \
		if (row_carry[tid] < m)                                                                 \
			y[row_carry[tid]] += val_carry[tid];                                            \
	free(row_carry);                                                                                \
	free(val_carry);

void
orc_merge_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m, long nnz,
		const double * x, double * y, int num_threads)
{
	if (num_threads < 1) num_threads = 1;
	MERGE_BODY(double, fma)
}

void
orc_merge_spmv_f32(const int32_t * row_ptr, const int32_t * col_idx, const float * a, long m, long nnz,
		const float * x, float * y, int num_threads)
{
	if (num_threads < 1) num_threads = 1;
	MERGE_BODY(float, fmaf)
}

/* ======================================================================================== SELL-C-sigma */

/* BENCH/spmv_kernels/sell_sorted.cpp:112-298. C = VEC_LEN rows per slice. The sort window "sigma" is each
 * thread's row range [i_s,i_e): slices are nnz-balanced over threads (:165), rows inside a range are
 * stably bucket-sorted by ascending degree (:184-189, lib/sort/bucketsort/bucketsort_gen.c:164-193),
 * the CSR is rebuilt in sorted order (:191-215), slice width = longest row of the slice (:234-249),
 * padding = value 0 with the column of the last real entry written so far by that thread (:269-287;
 * 'col' starts at 0 per thread), each slice stored column-major (:288-289). */
int
orc_sell_sorted_build(const int32_t * row_ptr, const int32_t * col_idx, const double * values, long m, long nnz,
		int C, int num_threads, int value_bytes, orc_sell_t * out)
{
	long num_slices = (m + C - 1) / C;
	long t, i, j, k, ii;
	int32_t * slice_ptr = (int32_t *) calloc(num_slices + 1, sizeof(*slice_ptr));
	int32_t * perm = (int32_t *) calloc(m > 0 ? m : 1, sizeof(*perm));
	int32_t * rev = (int32_t *) calloc(m > 0 ? m : 1, sizeof(*rev));
	int32_t * rp_re = (int32_t *) calloc(m + 1, sizeof(*rp_re));
	int32_t * ci_re = (int32_t *) calloc(nnz > 0 ? nnz : 1, sizeof(*ci_re));
	double * v_re = (double *) calloc(nnz > 0 ? nnz : 1, sizeof(*v_re));
	long * T_ii_s = (long *) calloc(num_threads, sizeof(long));
	long * T_ii_e = (long *) calloc(num_threads, sizeof(long));
	long * T_i_s = (long *) calloc(num_threads, sizeof(long));
	long * T_i_e = (long *) calloc(num_threads, sizeof(long));

	/* slice_ptr first holds the CSR offset of each slice's first row: the prefix sums the partitioner sees. */
	for (i = 0; i < num_slices; i++)
		slice_ptr[i] = row_ptr[i * C];
	slice_ptr[num_slices] = row_ptr[m];

	for (t = 0; t < num_threads; t++)
	{
		long ii_s, ii_e, i_s, i_e;
		orc_partition_prefix_sums(num_threads, t, slice_ptr, num_slices, nnz, &ii_s, &ii_e);
		i_s = ii_s * C;
		i_e = ii_e * C;
		if (t == num_threads - 1)
			i_e = m;
		T_ii_s[t] = ii_s; T_ii_e[t] = ii_e; T_i_s[t] = i_s; T_i_e[t] = i_e;

		/* stable counting sort of rows [i_s,i_e) by ascending degree */
		long degree_max = 0;
		for (i = i_s; i < i_e; i++)
			if (row_ptr[i + 1] - row_ptr[i] > degree_max)
				degree_max = row_ptr[i + 1] - row_ptr[i];
		long * offsets = (long *) calloc(degree_max + 2, sizeof(long));
		for (i = i_s; i < i_e; i++)
			offsets[row_ptr[i + 1] - row_ptr[i]]++;
		long acc = 0;
		for (k = 0; k < degree_max + 2; k++)   /* inclusive scan: offsets[b] = end of bucket b */
		{
			acc += offsets[k];
			offsets[k] = acc;
		}
		for (i = i_e - 1; i >= i_s; i--)
		{
			long b = row_ptr[i + 1] - row_ptr[i];
			offsets[b]--;
			perm[i] = (int32_t) (offsets[b] + i_s);
		}
		free(offsets);
		for (i = i_s; i < i_e; i++)
			rev[perm[i]] = (int32_t) i;
		for (i = i_s; i < i_e; i++)
			rp_re[perm[i]] = row_ptr[i + 1] - row_ptr[i];
	}
	/* exclusive scan of the sorted degrees -> reordered row_ptr (:207-208) */
	{
		long acc = 0;
		for (i = 0; i < m; i++)
		{
			long d = rp_re[i];
			rp_re[i] = (int32_t) acc;
			acc += d;
		}
		rp_re[m] = (int32_t) acc;
	}
	for (t = 0; t < num_threads; t++)
	{
		k = (T_i_s[t] < m) ? row_ptr[T_i_s[t]] : row_ptr[m];
		for (i = T_i_s[t]; i < T_i_e[t]; i++)
			for (j = row_ptr[rev[i]]; j < row_ptr[rev[i] + 1]; j++, k++)
			{
				ci_re[k] = col_idx[j];
				v_re[k] = values[j];
			}
	}

	/* slice widths, exclusive scan (:234-256) */
	for (i = 0; i < m; i += C)
	{
		long width = 0;
		long k_e = (i + C > m) ? m : i + C;
		for (k = i; k < k_e; k++)
			if (rp_re[k + 1] - rp_re[k] > width)
				width = rp_re[k + 1] - rp_re[k];
		slice_ptr[i / C] = (int32_t) (C * width);
	}
	{
		long acc = 0;
		for (i = 0; i < num_slices; i++)
		{
			long w = slice_ptr[i];
			slice_ptr[i] = (int32_t) acc;
			acc += w;
		}
		slice_ptr[num_slices] = (int32_t) acc;
	}
	long nnz_ext = slice_ptr[num_slices];
	double * a = (double *) calloc(nnz_ext > 0 ? nnz_ext : 1, sizeof(*a));
	int32_t * ja = (int32_t *) calloc(nnz_ext > 0 ? nnz_ext : 1, sizeof(*ja));
	double * rowmajor_a = (double *) malloc((size_t) 1 * sizeof(double));
	int32_t * rowmajor_ja = (int32_t *) malloc((size_t) 1 * sizeof(int32_t));
	long rowmajor_cap = 1;

	for (t = 0; t < num_threads; t++)
	{
		long col = 0;
		for (ii = T_ii_s[t]; ii < T_ii_e[t]; ii++)
		{
			long width = (slice_ptr[ii + 1] - slice_ptr[ii]) / C;
			long i_c_s = C * ii;
			long i_c_e = (i_c_s + C > m) ? m : i_c_s + C;
			long base = slice_ptr[ii];
			long sz = C * width;
			if (sz > rowmajor_cap)
			{
				rowmajor_cap = sz;
				rowmajor_a = (double *) realloc(rowmajor_a, sz * sizeof(double));
				rowmajor_ja = (int32_t *) realloc(rowmajor_ja, sz * sizeof(int32_t));
			}
			long jj = 0;
			for (i = i_c_s; i < i_c_e; i++)
			{
				for (j = rp_re[i]; j < rp_re[i + 1]; j++, jj++)
				{
					rowmajor_a[jj] = v_re[j];
					col = ci_re[j];
					rowmajor_ja[jj] = (int32_t) col;
				}
				for (; j < rp_re[i] + width; j++, jj++)
				{
					rowmajor_a[jj] = 0;
					rowmajor_ja[jj] = (int32_t) col;
				}
			}
			for (; jj < sz; jj++)
			{
				rowmajor_a[jj] = 0;
				rowmajor_ja[jj] = (int32_t) col;
			}
			/* transpose C x width (row-major) -> width x C (:288-289) */
			for (long c = 0; c < width; c++)
				for (long r = 0; r < C; r++)
				{
					a[base + c * C + r] = rowmajor_a[r * width + c];
					ja[base + c * C + r] = rowmajor_ja[r * width + c];
				}
		}
	}
	free(rowmajor_a); free(rowmajor_ja);
	free(rp_re); free(ci_re); free(v_re);
	free(T_ii_s); free(T_ii_e); free(T_i_s); free(T_i_e);

	out->m = m; out->nnz = nnz; out->C = C; out->num_slices = num_slices; out->nnz_ext = nnz_ext;
	out->slice_ptr = slice_ptr; out->ja = ja; out->a = a;
	out->permutation = perm; out->rev_permutation = rev;
	/* sell_sorted.cpp:297 */
	out->mem_footprint = (double) (num_slices + 1) * 4 + (double) nnz_ext * (value_bytes + 4) + (double) m * 4;
	return 0;
}

void
orc_sell_free(orc_sell_t * s)
{
	free(s->slice_ptr); free(s->ja); free(s->a); free(s->permutation); free(s->rev_permutation);
	memset(s, 0, sizeof(*s));
}

/* BENCH/spmv_kernels/sell_sorted.cpp:338-419: lane k of slice ii FMA-accumulates a[jj+k]*x[ja[jj+k]] over the
 * slice's columns in order and the result goes to y[rev_permutation[C*ii+k]]. */
#define SELL_BODY(T, FMA, CAST)                                                              \
	long ii, jj, k;                                                                      \
	T lanes[64];                                                                         \
	for (ii = 0; ii < s->num_slices; ii++)                                               \
	{                                                                                    \
		for (k = 0; k < s->C; k++)                                                   \
			lanes[k] = 0;                                                        \
		for (jj = s->slice_ptr[ii]; jj < s->slice_ptr[ii + 1]; jj += s->C)           \
			for (k = 0; k < s->C; k++)                                           \
				lanes[k] = FMA(CAST s->a[jj + k], x[s->ja[jj + k]], lanes[k]); \
		for (k = 0; k < s->C; k++)                                                   \
			if (s->C * ii + k < s->m)                                            \
				y[s->rev_permutation[s->C * ii + k]] = lanes[k];             \
	}

void
orc_sell_spmv_f64(const orc_sell_t * s, const double * x, double * y)
{
	SELL_BODY(double, fma, )
}

/* values were narrowed to float at construction in an fp32 build (sell_sorted.cpp:272: a[jj] = values_reordered[j]
 * with ValueType a) — the cast below performs the same narrowing. */
void
orc_sell_spmv_f32(const orc_sell_t * s, const float * x, float * y)
{
	SELL_BODY(float, fmaf, (float))
}

/* ================================================================================================= COO */

/* BENCH/spmv_kernels/mkl_coo.cpp:79-90 */
void
orc_csr_to_coo_rows(const int32_t * row_ptr, long m, int32_t * rowind)
{
	for (long i = 0; i < m; i++)
		for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++)
			rowind[j] = (int32_t) i;
}

/* Published COO definition (MKL mkl_cspblas_?coogemv: y := A*x, zero-based): y = 0; y[r] += v * x[c]. */
void
orc_coo_spmv_f64(const int32_t * rowind, const int32_t * colind, const double * val, long m, long nnz,
		const double * x, double * y)
{
	for (long i = 0; i < m; i++)
		y[i] = 0;
	for (long j = 0; j < nnz; j++)
		y[rowind[j]] = fma(val[j], x[colind[j]], y[rowind[j]]);
}

void
orc_coo_spmv_f32(const int32_t * rowind, const int32_t * colind, const float * val, long m, long nnz,
		const float * x, float * y)
{
	for (long i = 0; i < m; i++)
		y[i] = 0;
	for (long j = 0; j < nnz; j++)
		y[rowind[j]] = fmaf(val[j], x[colind[j]], y[rowind[j]]);
}

/* ======================================================================================== gold + metrics */

/* BENCH/bench_spmv.cpp:151-170: Kahan-compensated row sums in _Float128 (no fused operations in software quad). */
static void
gold_rows(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m, const double * x, __float128 * y_gold)
{
	#pragma omp parallel for
	for (long i = 0; i < m; i++)
	{
		__float128 sum = 0, compensation = 0, val, tmp;
		for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++)
		{
			val = (__float128) a[j] * (__float128) x[col_idx[j]] - compensation;
			tmp = sum + val;
			compensation = (tmp - sum) - val;
			sum = tmp;
		}
		y_gold[i] = sum;
	}
}

void
orc_gold_spmv(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y_gold_as_double)
{
	__float128 * g = (__float128 *) malloc((m > 0 ? m : 1) * sizeof(*g));
	gold_rows(row_ptr, col_idx, a, m, x, g);
	for (long i = 0; i < m; i++)
		y_gold_as_double[i] = (double) g[i];
	free(g);
}

/* BENCH/bench_spmv.cpp:108-235 + lib/array_metrics.c:1477-2149. */
double
orc_check_accuracy(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, const double * y_test, int is_double, double * metrics_out)
{
	__float128 epsilon = is_double ? 1e-10Q : 1e-7Q;
	__float128 * g = (__float128 *) malloc((m > 0 ? m : 1) * sizeof(*g));
	gold_rows(row_ptr, col_idx, a, m, x, g);
	__float128 maxDiff = 0;
	for (long i = 0; i < m; i++)
	{
		__float128 yt = y_test[i];
		__float128 diff = fabsq(g[i] - yt);
		if (g[i] > epsilon)     /* positive rows only: SURVEY.md Q10 */
		{
			diff = diff / fabsq(g[i]);
			if (diff > maxDiff)
				maxDiff = diff;
		}
	}
	if (metrics_out)
	{
		double mae = 0, max_ae = 0, mse = 0, mare = 0, smare = 0, lnq = 0;
		for (long i = 0; i < m; i++)
		{
			double A = (double) g[i], F = y_test[i];
			double ae = fabs(A - F);
			mae += ae;
			if (ae > max_ae) max_ae = ae;
			mse += (A - F) * (A - F);
			mare += ae / fmax(fabs(A), DBL_EPSILON);
			smare += ae / fmax(fabs(A) + fabs(F), DBL_EPSILON);
			lnq += log10(fmax(fabs(F), DBL_EPSILON)) - log10(fmax(fabs(A), DBL_EPSILON));
		}
		double N = (double) m;
		metrics_out[0] = mae / N;
		metrics_out[1] = max_ae;
		metrics_out[2] = mse / N;
		metrics_out[3] = 100.0 * mare / N;
		metrics_out[4] = 100.0 * smare / N;
		metrics_out[5] = lnq / N;
		long double e = metrics_out[5];
		metrics_out[6] = (double) log10l(fabsl(powl(10, e) - 1));
		metrics_out[7] = pow(10, metrics_out[6]);
	}
	free(g);
	return (double) maxDiff;
}

/* ============================================================================================== timing */

static int
cmp_double(const void * p, const void * q)
{
	double u = *(const double *) p, v = *(const double *) q;
	return (u > v) - (u < v);
}

/* Protocol of BENCH/bench_spmv.cpp:296-301,335-382: one warm-up call, then per-call CLOCK_MONOTONIC_RAW
 * (lib/time_it.h:35-57) until >= min_loops and >= min_runtime; median returned. Thread ranges are computed
 * once, as the reference does in its constructor (csr.cpp:82-146). */
double
orc_time_csr_spmv_f64(const int32_t * row_ptr, const int32_t * col_idx, const double * a, long m,
		const double * x, double * y, int num_threads, long min_loops, double min_runtime,
		long * loops_out, double * tmin_out, double * tmax_out)
{
	if (num_threads < 1)
		num_threads = 1;
	long nnz = row_ptr[m] - row_ptr[0];
	long * i_s = (long *) malloc(num_threads * sizeof(long));
	long * i_e = (long *) malloc(num_threads * sizeof(long));
	for (int t = 0; t < num_threads; t++)
		orc_partition_prefix_sums(num_threads, t, row_ptr, m, nnz, &i_s[t], &i_e[t]);
	long cap = 1 << 20, n = 0;
	double * tt = (double *) malloc(cap * sizeof(double));
	double total = 0;
	for (long it = -1; it < cap; it++)
	{
		struct timespec t0, t1;
		clock_gettime(CLOCK_MONOTONIC_RAW, &t0);
		#pragma omp parallel num_threads(num_threads)
		{
			int t = omp_get_thread_num();
			csr_rows_f64(row_ptr, col_idx, a, x, y, i_s[t], i_e[t]);
		}
		clock_gettime(CLOCK_MONOTONIC_RAW, &t1);
		if (it < 0)
			continue;   /* warm-up */
		tt[n] = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
		total += tt[n];
		n++;
		if (total >= min_runtime && n >= min_loops)
			break;
	}
	qsort(tt, n, sizeof(double), cmp_double);
	double med = tt[n / 2];
	if (loops_out) *loops_out = n;
	if (tmin_out) *tmin_out = tt[0];
	if (tmax_out) *tmax_out = tt[n - 1];
	free(tt); free(i_s); free(i_e);
	return med;
}


/* ---- the BSC SELL-C-sigma library's layout (row a7 of SURVEY §8a): benchmark_code/BENCH/src/spmv_kernels/sell_c_s.cpp:58-75 calls
 * sellcs_init_params(C = 256, sigma = 16384) + sellcs_create_matrix_from_CSR_rd (sell-C-s/RISC-V/sellcs_format.c:205-226 ->
 * csr_to_sellcs :137-200):
 *   - row sizes (sellcs_utils.c:36-48), then per window of sigma rows an LSD radix sort by size, DESCENDING, paired with the row
 *     ids (radix_sort.c:36-122: 4-bit digits, stable within equal sizes) -> row_order[sorted position] = original row;
 *   - slice s = sorted rows [s*C, (s+1)*C); its width = size of its FIRST row (sellcs_format.c:150-158);
 *   - values / column indices zero-initialised (:168-176) and written column-major: entry k of sorted row r at
 *     slice_pointers[s] + k*C + (r % C) (:178-189); padding stays (column 0, value 0.0); the last slice of a matrix whose row
 *     count is not a multiple of C keeps C lanes, the missing rows are all padding.
 * Arrays are caller-allocated: row_order[m], widths[nslices], slice_ptr[nslices+1]; col/val may be NULL to query nnz_ext. */
long
orc_sellcs_layout(const int32_t * row_ptr, const int32_t * col_idx, const double * values, long m, long C, long sigma,
		int32_t * row_order, int32_t * widths, int64_t * slice_ptr, int32_t * col, double * val)
{
	long nslices = (m + C - 1) / C, i, k, s;
	for (i = 0; i < m; i++)
		row_order[i] = (int32_t) i;
	/* a stable descending sort by size inside each window (what the paired LSD radix sort produces) */
	for (k = 0; k < m; k += sigma)
	{
		long e = k + sigma > m ? m : k + sigma, n = e - k, maxlen = 0, a;
		for (i = k; i < e; i++)
			if (row_ptr[i + 1] - row_ptr[i] > maxlen)
				maxlen = row_ptr[i + 1] - row_ptr[i];
		long * cnt = (long *) calloc((size_t) maxlen + 2, sizeof(long));
		int32_t * tmp = (int32_t *) malloc((size_t) (n > 0 ? n : 1) * sizeof(int32_t));
		for (i = k; i < e; i++)
			cnt[maxlen - (row_ptr[i + 1] - row_ptr[i]) + 1]++;
		for (a = 0; a <= maxlen; a++)
			cnt[a + 1] += cnt[a];
		for (i = k; i < e; i++)
			tmp[cnt[maxlen - (row_ptr[i + 1] - row_ptr[i])]++] = (int32_t) i;
		for (i = 0; i < n; i++)
			row_order[k + i] = tmp[i];
		free(cnt);
		free(tmp);
	}
	slice_ptr[0] = 0;
	for (s = 0; s < nslices; s++)
	{
		long first = row_order[s * C];
		widths[s] = row_ptr[first + 1] - row_ptr[first];
		slice_ptr[s + 1] = slice_ptr[s] + (int64_t) widths[s] * C;
	}
	if (col && val)
	{
		for (i = 0; i < slice_ptr[nslices]; i++)
		{
			col[i] = 0;
			val[i] = 0.0;
		}
		for (i = 0; i < m; i++)
		{
			long o = row_order[i];
			int64_t w = slice_ptr[i / C] + (i % C);
			for (k = row_ptr[o]; k < row_ptr[o + 1]; k++, w += C)
			{
				col[w] = col_idx[k];
				val[w] = values[k];
			}
		}
	}
	return (long) slice_ptr[nslices];
}
