/* TEST INFRASTRUCTURE — NOT PRODUCT CODE. See oracle.h.
 *
 * Sequential restatement of the reference's Matrix-Market loader and COO->CSR converter:
 *   lib/parallel_io.c:210-263 + lib/string_util.c:26-34,79-170  (file -> non-empty '\n'-separated lines)
 *   lib/storage_formats/matrix_market/matrix_market.c:150-255    (header)
 *   lib/storage_formats/matrix_market/matrix_market_gen.c:131-202 (coordinate data, 1-based -> 0-based)
 *   lib/storage_formats/matrix_market/matrix_market_gen.c:65-128  (symmetry expansion)
 *   lib/storage_formats/matrix_market/matrix_market.c:420-454     (values -> real)
 *   lib/storage_formats/csr/csr_gen.c:99-213                      (coo_to_csr + csr_sort_columns)
 * Behavioural quirks reproduced on purpose are listed in SURVEY.md §8 Q5-Q9.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <errno.h>
#include <complex.h>

#include "oracle.h"

#define FAIL(...) do { snprintf(err, err_n, __VA_ARGS__); goto fail; } while (0)

/* lib/genlib.h:407-433: strtol/strtod on a NUL-terminated copy of the remaining line; returns chars consumed
 * (0 = no number). Lines are already NUL-terminated here, so the copy is unnecessary. */
static long
parse_long(const char * s, long * out)
{
	char * end;
	*out = strtol(s, &end, 10);
	return end - s;
}

static long
parse_double(const char * s, double * out)
{
	char * end;
	*out = strtod(s, &end);
	return end - s;
}

int
orc_mtx_read(const char * filename, int num_threads, orc_coo_t * out, char * err, long err_n)
{
	FILE * f = NULL;
	char * buf = NULL;
	char ** lines = NULL;
	int32_t * R = NULL, * C = NULL;
	double * V = NULL;
	double * Vim = NULL;
	(void) num_threads;   /* the mirrored-entry order does not depend on the thread count (see below) */
	memset(out, 0, sizeof(*out));

	f = fopen(filename, "rb");
	if (!f)
		FAIL("cannot open '%s'", filename);
	fseek(f, 0, SEEK_END);
	long N = ftell(f);
	fseek(f, 0, SEEK_SET);
	buf = (char *) malloc(N + 1);
	if (fread(buf, 1, N, f) != (size_t) N)
		FAIL("short read on '%s'", filename);
	buf[N] = 0;
	fclose(f);
	f = NULL;

	/* tokenise into non-empty lines ('\n' or NUL delimited; a lone '\r' is NOT empty, as in the reference) */
	long num_lines = 0, cap = 1024, i, j;
	lines = (char **) malloc(cap * sizeof(*lines));
	for (i = 0; i < N;)
	{
		for (j = i; j < N && buf[j] != '\n' && buf[j] != 0; j++)
			;
		buf[j] = 0;
		if (j > i)
		{
			if (num_lines == cap)
			{
				cap *= 2;
				lines = (char **) realloc(lines, cap * sizeof(*lines));
			}
			lines[num_lines++] = buf + i;
		}
		i = j + 1;
	}
	if (num_lines == 0)
		FAIL("empty file");

	/* header: matrix_market.c:150-255 */
	char tok[5][1000];
	int symmetric = 0, skew = 0, herm = 0;
	char format[1000] = "coordinate", field[1000] = "real";
	long li = 0;
	{
		int nc = 0, pos = 0, k;
		for (k = 0; k < 5; k++)
			tok[k][0] = 0;
		sscanf(lines[0], "%999s%n", tok[0], &nc);
		if (strcmp(tok[0], "%%MatrixMarket") == 0)
		{
			pos = nc;
			for (k = 1; k < 5; k++)
			{
				nc = 0;
				if (sscanf(lines[0] + pos, "%999s%n", tok[k], &nc) < 1)
					break;
				pos += nc;
			}
			if (strcmp(tok[1], "matrix") || (strcmp(tok[2], "coordinate") && strcmp(tok[2], "array")))
				FAIL("only allow matrix coordinate or array format");
			strcpy(format, tok[2]);
			strcpy(field, tok[3]);
			li = 1;
			if (!strcmp(tok[4], "symmetric")) symmetric = 1;
			else if (!strcmp(tok[4], "skew-symmetric")) { symmetric = 1; skew = 1; }
			else if (!strcmp(tok[4], "Hermitian")) { symmetric = 1; herm = 1; }
			else if (!strcmp(tok[4], "general")) symmetric = 0;
			else FAIL("unsupported symmetry type: %s", tok[4]);
		}
		/* no banner: silently 'coordinate real general', line 0 is then a comment or the size line (Q5) */
	}
	while (li < num_lines && lines[li][0] == '%')
		li++;
	if (li >= num_lines)
		FAIL("invalid/missing matrix sizes");
	if (strcmp(format, "coordinate"))
		FAIL("array format is not supported on the SpMV path (bench.cpp:180-224 needs coordinates)");
	long M, Nc, nnz_sym;
	if (sscanf(lines[li++], "%ld%ld%ld", &M, &Nc, &nnz_sym) != 3)
		FAIL("invalid/missing matrix sizes: %s", lines[li - 1]);
	if (nnz_sym != num_lines - li)
		FAIL("remaining number of file lines (%ld) don't match the number of non-zeros (%ld)", num_lines - li, nnz_sym);

	int is_real = !strcmp(field, "real"), is_int = !strcmp(field, "integer");
	int is_cplx = !strcmp(field, "complex"), is_pat = !strcmp(field, "pattern");
	if (!(is_real || is_int || is_cplx || is_pat))
		FAIL("unrecognized field type: %s", field);

	long nnz_alloc = symmetric ? 2 * nnz_sym : nnz_sym;
	R = (int32_t *) malloc((nnz_alloc > 0 ? nnz_alloc : 1) * sizeof(*R));
	C = (int32_t *) malloc((nnz_alloc > 0 ? nnz_alloc : 1) * sizeof(*C));
	V = (double *) malloc((nnz_alloc > 0 ? nnz_alloc : 1) * sizeof(*V));
	if (is_cplx)
		Vim = (double *) malloc((nnz_alloc > 0 ? nnz_alloc : 1) * sizeof(*Vim));

	/* coordinate data: matrix_market_gen.c:150-185 */
	long non_diag = 0;
	for (i = 0; i < nnz_sym; i++)
	{
		const char * p = lines[li + i];
		long r, c, k, len;
		k = parse_long(p, &r);
		len = parse_long(p + k, &c);
		k += len;
		R[i] = (int32_t) r - 1;
		C[i] = (int32_t) c - 1;
		if (is_real)
			parse_double(p + k, &V[i]);
		else if (is_int)
		{
			long v;
			parse_long(p + k, &v);
			V[i] = (double) (int) v;     /* stored as int, widened later: matrix_market.c:424-431 */
		}
		else if (is_cplx)
		{
			len = parse_double(p + k, &V[i]);
			parse_double(p + k + len, &Vim[i]);
		}
		else
			V[i] = 1.0;                  /* pattern_dummy_vals: matrix_market.c:308-317 */
		if (C[i] != R[i])
			non_diag++;
	}
	long nnz_diag = nnz_sym - non_diag;
	long nnz = symmetric ? 2 * non_diag + nnz_diag : nnz_sym;

	/* symmetry expansion: matrix_market_gen.c:65-128. Thread t mirrors its contiguous chunk of the file entries to
	 * offset nnz_sym + (number of off-diagonals in earlier chunks): the result is "off-diagonal entries in file
	 * order", independent of the thread count. */
	if (symmetric)
	{
		j = nnz_sym;
		for (i = 0; i < nnz_sym; i++)
			if (C[i] != R[i])
			{
				R[j] = C[i];
				C[j] = R[i];
				V[j] = (skew && !is_pat) ? -V[i] : V[i];   /* pattern values are filled with 1.0 after the expansion */
				if (is_cplx)
					Vim[j] = skew ? Vim[i] : -Vim[i];   /* -conj(z) / conj(z) */
				j++;
			}
	}
	/* values -> real: matrix_market.c:420-454 (complex -> magnitude) */
	if (is_cplx)
		for (i = 0; i < nnz; i++)
			V[i] = cabs(V[i] + Vim[i] * I);

	out->m = M; out->n = Nc; out->nnz = nnz; out->nnz_sym = nnz_sym;
	out->nnz_diag = nnz_diag; out->nnz_non_diag = non_diag;
	out->symmetric = symmetric; out->skew = skew; out->hermitian = herm;
	snprintf(out->field, sizeof(out->field), "%s", field);
	out->R = R; out->C = C; out->V = V;
	free(Vim); free(lines); free(buf);
	return 0;
fail:
	if (f) fclose(f);
	free(buf); free(lines); free(R); free(C); free(V); free(Vim);
	return 1;
}

void
orc_coo_free(orc_coo_t * coo)
{
	free(coo->R); free(coo->C); free(coo->V);
	memset(coo, 0, sizeof(*coo));
}

/* csr_gen.c:178-213 + :99-174. The reference buckets entries by row with atomics (placement inside a row is not
 * deterministic) and then sorts each row by column (quicksort, or a stable bucket sort for very long rows), keeping
 * duplicates. The observable contract is: rows ascending, columns ascending inside a row, duplicates kept, and the
 * order among exact (row,col) duplicates unspecified. This restatement is the stable version of that contract
 * (duplicates keep input order). */
void
orc_coo_to_csr(const int32_t * R, const int32_t * C, const double * V, long m, long n, long nnz,
		int32_t * row_ptr, int32_t * col_idx, double * values)
{
	long i, j;
	(void) n;
	for (i = 0; i <= m; i++)
		row_ptr[i] = 0;
	for (j = 0; j < nnz; j++)
		row_ptr[R[j] + 1]++;
	for (i = 0; i < m; i++)
		row_ptr[i + 1] += row_ptr[i];
	int32_t * fill = (int32_t *) malloc((m > 0 ? m : 1) * sizeof(*fill));
	for (i = 0; i < m; i++)
		fill[i] = row_ptr[i];
	for (j = 0; j < nnz; j++)
	{
		long pos = fill[R[j]]++;
		col_idx[pos] = C[j];
		values[pos] = V[j];
	}
	free(fill);
	/* stable insertion/merge sort per row on (col) */
	for (i = 0; i < m; i++)
	{
		long s = row_ptr[i], e = row_ptr[i + 1], len = e - s;
		if (len < 2)
			continue;
		if (len <= 32)
		{
			for (long a = s + 1; a < e; a++)
			{
				int32_t c = col_idx[a];
				double v = values[a];
				long b = a - 1;
				while (b >= s && col_idx[b] > c)
				{
					col_idx[b + 1] = col_idx[b];
					values[b + 1] = values[b];
					b--;
				}
				col_idx[b + 1] = c;
				values[b + 1] = v;
			}
		}
		else
		{
			/* bottom-up stable merge sort */
			int32_t * tc = (int32_t *) malloc(len * sizeof(*tc));
			double * tv = (double *) malloc(len * sizeof(*tv));
			int32_t * sc = col_idx + s;
			double * sv = values + s;
			for (long w = 1; w < len; w *= 2)
			{
				for (long lo = 0; lo < len; lo += 2 * w)
				{
					long mid = lo + w < len ? lo + w : len;
					long hi = lo + 2 * w < len ? lo + 2 * w : len;
					long a = lo, b = mid, k = lo;
					while (a < mid && b < hi)
					{
						if (sc[b] < sc[a]) { tc[k] = sc[b]; tv[k++] = sv[b++]; }
						else               { tc[k] = sc[a]; tv[k++] = sv[a++]; }
					}
					while (a < mid) { tc[k] = sc[a]; tv[k++] = sv[a++]; }
					while (b < hi)  { tc[k] = sc[b]; tv[k++] = sv[b++]; }
				}
				memcpy(sc, tc, len * sizeof(*tc));
				memcpy(sv, tv, len * sizeof(*tv));
			}
			free(tc); free(tv);
		}
	}
}
