// Matrix-Market loader of the MI355X SpMV engine — parallel, with the reference loader's observable behaviour
// (lib/storage_formats/matrix_market/matrix_market.c:150-323,420-454 and matrix_market_gen.c:65-202):
//   * the file is split at '\n' into NON-EMPTY lines (a lone '\r' counts as a line);
//   * no "%%MatrixMarket" banner -> silently "coordinate real general";
//   * only `matrix coordinate` is usable on the SpMV path (the reference also parses `array`, but its driver needs
//     coordinates: bench.cpp:180-224); symmetry in {general, symmetric, skew-symmetric, Hermitian}, case-sensitive;
//   * the number of lines after the size line must equal the declared nnz;
//   * indices 1-based -> 0-based int32; real -> strtod, integer -> strtol narrowed to int, complex -> |z|,
//     pattern -> 1.0;
//   * symmetric files are expanded: entries [0,nnz_sym) in file order, then every off-diagonal mirrored (skew: negated)
//     in file order; nnz = 2*nnz_non_diag + nnz_diag.
// Numbers are converted with std::from_chars (correctly rounded, same value as strtod); anything from_chars rejects
// (hex floats, "+x", leading blanks) goes through strtod/strtol so acceptance matches the reference.

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <errno.h>
#include <math.h>
#include <complex>
#include <charconv>
#include <algorithm>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <omp.h>

#include "host.hpp"

namespace spmv_host {

static thread_local char g_err[1024] = "";

void
set_error(const char * fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

const char *
last_error()
{
	return g_err;
}

static inline bool
is_ws(char c)
{
	return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f';
}

// parse a long at p (< end); returns pointer past the number (== p when nothing was parsed)
static inline const char *
parse_long(const char * p, const char * end, long * out)
{
	const char * q = p;
	while (q < end && is_ws(*q))
		q++;
	const char * num = q;
	if (num < end && *num == '+')
		num++;
	auto r = std::from_chars(num, end, *out, 10);
	if (r.ec == std::errc())
		return r.ptr;
	// fallback: strtol on a bounded copy (same acceptance as the reference's GENLIB_safe_strtol)
	char buf[128];
	size_t len = std::min<size_t>(sizeof(buf) - 1, (size_t) (end - p));
	memcpy(buf, p, len);
	buf[len] = 0;
	char * e;
	*out = strtol(buf, &e, 10);
	return p + (e - buf);
}

static inline const char *
parse_double(const char * p, const char * end, double * out)
{
	const char * q = p;
	while (q < end && is_ws(*q))
		q++;
	const char * num = q;
	if (num < end && *num == '+')
		num++;
	auto r = std::from_chars(num, end, *out, std::chars_format::general);
	if (r.ec == std::errc())
		return r.ptr;
	char buf[512];
	size_t len = std::min<size_t>(sizeof(buf) - 1, (size_t) (end - p));
	memcpy(buf, p, len);
	buf[len] = 0;
	char * e;
	*out = strtod(buf, &e);
	return p + (e - buf);
}

struct Line { const char * s; const char * e; };

// next non-empty line in [p,end): returns false at end of buffer
static inline bool
next_line(const char *& p, const char * end, Line & ln)
{
	while (p < end)
	{
		const char * nl = (const char *) memchr(p, '\n', end - p);
		const char * le = nl ? nl : end;
		// the reference also treats NUL as a delimiter; text files do not contain it, ignore
		if (le > p)
		{
			ln.s = p;
			ln.e = le;
			p = nl ? nl + 1 : end;
			return true;
		}
		p = nl ? nl + 1 : end;
	}
	return false;
}

static std::string
token(const char *& p, const char * e)
{
	while (p < e && (is_ws(*p)))
		p++;
	const char * s = p;
	while (p < e && !is_ws(*p))
		p++;
	return std::string(s, p);
}

int
mtx_read(const char * filename, spmv_host_coo * out)
{
	memset(out, 0, sizeof(*out));
	FileBuf file;                                   // plain, .gz, .zst, .tar and combinations (file_load.cpp)
	if (file_load(filename, file))
		return 1;
	const size_t N = file.size;
	if (N == 0)
	{
		set_error("Error parsing MARKET matrix '%s': empty file", filename);
		return 1;
	}
	const char * buf = file.data;
	const char * end = buf + N;
	int rc = 1;
	int32_t * R = NULL, * C = NULL;
	double * V = NULL;
	do
	{
		// ---- header
		const char * p = buf;
		Line ln;
		if (!next_line(p, end, ln))
		{
			set_error("Error parsing MARKET matrix '%s': empty file", filename);
			break;
		}
		int symmetric = 0, skew = 0, herm = 0;
		std::string format = "coordinate", field = "real";
		{
			const char * q = ln.s;
			std::string t0 = token(q, ln.e);
			if (t0 == "%%MatrixMarket")
			{
				std::string object = token(q, ln.e);
				format = token(q, ln.e);
				field = token(q, ln.e);
				std::string symmetry = token(q, ln.e);
				if (object != "matrix" || (format != "coordinate" && format != "array"))
				{
					set_error("Error parsing MARKET matrix '%s': only allow matrix coordinate or array format", filename);
					break;
				}
				if (symmetry == "symmetric") symmetric = 1;
				else if (symmetry == "skew-symmetric") { symmetric = 1; skew = 1; }
				else if (symmetry == "Hermitian") { symmetric = 1; herm = 1; }
				else if (symmetry == "general") symmetric = 0;
				else
				{
					set_error("Error parsing MARKET matrix '%s': unsupported symmetry type: %s", filename, symmetry.c_str());
					break;
				}
				if (!next_line(p, end, ln))
				{
					set_error("Error parsing MARKET matrix '%s': invalid/missing matrix sizes", filename);
					break;
				}
			}
		}
		bool ok = true;
		while (ln.s[0] == '%')
			if (!next_line(p, end, ln))
			{
				ok = false;
				break;
			}
		if (!ok)
		{
			set_error("Error parsing MARKET matrix '%s': invalid/missing matrix sizes", filename);
			break;
		}
		if (format != "coordinate")
		{
			set_error("Error parsing MARKET matrix '%s': array format is not supported on the SpMV path", filename);
			break;
		}
		long M, Nc, nnz_sym;
		{
			const char * q = ln.s;
			const char * a = parse_long(q, ln.e, &M);
			const char * b = (a > q) ? parse_long(a, ln.e, &Nc) : a;
			const char * c = (b > a) ? parse_long(b, ln.e, &nnz_sym) : b;
			if (!(a > q && b > a && c > b))
			{
				set_error("Error parsing MARKET matrix '%s': invalid/missing matrix sizes: %.*s", filename, (int) (ln.e - ln.s), ln.s);
				break;
			}
		}
		const bool is_real = field == "real", is_int = field == "integer", is_cplx = field == "complex", is_pat = field == "pattern";
		if (!(is_real || is_int || is_cplx || is_pat))
		{
			set_error("Error parsing MARKET matrix '%s': unrecognized field type: %s", filename, field.c_str());
			break;
		}
		if (M < 0 || Nc < 0 || nnz_sym < 0 || M >= 0x7fffffffL || Nc >= 0x7fffffffL)
		{
			set_error("Error parsing MARKET matrix '%s': sizes out of the int32 index range", filename);
			break;
		}

		// ---- split the data region into chunks at line boundaries, count non-empty lines per chunk
		const char * data = p;
		const int T = spmv::host_threads();
		std::vector<const char *> cs(T + 1);
		for (int t = 0; t <= T; t++)
		{
			const char * q = data + (size_t) ((double) (end - data) * t / T);
			if (t == 0)
				q = data;
			else if (t == T)
				q = end;
			else
			{
				// advance to the character after the next newline (chunk starts at a line start)
				const char * nl = (q > data && q[-1] == '\n') ? q - 1 : (const char *) memchr(q, '\n', end - q);
				q = nl ? nl + 1 : end;
			}
			cs[t] = q;
		}
		for (int t = 1; t <= T; t++)
			if (cs[t] < cs[t - 1])
				cs[t] = cs[t - 1];
		std::vector<long> cnt(T + 1, 0);
		#pragma omp parallel num_threads(T)
		{
			int t = omp_get_thread_num();
			const char * q = cs[t];
			Line l;
			long c = 0;
			while (next_line(q, cs[t + 1], l))
				c++;
			cnt[t + 1] = c;
		}
		for (int t = 0; t < T; t++)
			cnt[t + 1] += cnt[t];
		if (cnt[T] != nnz_sym)
		{
			set_error("Error parsing MARKET matrix '%s': remaining number of file lines (%ld) don't match the number of non-zeros (%ld)",
					filename, cnt[T], nnz_sym);
			break;
		}
		const long nnz_alloc = symmetric ? 2 * nnz_sym : nnz_sym;
		if (nnz_alloc >= 0x7fffffffL)
		{
			set_error("Error parsing MARKET matrix '%s': more than 2^31-1 non-zeros (INT_T is int32)", filename);
			break;
		}
		R = (int32_t *) malloc(std::max<long>(nnz_alloc, 1) * sizeof(int32_t));
		C = (int32_t *) malloc(std::max<long>(nnz_alloc, 1) * sizeof(int32_t));
		V = (double *) malloc(std::max<long>(nnz_alloc, 1) * sizeof(double));
		if (!R || !C || !V)
		{
			set_error("out of memory for %ld entries", nnz_alloc);
			break;
		}

		// ---- parse the entries
		std::vector<long> off_diag(T + 1, 0);
		#pragma omp parallel num_threads(T)
		{
			int t = omp_get_thread_num();
			const char * q = cs[t];
			Line l;
			long i = cnt[t], nd = 0;
			while (next_line(q, cs[t + 1], l))
			{
				long r = 0, c = 0;
				const char * a = parse_long(l.s, l.e, &r);
				const char * b = parse_long(a, l.e, &c);
				R[i] = (int32_t) r - 1;
				C[i] = (int32_t) c - 1;
				if (is_real)
					parse_double(b, l.e, &V[i]);
				else if (is_int)
				{
					long v = 0;
					parse_long(b, l.e, &v);
					V[i] = (double) (int) v;
				}
				else if (is_cplx)
				{
					double re = 0, im = 0;
					const char * d = parse_double(b, l.e, &re);
					parse_double(d, l.e, &im);
					V[i] = std::abs(std::complex<double>(re, im));     // cabs(): matrix_market.c:436
				}
				else
					V[i] = 1.0;
				if (C[i] != R[i])
					nd++;
				i++;
			}
			off_diag[t + 1] = nd;
		}
		for (int t = 0; t < T; t++)
			off_diag[t + 1] += off_diag[t];
		const long non_diag = off_diag[T];
		const long nnz_diag = nnz_sym - non_diag;
		const long nnz = symmetric ? 2 * non_diag + nnz_diag : nnz_sym;

		// ---- symmetry expansion (mirrored entries in file order; |z| is unchanged by conj / negation)
		if (symmetric)
		{
			#pragma omp parallel num_threads(T)
			{
				int t = omp_get_thread_num();
				long j = nnz_sym + off_diag[t];
				for (long i = cnt[t]; i < cnt[t + 1]; i++)
					if (C[i] != R[i])
					{
						R[j] = C[i];
						C[j] = R[i];
						V[j] = (skew && !is_cplx && !is_pat) ? -V[i] : V[i];   // pattern: dummy 1.0 is filled AFTER the expansion (matrix_market.c:308-317)
						j++;
					}
			}
		}
		out->m = M; out->n = Nc; out->nnz = nnz; out->nnz_sym = nnz_sym;
		out->nnz_diag = nnz_diag; out->nnz_non_diag = non_diag;
		out->symmetric = symmetric; out->skew = skew; out->hermitian = herm;
		snprintf(out->field, sizeof(out->field), "%s", field.c_str());
		out->R = R; out->C = C; out->V = V;
		R = C = NULL;
		V = NULL;
		rc = 0;
	} while (0);
	file.release();
	free(R); free(C); free(V);
	return rc;
}

}  // namespace spmv_host
