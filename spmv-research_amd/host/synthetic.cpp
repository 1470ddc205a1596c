// Synthetic stand-ins for the SuiteSparse matrices named in BASELINE.json. No .mtx file ships with the reference and
// there is no network, and the reference's own generator (artificial-matrix-generator/) is an un-vendored, empty
// submodule, so these generators are OURS. They consume the reference's feature vector (argument order of
// benchmark_code/BENCH/src/bench.cpp:569-579; the strings for cant / scircuit / pwtk / soc-LiveJournal1 are at
// benchmark_code/BENCH/config.sh:402,413,430,449; feature definitions lib/storage_formats/csr_util/csr_util_gen.c:
// 437-447,596-695,961) and produce sorted-column CSR directly, in parallel, deterministically per (seed, row).
//
// nlpkkt240 has no twin string: gen_kkt() builds a symmetric KKT-shaped matrix [H A^T; A 0] over an N^3 grid
// (m = 2N^3 + 6N^2 = 27,993,600 and ~770 M non-zeros for N = 240; 27-point H, two 7-point stencils per constraint row).

#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include <omp.h>

#include "host.hpp"
#include "kkt_grid.hpp"

namespace spmv_host {

double
Rng::normal()
{
	// Box-Muller, one value per call
	double u1 = uniform(), u2 = uniform();
	if (u1 < 1e-300)
		u1 = 1e-300;
	return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

static int
alloc_csr(spmv_host_csr * out, long m, long n, long nnz)
{
	out->m = m;
	out->n = n;
	out->nnz = nnz;
	out->row_ptr = (int32_t *) malloc(((size_t) m + 1) * sizeof(int32_t));
	out->col_idx = (int32_t *) malloc((size_t) std::max<long>(nnz, 1) * sizeof(int32_t));
	out->values = (double *) malloc((size_t) std::max<long>(nnz, 1) * sizeof(double));
	if (!out->row_ptr || !out->col_idx || !out->values)
	{
		set_error("out of memory for a %ld x %ld matrix with %ld non-zeros", m, n, nnz);
		return 1;
	}
	return 0;
}

// row length model: Normal(avg,std) when the spread is moderate, log-normal with the same mean/std when std > avg/2
// (power-law-ish matrices such as soc-LiveJournal1: avg 14.2, std 36), clipped to [0,n]; rows on a geometric ladder
// below avg*(1+skew) reproduce the `skew` feature (a few very long rows).
static long
row_length(Rng & g, double avg, double std, long n)
{
	double v;
	if (std <= 0.5 * avg)
		v = avg + std * g.normal();
	else
	{
		double s2 = log(1.0 + (std / avg) * (std / avg));
		v = exp(log(avg) - 0.5 * s2 + sqrt(s2) * g.normal());
	}
	long L = (long) floor(v + 0.5);
	return std::min(std::max(L, 0L), n);
}

static void
fill_row(Rng & g, long i, long m, long n, long L, double bw_scaled, double p_neigh, double crs,
		const int32_t * prev, long prev_len, std::vector<int32_t> & cols)
{
	cols.clear();
	if (L <= 0)
		return;
	// window of total span bw_scaled*n around the scaled diagonal
	double span = std::max(bw_scaled * (double) n, (double) L * 1.25);
	span = std::min(span, (double) n);
	double centre = (m > 1) ? (double) i * (double) (n - 1) / (double) (m - 1) : 0;
	long lo = (long) (centre - 0.5 * span), hi = lo + (long) span;
	if (lo < 0) { hi -= lo; lo = 0; }
	if (hi > n) { lo -= (hi - n); hi = n; }
	if (lo < 0) lo = 0;
	const long width = std::max(hi - lo, 1L);
	long attempts = 0;
	int32_t last = -1;
	while ((long) cols.size() < L && attempts < 8 * L + 64)
	{
		attempts++;
		int32_t c;
		double u = g.uniform();
		if (u < crs && prev_len > 0)
		{
			c = prev[g.below(prev_len)];                 // reuse a column of the previous row (cross-row similarity)
			if (g.uniform() < 0.25)
				c += (g.uniform() < 0.5) ? -1 : 1;
		}
		else if (last >= 0 && g.uniform() < p_neigh)
			c = last + 1;                                // extend a run (same-row neighbours)
		else
			c = (int32_t) (lo + g.below(width));
		if (c < 0 || c >= n)
			continue;
		cols.push_back(c);
		last = c;
		if ((long) cols.size() == L || (attempts & 63) == 0)
		{
			std::sort(cols.begin(), cols.end());
			cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
		}
	}
	std::sort(cols.begin(), cols.end());
	cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
	// top up deterministically if rejection sampling fell short (dense windows)
	for (long c = lo; (long) cols.size() < L && c < hi; c++)
		if (!std::binary_search(cols.begin(), cols.end(), (int32_t) c))
			cols.insert(std::upper_bound(cols.begin(), cols.end(), (int32_t) c), (int32_t) c);
	if ((long) cols.size() > L)
		cols.resize(L);
}

int
gen_twin(long m, long n, double avg, double std, double bw_scaled, double skew, double neigh, double crs,
		unsigned long seed, int pattern, spmv_host_csr * out)
{
	memset(out, 0, sizeof(*out));
	if (m <= 0 || n <= 0 || avg < 0)
	{
		set_error("bad twin parameters");
		return 1;
	}
	std::vector<long> len((size_t) m);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 4096)
	for (long i = 0; i < m; i++)
	{
		Rng g(seed, 2 * (uint64_t) i);
		len[i] = row_length(g, avg, std, n);
	}
	// skew: max row = avg*(1+skew); put a ladder max, max/2, max/4, ... on rows spread over the matrix
	long maxlen = std::min((long) floor(avg * (1.0 + skew) + 0.5), n);
	if (skew > 3.0)
	{
		long L = maxlen;
		for (int k = 0; L > 2 * (avg + 3 * std) && k < 24; k++, L = L * 2 / 3)
		{
			Rng g(seed ^ 0xABCDEF, (uint64_t) k);
			len[g.below(m)] = L;
		}
	}
	else
	{
		long cur = *std::max_element(len.begin(), len.end());
		if (maxlen > cur)
			len[m / 2] = maxlen;
	}
	long nnz = 0;
	for (long i = 0; i < m; i++)
		nnz += len[i];
	if (nnz >= 0x7fffffffL)
	{
		set_error("twin would have %ld non-zeros (int32 limit)", nnz);
		return 1;
	}
	if (alloc_csr(out, m, n, nnz))
		return 1;
	out->row_ptr[0] = 0;
	for (long i = 0; i < m; i++)
		out->row_ptr[i + 1] = out->row_ptr[i] + (int32_t) len[i];
	// p(extend run): each extension adds 2 to the neighbour count of its two ends -> avg_num_neighbours ~ 2p/(1-p) capped
	double p_neigh = neigh / (2.0 + neigh);
	p_neigh = std::min(std::max(p_neigh, 0.0), 0.95);
	const double p_crs = std::min(std::max(crs, 0.0), 0.98);
	// rows are generated in blocks so that "previous row" is available without a serial dependency across blocks
	const long BLK = 256;
	#pragma omp parallel num_threads(spmv::host_threads())
	{
		std::vector<int32_t> cols, prev;
		#pragma omp for schedule(dynamic, 16)
		for (long b = 0; b < (m + BLK - 1) / BLK; b++)
		{
			prev.clear();
			for (long i = b * BLK; i < std::min(m, (b + 1) * BLK); i++)
			{
				Rng g(seed, 2 * (uint64_t) i + 1);
				long L = out->row_ptr[i + 1] - out->row_ptr[i];
				// very long rows span the whole matrix
				double bw = (L > 8 * (avg + 1)) ? 1.0 : bw_scaled;
				fill_row(g, i, m, n, L, bw, p_neigh, (L > 8 * (avg + 1)) ? 0.0 : p_crs, prev.data(), (long) prev.size(), cols);
				// fill_row guarantees exactly L columns whenever the window holds L of them; otherwise pad from 0 up
				for (int32_t c = 0; (long) cols.size() < L; c++)
					if (!std::binary_search(cols.begin(), cols.end(), c))
						cols.insert(std::upper_bound(cols.begin(), cols.end(), c), c);
				int32_t * ci = out->col_idx + out->row_ptr[i];
				double * va = out->values + out->row_ptr[i];
				for (long k = 0; k < L; k++)
				{
					ci[k] = cols[k];
					va[k] = pattern ? 1.0 : g.uniform(0.25, 1.0);      // one sign: row sums with x = ones do not cancel (see sym_value)
				}
				prev = cols;
			}
		}
	}
	return 0;
}

// ------------------------------------------------------------------------------------------------------- KKT
static inline double
sym_value(uint64_t seed, long i, long j)
{
	long a = std::min(i, j), b = std::max(i, j);
	Rng g(seed, (uint64_t) a * 0x100000001B3ull + (uint64_t) b);
	// Off-diagonal values of ONE sign: the reference driver checks y against its quad-precision gold RELATIVE to |y_gold| with
	// x = ones (bench_spmv.cpp:173-199), i.e. relative to the row sum; with values of both signs some of 28 M row sums cancel to
	// 1e-7 of their terms and any correctly rounded fp64 kernel — the reference's own CPU one included — prints `Test failed!`.
	return (i == j) ? 4.0 : g.uniform(0.25, 1.0);
}

// global row_ptr only (m+1 entries): lets every rank of a row-partitioned run find its block without building A
int
gen_kkt_row_ptr(long N, int32_t * row_ptr, long * m_out, long * nnz_out)
{
	Grid G;
	if (make_grid(N, G))
		return 1;
	const long m = G.n1 + G.n2;
	if (m_out)
		*m_out = m;
	if (!row_ptr)
		return 0;
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 8192)
	for (long i = 0; i < m; i++)
		row_ptr[i + 1] = kkt_row_len(G, i);
	long acc = 0;
	row_ptr[0] = 0;
	for (long i = 0; i < m; i++)
	{
		acc += row_ptr[i + 1];
		if (acc >= 0x7fffffffL)
		{
			set_error("KKT matrix too large for int32 indices");
			return 1;
		}
		row_ptr[i + 1] = (int32_t) acc;
	}
	if (nnz_out)
		*nnz_out = acc;
	return 0;
}

// the rows rows[0..lm) (or r0 .. r0+lm when rows == nullptr) of the KKT matrix as a local CSR (row_ptr starts at 0, columns global)
static int
gen_kkt_some(long N, unsigned long seed, const int32_t * rows, long r0, long lm, spmv_host_csr * out)
{
	memset(out, 0, sizeof(*out));
	Grid G;
	if (make_grid(N, G))
		return 1;
	const long m = G.n1 + G.n2;
	auto row_of = [&](long li) { return rows ? (long) rows[li] : r0 + li; };
	std::vector<int32_t> len((size_t) std::max<long>(lm, 1));
	long bad = -1;
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 8192) reduction(max : bad)
	for (long i = 0; i < lm; i++)
	{
		const long r = row_of(i);
		if (r < 0 || r >= m)
		{
			bad = std::max(bad, i);
			len[i] = 0;
		}
		else
			len[i] = kkt_row_len(G, r);
	}
	if (bad >= 0)
	{
		set_error("KKT row list: entry %ld is outside [0,%ld)", bad, m);
		return 1;
	}
	long nnz = 0;
	for (long i = 0; i < lm; i++)
		nnz += len[i];
	if (nnz >= 0x7fffffffL || m + nnz >= 0x7fffffffL)
	{
		set_error("KKT block too large for int32 indices (rows=%ld nnz=%ld)", lm, nnz);
		return 1;
	}
	if (alloc_csr(out, lm, m, nnz))
		return 1;
	out->row_ptr[0] = 0;
	for (long i = 0; i < lm; i++)
		out->row_ptr[i + 1] = out->row_ptr[i] + len[i];
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 8192)
	for (long li = 0; li < lm; li++)
	{
		const long i = row_of(li);
		int32_t * ci = out->col_idx + out->row_ptr[li];
		double * va = out->values + out->row_ptr[li];
		const int k = kkt_row_cols(G, i, ci);
		for (int q = 0; q < k; q++)
			va[q] = sym_value(seed, i, ci[q]);
	}
	return 0;
}

// the same rows written into CALLER-allocated arrays (row_ptr[count+1], col_idx[nnz], values[nnz] or NULL for the structure alone):
// a rank of a multi-GPU run knows its rows' lengths from the global row_ptr and never needs a second copy of its block
int
gen_kkt_rows_into(long N, unsigned long seed, const int32_t * rows, long r0, long count, int32_t * row_ptr, int32_t * col_idx, double * values,
		long capacity)
{
	Grid G;
	if (make_grid(N, G))
		return 1;
	const long m = G.n1 + G.n2;
	auto row_of = [&](long li) { return rows ? (long) rows[li] : r0 + li; };
	long acc = 0;
	row_ptr[0] = 0;
	for (long i = 0; i < count; i++)
	{
		const long r = row_of(i);
		if (r < 0 || r >= m)
		{
			set_error("KKT row list: entry %ld is outside [0,%ld)", i, m);
			return 1;
		}
		acc += kkt_row_len(G, r);
		if (acc >= 0x7fffffffL || acc > capacity)
		{
			set_error("KKT rows: %ld non-zeros exceed the caller's arrays (%ld) or the int32 index range", acc, capacity);
			return 1;
		}
		row_ptr[i + 1] = (int32_t) acc;
	}
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 8192)
	for (long li = 0; li < count; li++)
	{
		const long i = row_of(li);
		int32_t * ci = col_idx + row_ptr[li];
		const int k = kkt_row_cols(G, i, ci);
		if (values)
			for (int q = 0; q < k; q++)
				values[row_ptr[li] + q] = sym_value(seed, i, ci[q]);
	}
	return 0;
}

// Rows with a COLUMN FILTER applied while they are generated: keep_inside = 1 keeps columns in [col_lo, col_hi), 0 keeps the others —
// the local-column / remote-column halves of a rank's row block (SURVEY §8e overlap scheme) without ever holding the unfiltered
// block. col_idx == NULL: only row_ptr (the filtered lengths) is computed, so that the caller can size its arrays exactly.
int
gen_kkt_rows_filtered(long N, unsigned long seed, const int32_t * rows, long r0, long count, long col_lo, long col_hi, int keep_inside,
		int32_t * row_ptr, int32_t * col_idx, double * values, long capacity)
{
	Grid G;
	if (make_grid(N, G))
		return 1;
	const long m = G.n1 + G.n2;
	auto row_of = [&](long li) { return rows ? (long) rows[li] : r0 + li; };
	long bad = -1;
	row_ptr[0] = 0;
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 8192) reduction(max : bad)
	for (long li = 0; li < count; li++)
	{
		const long r = row_of(li);
		if (r < 0 || r >= m)
		{
			bad = std::max(bad, li);
			row_ptr[li + 1] = 0;
			continue;
		}
		int32_t tmp[64];
		const int k = kkt_row_cols(G, r, tmp);
		int kept = 0;
		for (int q = 0; q < k; q++)
			kept += ((tmp[q] >= col_lo && tmp[q] < col_hi) ? 1 : 0) == keep_inside;
		row_ptr[li + 1] = kept;
	}
	if (bad >= 0)
	{
		set_error("KKT row list: entry %ld is outside [0,%ld)", bad, m);
		return 1;
	}
	long acc = 0;
	for (long li = 0; li < count; li++)
	{
		acc += row_ptr[li + 1];
		if (acc >= 0x7fffffffL)
		{
			set_error("KKT rows: more than 2^31 non-zeros");
			return 1;
		}
		row_ptr[li + 1] = (int32_t) acc;
	}
	if (!col_idx)
		return 0;
	if (acc > capacity)
	{
		set_error("KKT rows: %ld non-zeros exceed the caller's arrays (%ld)", acc, capacity);
		return 1;
	}
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 8192)
	for (long li = 0; li < count; li++)
	{
		const long i = row_of(li);
		int32_t tmp[64];
		const int k = kkt_row_cols(G, i, tmp);
		long o = row_ptr[li];
		for (int q = 0; q < k; q++)
			if (((tmp[q] >= col_lo && tmp[q] < col_hi) ? 1 : 0) == keep_inside)
			{
				col_idx[o] = tmp[q];
				if (values)
					values[o] = sym_value(seed, i, tmp[q]);
				o++;
			}
	}
	return 0;
}

int
gen_kkt_block(long N, unsigned long seed, long r0, long r1, spmv_host_csr * out)
{
	Grid G;
	if (make_grid(N, G))
		return 1;
	if (r0 < 0 || r1 > G.n1 + G.n2 || r0 > r1)
	{
		set_error("bad KKT row block [%ld,%ld) of %ld", r0, r1, G.n1 + G.n2);
		return 1;
	}
	return gen_kkt_some(N, seed, nullptr, r0, r1 - r0, out);
}

int
gen_kkt_rows(long N, unsigned long seed, const int32_t * rows, long count, spmv_host_csr * out)
{
	if (count < 0 || (count > 0 && !rows))
	{
		set_error("gen_kkt_rows: bad row list");
		return 1;
	}
	return gen_kkt_some(N, seed, rows, 0, count, out);
}

// Perturb a generated matrix so that its column patterns stop being translation-invariant: in a fraction `frac` of the rows
// every off-diagonal column moves by a random offset in [-span, span] (kept inside [0,n), kept off the diagonal, the row
// re-sorted, collisions inside a row resolved by leaving the later entry where it was). Values stay attached to their
// entries. Used to measure how much of the compressed-index SELL format's advantage is a property of the perfectly regular
// twin (bench.py --jitter).
int
jitter_columns(long m, long n, const int32_t * row_ptr, int32_t * col_idx, double * values, double frac, long span, unsigned long seed)
{
	if (!(frac >= 0 && frac <= 1) || span < 1)
	{
		set_error("jitter_columns: frac must be in [0,1] and span >= 1");
		return 1;
	}
	#pragma omp parallel num_threads(spmv::host_threads())
	{
		std::vector<std::pair<int32_t, double>> row;
		#pragma omp for schedule(static, 4096)
		for (long i = 0; i < m; i++)
		{
			Rng g(seed ^ 0x5EEDull, (uint64_t) i);
			if (g.uniform() >= frac)
				continue;
			const long s0 = row_ptr[i], L = row_ptr[i + 1] - s0;
			row.resize((size_t) L);
			for (long k = 0; k < L; k++)
				row[(size_t) k] = std::make_pair(col_idx[s0 + k], values[s0 + k]);
			for (long k = 0; k < L; k++)
			{
				const int32_t c = row[(size_t) k].first;
				if (c == i)
					continue;
				long nc = (long) c + g.below(2 * span + 1) - span;
				if (nc < 0 || nc >= n || nc == i)
					continue;
				bool taken = false;
				for (long q = 0; q < L && !taken; q++)
					taken = q != k && row[(size_t) q].first == (int32_t) nc;
				if (!taken)
					row[(size_t) k].first = (int32_t) nc;
			}
			std::sort(row.begin(), row.end(), [](const std::pair<int32_t, double> & a, const std::pair<int32_t, double> & b) { return a.first < b.first; });
			for (long k = 0; k < L; k++)
			{
				col_idx[s0 + k] = row[(size_t) k].first;
				values[s0 + k] = row[(size_t) k].second;
			}
		}
	}
	return 0;
}

int
gen_kkt(long N, unsigned long seed, spmv_host_csr * out)
{
	Grid G;
	if (make_grid(N, G))
		return 1;
	return gen_kkt_block(N, seed, 0, G.n1 + G.n2, out);
}

// Row-partitioned runs keep x as P slices padded to a common length (one equal-sized RCCL allgather, no compaction
// pass): column c owned by part p moves to p*padded + (c - offsets[p]).
int
remap_columns(int32_t * col_idx, long nnz, const long * offsets, long parts, long padded)
{
	long bad = 0;
	#pragma omp parallel for num_threads(spmv::host_threads()) reduction(+ : bad)
	for (long j = 0; j < nnz; j++)
	{
		long c = col_idx[j];
		long lo = 0, hi = parts;          // largest p with offsets[p] <= c
		while (hi - lo > 1)
		{
			long mid = (lo + hi) / 2;
			if (offsets[mid] <= c)
				lo = mid;
			else
				hi = mid;
		}
		long v = lo * padded + (c - offsets[lo]);
		if (c < offsets[0] || c >= offsets[parts] || c - offsets[lo] >= padded || v >= 0x7fffffffL)
			bad++;
		else
			col_idx[j] = (int32_t) v;
	}
	if (bad)
	{
		set_error("%ld column indices outside the partition / padded slice", bad);
		return 1;
	}
	return 0;
}

// For a block whose columns live in the padded slice layout: the sub-range [lo[q], hi[q]) of every part q's slice that
// the block references at all (hi <= lo: nothing). Lets a rank fetch only those ranges instead of whole slices.
int
column_ranges(const int32_t * col_idx, long nnz, long padded, long parts, long * lo, long * hi)
{
	for (long q = 0; q < parts; q++)
	{
		lo[q] = padded;
		hi[q] = 0;
	}
	int bad = 0;
	#pragma omp parallel num_threads(spmv::host_threads())
	{
		std::vector<long> tlo((size_t) parts, padded), thi((size_t) parts, 0);
		#pragma omp for nowait
		for (long j = 0; j < nnz; j++)
		{
			long c = col_idx[j];
			long q = c / padded;
			if (c < 0 || q >= parts)
			{
				bad = 1;
				continue;
			}
			long l = c - q * padded;
			if (l < tlo[q]) tlo[q] = l;
			if (l + 1 > thi[q]) thi[q] = l + 1;
		}
		#pragma omp critical
		for (long q = 0; q < parts; q++)
		{
			lo[q] = std::min(lo[q], tlo[q]);
			hi[q] = std::max(hi[q], thi[q]);
		}
	}
	if (bad)
	{
		set_error("column index outside the padded layout");
		return 1;
	}
	return 0;
}

}  // namespace spmv_host
