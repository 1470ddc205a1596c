// extern "C" surface of libspmv_host.so (include/spmv_host.h).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>

#include "host.hpp"

namespace spmv_host {
const char * last_error();
}

using namespace spmv_host;

extern "C" {

const char *
spmv_host_last_error(void)
{
	return last_error();
}

void
spmv_host_free(void * p)
{
	free(p);
}

int
spmv_host_mtx_read(const char * filename, spmv_host_coo * out)
{
	return mtx_read(filename, out);
}

void
spmv_host_coo_free(spmv_host_coo * coo)
{
	free(coo->R);
	free(coo->C);
	free(coo->V);
	memset(coo, 0, sizeof(*coo));
}

int
spmv_host_coo_to_csr(const int32_t * R, const int32_t * C, const double * V, long m, long n, long nnz,
		int32_t * row_ptr, int32_t * col_idx, double * values)
{
	return coo_to_csr(R, C, V, m, n, nnz, row_ptr, col_idx, values);
}

int
spmv_host_mtx_write_csr(const char * filename, const int32_t * row_ptr, const int32_t * col_idx, const double * values,
		long m, long n)
{
	FILE * f = fopen(filename, "w");
	if (!f)
	{
		set_error("cannot create '%s'", filename);
		return 1;
	}
	fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%ld %ld %ld\n", m, n, (long) row_ptr[m]);
	for (long i = 0; i < m; i++)
		for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++)
			fprintf(f, "%ld %d %.17g\n", i + 1, col_idx[j] + 1, values[j]);
	fclose(f);
	return 0;
}

int
spmv_host_partition_iterations(long num_workers, long worker_pos, long start, long end, long * s, long * e)
{
	if (num_workers < 1 || worker_pos < 0 || worker_pos >= num_workers)
	{
		set_error("bad worker %ld of %ld", worker_pos, num_workers);
		return 1;
	}
	partition_iterations(num_workers, worker_pos, start, end, s, e);
	return 0;
}

int
spmv_host_partition_prefix_sums(long num_workers, long worker_pos, const int32_t * sums, long N, long total_sum, long * s, long * e)
{
	if (num_workers < 1 || worker_pos < 0 || worker_pos >= num_workers)
	{
		set_error("bad worker %ld of %ld", worker_pos, num_workers);
		return 1;
	}
	if (N < 1)
	{
		set_error("Empty Sums array.");       // parallel_util.h:166-167
		return 1;
	}
	partition_prefix_sums(num_workers, worker_pos, sums, N, total_sum, s, e);
	return 0;
}

void
spmv_host_csr_free(spmv_host_csr * csr)
{
	free(csr->row_ptr);
	free(csr->col_idx);
	free(csr->values);
	memset(csr, 0, sizeof(*csr));
}

int
spmv_host_gen_twin(long nr_rows, long nr_cols, double avg, double std, double bw, double skew, double neigh, double crs,
		unsigned long seed, int pattern, spmv_host_csr * out)
{
	return gen_twin(nr_rows, nr_cols, avg, std, bw, skew, neigh, crs, seed, pattern, out);
}

int
spmv_host_gen_kkt(long N, unsigned long seed, spmv_host_csr * out)
{
	return gen_kkt(N, seed, out);
}

int
spmv_host_gen_kkt_row_ptr(long N, int32_t * row_ptr, long * m_out, long * nnz_out)
{
	return gen_kkt_row_ptr(N, row_ptr, m_out, nnz_out);
}

int
spmv_host_gen_kkt_block(long N, unsigned long seed, long row_begin, long row_end, spmv_host_csr * out)
{
	return gen_kkt_block(N, seed, row_begin, row_end, out);
}

int
spmv_host_remap_columns(int32_t * col_idx, long nnz, const long * offsets, long parts, long padded)
{
	return remap_columns(col_idx, nnz, offsets, parts, padded);
}

int
spmv_host_column_ranges(const int32_t * col_idx, long nnz, long padded, long parts, long * lo, long * hi)
{
	return column_ranges(col_idx, nnz, padded, parts, lo, hi);
}

int
spmv_host_bfs_order(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, int32_t * order)
{
	return bfs_order(row_ptr, col_idx, m, n, order);
}

int
spmv_host_gen_kkt_rows(long N, unsigned long seed, const int32_t * rows, long count, spmv_host_csr * out)
{
	return gen_kkt_rows(N, seed, rows, count, out);
}

int
spmv_host_gen_kkt_rows_filtered(long N, unsigned long seed, const int32_t * rows, long row_begin, long count, long col_lo, long col_hi,
		int keep_inside, int32_t * row_ptr, int32_t * col_idx, double * values, long capacity)
{
	return gen_kkt_rows_filtered(N, seed, rows, row_begin, count, col_lo, col_hi, keep_inside ? 1 : 0, row_ptr, col_idx, values, capacity);
}

int
spmv_host_gen_kkt_rows_into(long N, unsigned long seed, const int32_t * rows, long row_begin, long count, int32_t * row_ptr, int32_t * col_idx,
		double * values, long capacity)
{
	return gen_kkt_rows_into(N, seed, rows, row_begin, count, row_ptr, col_idx, values, capacity);
}

int
spmv_host_jitter_columns(long m, long n, const int32_t * row_ptr, int32_t * col_idx, double * values, double frac, long span, unsigned long seed)
{
	return jitter_columns(m, n, row_ptr, col_idx, values, frac, span, seed);
}

int
spmv_host_kkt_bfs_owner(long N, long parts, int32_t * owner)
{
	return kkt_bfs_owner(N, parts, owner);
}

int
spmv_host_kkt_partition_volume(long N, const int32_t * owner, long parts, long * volume)
{
	return kkt_partition_volume(N, owner, parts, volume);
}

int
spmv_host_owners_from_order(const int32_t * row_ptr, long m, const int32_t * order, long parts, int32_t * owner)
{
	return owners_from_order(row_ptr, m, order, parts, owner);
}

int
spmv_host_partition_volume(const int32_t * row_ptr, const int32_t * col_idx, long m, const int32_t * owner, long parts, long * volume)
{
	return partition_volume(row_ptr, col_idx, m, owner, parts, volume);
}

int
spmv_host_partition_layout(const int32_t * row_ptr, const int32_t * col_idx, long m, const int32_t * owner, long parts,
		int32_t * perm, long * offsets)
{
	return partition_layout(row_ptr, col_idx, m, owner, parts, perm, offsets);
}

int
spmv_host_permuted_block(const int32_t * row_ptr, const int32_t * col_idx, const double * values, long m, const int32_t * perm,
		const int32_t * inv, long row_begin, long row_end, spmv_host_csr * out)
{
	return permuted_block(row_ptr, col_idx, values, m, perm, inv, row_begin, row_end, out);
}

int
spmv_host_halo_lists(const int32_t * row_ptr, const int32_t * col_idx, long m, const int32_t * owner, long parts, long rank,
		long * send_offsets, int32_t ** send_list, long * recv_offsets, int32_t ** recv_list)
{
	return halo_lists(row_ptr, col_idx, m, owner, parts, rank, send_offsets, send_list, recv_offsets, recv_list);
}

// Twin strings of benchmark_code/BENCH/config.sh:402,413,430,449 (seed 14 in every one).
int
spmv_host_gen_named(const char * name, double scale, spmv_host_csr * out)
{
	struct Twin { const char * name; long m; double avg, std, bw, skew, neigh, crs; int pattern; };
	static const Twin twins[] = {
		{"scircuit",         170998,  5.6078784547,  4.3921621102, 0.2972525308,   61.9471560146, 0.8033653966, 0.6330185674, 0},
		{"cant",              62451, 64.1684360539, 14.0562609915, 0.0086040976,    0.2155508969, 1.6157502290, 0.9147287701, 0},
		{"pwtk",             217918, 53.3889995319,  4.7438951025, 0.0593207019,    2.3714810462, 1.8783518634, 0.9498085123, 0},
		{"soc-LiveJournal1", 4847571, 14.2326482686, 36.0802804379, 0.3469818665, 1424.8063304206, 0.2842716835, 0.2817802019, 1},
	};
	if (!(scale > 0 && scale <= 1))
	{
		set_error("scale must be in (0,1]");
		return 1;
	}
	std::string nm(name);
	if (nm == "nlpkkt240")
	{
		long N = (long) floor(240 * cbrt(scale) + 0.5);
		return gen_kkt(N < 4 ? 4 : N, 14, out);
	}
	for (const Twin & t : twins)
		if (nm == t.name)
		{
			long m = (long) floor(t.m * scale + 0.5);
			if (m < 64)
				m = 64;
			// keep the absolute row span when the matrix is scaled down: bw_scaled grows as n shrinks (capped at 1)
			double bw = t.bw * (double) t.m / (double) m;
			if (bw > 1)
				bw = 1;
			double skew = t.skew;
			if (t.avg * (1 + skew) > m)
				skew = (double) m / t.avg - 1;
			return gen_twin(m, m, t.avg, t.std, bw, skew, t.neigh, t.crs, 14, t.pattern, out);
		}
	set_error("unknown matrix '%s' (cant, scircuit, pwtk, soc-LiveJournal1, nlpkkt240)", name);
	return 1;
}

int
spmv_host_csr_am_stats(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, double * out15, char * mem_range, long mem_range_n)
{
	return csr_am_stats(row_ptr, col_idx, m, n, out15, mem_range, mem_range_n);
}

int
spmv_host_csr_features(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, double * out7)
{
	return csr_features(row_ptr, col_idx, m, n, out7);
}

}  // extern "C"
