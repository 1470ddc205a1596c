// Internal header of libspmv_host.so (host side of the MI355X SpMV engine; see include/spmv_host.h).
#pragma once

#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/spmv_host.h"
#include "../csrc/host_threads.hpp"

namespace spmv_host {

void set_error(const char * fmt, ...);

// lib/parallel_util.h:47-91 (increment +1) and :156-184 of the reference
void partition_iterations(long num_workers, long worker_pos, long start, long end, long * s, long * e);
void partition_prefix_sums(long num_workers, long worker_pos, const int32_t * sums, long N, long total_sum, long * s, long * e);

// A file's bytes: either the mmap of the file itself or a malloc'ed buffer holding the decompressed / extracted text
// (file_load.cpp; replaces lib/parallel_io.c:28-130 of the reference).
struct FileBuf {
	const char * data = nullptr;
	size_t size = 0;
	bool mapped = false;
	void release();
};
int file_load(const char * path, FileBuf & out);

int mtx_read(const char * filename, spmv_host_coo * out);
int coo_to_csr(const int32_t * R, const int32_t * C, const double * V, long m, long n, long nnz,
		int32_t * row_ptr, int32_t * col_idx, double * values);

int gen_twin(long nr_rows, long nr_cols, double avg, double std, double bw_scaled, double skew, double neigh,
		double crs, unsigned long seed, int pattern, spmv_host_csr * out);
int gen_kkt(long N, unsigned long seed, spmv_host_csr * out);
int gen_kkt_row_ptr(long N, int32_t * row_ptr, long * m_out, long * nnz_out);
int gen_kkt_block(long N, unsigned long seed, long r0, long r1, spmv_host_csr * out);
int gen_kkt_rows(long N, unsigned long seed, const int32_t * rows, long count, spmv_host_csr * out);
int gen_kkt_rows_filtered(long N, unsigned long seed, const int32_t * rows, long r0, long count, long col_lo, long col_hi, int keep_inside,
		int32_t * row_ptr, int32_t * col_idx, double * values, long capacity);
int gen_kkt_rows_into(long N, unsigned long seed, const int32_t * rows, long r0, long count, int32_t * row_ptr, int32_t * col_idx, double * values,
		long capacity);
int jitter_columns(long m, long n, const int32_t * row_ptr, int32_t * col_idx, double * values, double frac, long span, unsigned long seed);
int kkt_bfs_owner(long N, long parts, int32_t * owner);
int kkt_partition_volume(long N, const int32_t * owner, long parts, long * volume);
int column_ranges(const int32_t * col_idx, long nnz, long padded, long parts, long * lo, long * hi);
int remap_columns(int32_t * col_idx, long nnz, const long * offsets, long parts, long padded);
int bfs_order(const int32_t * rp, const int32_t * ci, long m, long n, int32_t * order);
int owners_from_order(const int32_t * rp, long m, const int32_t * order, long parts, int32_t * owner);
int partition_volume(const int32_t * rp, const int32_t * ci, long m, const int32_t * owner, long parts, long * volume);
int partition_layout(const int32_t * rp, const int32_t * ci, long m, const int32_t * owner, long parts, int32_t * perm, long * offsets);
int permuted_block(const int32_t * rp, const int32_t * ci, const double * va, long m, const int32_t * perm, const int32_t * inv,
		long r0, long r1, spmv_host_csr * out);
int halo_lists(const int32_t * rp, const int32_t * ci, long m, const int32_t * owner, long parts, long rank,
		long * send_offsets, int32_t ** send_list, long * recv_offsets, int32_t ** recv_list);
int csr_features(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, double * out7);
int csr_am_stats(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, double * out15, char * mem_range, long mem_range_n);

// counter-based generator: independent stream per (seed, row) so the generators are parallel AND deterministic
struct Rng {
	uint64_t s;
	explicit Rng(uint64_t seed, uint64_t stream) : s(seed * 0x9E3779B97F4A7C15ull + stream * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull) { next(); next(); }
	inline uint64_t next()
	{
		uint64_t z = (s += 0x9E3779B97F4A7C15ull);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}
	inline double uniform() { return (double) (next() >> 11) * (1.0 / 9007199254740992.0); }      // [0,1)
	inline double uniform(double a, double b) { return a + (b - a) * uniform(); }
	inline long below(long n) { return (long) (uniform() * (double) n); }
	double normal();
};

}  // namespace spmv_host
