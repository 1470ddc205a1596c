// COO -> CSR with the reference's observable contract (lib/storage_formats/csr/csr_gen.c:99-213, called as
// coo_to_csr(R,C,V,m,n,nnz,ia,ja,a, sort_columns=1, transpose=0) from bench.cpp:224): rows ascending, columns ascending
// inside a row, duplicates KEPT (not summed). The reference places entries with atomics, so the order among exact
// (row,col) duplicates is unspecified there; here it is the input order (deterministic).
// Also: CSR -> Matrix-Market writer and the structural features of csr_util_gen.c used to describe synthetic twins.

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include <omp.h>

#include "host.hpp"

namespace spmv_host {

int
coo_to_csr(const int32_t * R, const int32_t * C, const double * V, long m, long n, long nnz,
		int32_t * row_ptr, int32_t * col_idx, double * values)
{
	for (long j = 0; j < nnz; j++)        // validate before scattering
		if (R[j] < 0 || R[j] >= m || C[j] < 0 || C[j] >= n)
		{
			set_error("entry %ld = (%d,%d) outside a %ld x %ld matrix", j, R[j] + 1, C[j] + 1, m, n);
			return 1;
		}
	std::vector<int32_t> cnt((size_t) m + 1, 0);
	#pragma omp parallel for num_threads(spmv::host_threads())
	for (long j = 0; j < nnz; j++)
		__atomic_fetch_add(&cnt[R[j]], 1, __ATOMIC_RELAXED);
	row_ptr[0] = 0;
	for (long i = 0; i < m; i++)
		row_ptr[i + 1] = row_ptr[i] + cnt[i];
	// scatter (order inside a row arbitrary), remembering the input position so that the per-row sort is total
	std::vector<int32_t> fill(row_ptr, row_ptr + m);
	std::vector<int32_t> src((size_t) std::max<long>(nnz, 1));
	#pragma omp parallel for num_threads(spmv::host_threads())
	for (long j = 0; j < nnz; j++)
	{
		int32_t pos = __atomic_fetch_add(&fill[R[j]], 1, __ATOMIC_RELAXED);
		src[pos] = (int32_t) j;
	}
	#pragma omp parallel num_threads(spmv::host_threads())
	{
		std::vector<std::pair<int32_t, int32_t>> key;
		#pragma omp for schedule(dynamic, 1024)
		for (long i = 0; i < m; i++)
		{
			const long s = row_ptr[i], e = row_ptr[i + 1], len = e - s;
			if (len == 0)
				continue;
			key.resize(len);
			for (long k = 0; k < len; k++)
				key[k] = {C[src[s + k]], src[s + k]};
			std::sort(key.begin(), key.end());
			for (long k = 0; k < len; k++)
			{
				col_idx[s + k] = key[k].first;
				values[s + k] = V[key[k].second];
			}
		}
	}
	return 0;
}

int
csr_features(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, double * out)
{
	const long nnz = row_ptr[m];
	double sum_bw = 0, sum_neigh = 0, sum_sim = 0, sum_sq = 0;
	long nonempty = 0, maxlen = 0;
	const double avg = m > 0 ? (double) nnz / m : 0;
	#pragma omp parallel for num_threads(spmv::host_threads()) reduction(+ : sum_bw, sum_neigh, sum_sim, sum_sq, nonempty) reduction(max : maxlen) schedule(dynamic, 4096)
	for (long i = 0; i < m; i++)
	{
		const long s = row_ptr[i], e = row_ptr[i + 1], d = e - s;
		sum_sq += (d - avg) * (d - avg);
		maxlen = std::max(maxlen, d);
		if (d == 0)
			continue;
		nonempty++;
		sum_bw += col_idx[e - 1] - col_idx[s];                 // csr_util_gen.c:437-447 (columns sorted)
		for (long j = s; j < e; j++)                             // csr_util_gen.c:596-633, window 1
			for (long k = j + 1; k < e && col_idx[k] - col_idx[j] <= 1; k++)
				sum_neigh += 2;
		long l = i + 1;                                          // csr_util_gen.c:636-695, window 1
		while (l < m && row_ptr[l + 1] == row_ptr[l])
			l++;
		if (l < m)
		{
			long k = row_ptr[l], k_e = row_ptr[l + 1], sim = 0;
			for (long j = s; j < e; j++)
				while (k < k_e)
				{
					long diff = (long) col_idx[k] - col_idx[j];
					if (labs(diff) <= 1) { sim++; break; }
					if (diff <= 0) k++;
					else break;
				}
			sum_sim += (double) sim / d;
		}
	}
	out[0] = avg;
	out[1] = m > 0 ? sqrt(sum_sq / m) : 0;
	out[2] = (m > 0 && n > 0) ? sum_bw / m / n : 0;
	out[3] = avg > 0 ? (maxlen - avg) / avg : 0;
	out[4] = nnz > 0 ? sum_neigh / nnz : 0;
	out[5] = nonempty > 0 ? sum_sim / nonempty : 0;
	out[6] = (double) maxlen;
	return 0;
}

// The statistics the reference's artificial-matrix mode prints beside every result (bench_spmv.cpp:489-563: the fields of the
// generator's struct csr_matrix). The generator itself is an un-vendored submodule, so the definitions are taken from where the
// tree still states them: per-row bandwidth b = col_max - col_min and scatter s = degree / b (0 when b = 0)
// (lib/storage_formats/csr_util/csr_util_gen.c:437-449), density in percent and the CSR footprint in MiB with fp64 values and the
// power-of-two memory class (benchmark_code/FPGA/csr_to_vitis_converter/v2/artificial_matrix_generation.py:130-132,243-256);
// "scaled" = the same quantity with the bandwidth divided by the number of columns.
//   out[0..14] = density, mem_footprint, avg_nnz_per_row, std_nnz_per_row, avg_bw, std_bw, avg_bw_scaled, std_bw_scaled, avg_sc, std_sc,
//                avg_sc_scaled, std_sc_scaled, skew, avg_num_neighbours, cross_row_similarity;   mem_range = "[lo-hi]" (MiB)
int
csr_am_stats(const int32_t * row_ptr, const int32_t * col_idx, long m, long n, double * out, char * mem_range, long mem_range_n)
{
	double f[7];
	csr_features(row_ptr, col_idx, m, n, f);
	const long nnz = row_ptr[m];
	double sb = 0, sbb = 0, ss = 0, sss = 0;
	#pragma omp parallel for num_threads(spmv::host_threads()) reduction(+ : sb, sbb, ss, sss) schedule(static, 4096)
	for (long i = 0; i < m; i++)
	{
		const long d = row_ptr[i + 1] - row_ptr[i];
		if (d == 0)
			continue;
		const double b = (double) col_idx[row_ptr[i + 1] - 1] - col_idx[row_ptr[i]];
		const double sc = b > 0 ? (double) d / b : 0;
		sb += b;
		sbb += b * b;
		ss += sc;
		sss += sc * sc;
	}
	const double mm = m > 0 ? (double) m : 1, nn = n > 0 ? (double) n : 1;
	const double avg_bw = sb / mm, std_bw = sqrt(std::max(0.0, sbb / mm - avg_bw * avg_bw));
	const double avg_sc = ss / mm, std_sc = sqrt(std::max(0.0, sss / mm - avg_sc * avg_sc));
	const double footprint = ((64.0 + 32.0) * nnz + 32.0 * (m + 1)) / (8.0 * 1024 * 1024);
	out[0] = (m > 0 && n > 0) ? (double) nnz / ((double) m * n) * 100.0 : 0;
	out[1] = footprint;
	out[2] = f[0];
	out[3] = f[1];
	out[4] = avg_bw;
	out[5] = std_bw;
	out[6] = avg_bw / nn;
	out[7] = std_bw / nn;
	out[8] = avg_sc;
	out[9] = std_sc;
	out[10] = avg_sc * nn;
	out[11] = std_sc * nn;
	out[12] = f[3];
	out[13] = f[4];
	out[14] = f[5];
	if (mem_range && mem_range_n > 0)
	{
		long lo = 4, hi = 8;
		while (hi < 4096 && footprint >= hi)
		{
			lo = hi;
			hi *= 2;
		}
		if (footprint < 4 || footprint >= 4096)
			snprintf(mem_range, mem_range_n, "[-]");
		else
			snprintf(mem_range, mem_range_n, "[%ld-%ld]", lo, hi);
	}
	return 0;
}

}  // namespace spmv_host
