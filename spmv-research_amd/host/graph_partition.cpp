// Communication-aware row partition for the multi-GPU SpMV. The reference partitions rows for its threads
// by nnz-balanced CONTIGUOUS ranges (lib/parallel_util.h:156-184) — shared memory, so which x entries a thread reads
// costs nothing. Across GPUs every remote x entry is xGMI traffic: a KKT matrix [H A^T; A 0] cut into contiguous row
// ranges makes the constraint rows read the whole H part of x. The same nnz balance applied to a breadth-first ORDER of
// the matrix graph gives each GPU a slab of neighbouring vertices, and only the slab surfaces are exchanged.
//
//   bfs_order          vertex order of a breadth-first sweep from a pseudo-peripheral vertex (all components)
//   owners_from_order  nnz-balanced contiguous cuts of that order -> owner[v]
//   partition_volume   remote x entries each part reads under an owner map (the figure the two partitions are compared on)
//   partition_layout   new numbering: parts in order; inside a part first the vertices other parts read, grouped by the
//                      lowest such part, then the interior — both in ORIGINAL order, so that a part keeps the matrix's own
//                      locality and every peer's needs lie in a short prefix of the slice (sent in place, no packing)
//   permuted_block     rows [r0,r1) of P A P^T as a local CSR, columns ascending in the new numbering
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <utility>
#include <vector>
#include <omp.h>

#include "host.hpp"
#include "kkt_grid.hpp"
#include "../csrc/host_threads.hpp"

namespace spmv_host {

// The graph is seen through an adjacency provider: row(v, buf, out) gives the neighbours of v — a pointer into the CSR for a
// stored matrix, or columns computed on the fly for the analytic KKT matrix (kkt_grid.hpp), whose 28 M-vertex graph is then
// partitioned without the 9 GB matrix ever being built.
struct CsrAdj {
	const int32_t * rp;
	const int32_t * ci;
	inline long row(long v, int32_t *, const int32_t *& out) const
	{
		out = ci + rp[v];
		return rp[v + 1] - rp[v];
	}
};

struct KktAdj {
	Grid G;
	inline long row(long v, int32_t * buf, const int32_t *& out) const
	{
		out = buf;
		return kkt_row_cols(G, v, buf);
	}
};

// one sweep from `start` over the vertices not yet marked; appends to order[tail...]; returns the new tail
template <class Adj>
static long
sweep(const Adj & adj, long start, unsigned char * seen, int32_t * order, long tail)
{
	long head = tail;
	seen[start] = 1;
	order[tail++] = (int32_t) start;
	int32_t buf[64];
	while (head < tail)
	{
		const long v = order[head++];
		const int32_t * nb;
		const long L = adj.row(v, buf, nb);
		for (long j = 0; j < L; j++)
		{
			const int32_t c = nb[j];
			if (!seen[c])
			{
				seen[c] = 1;
				order[tail++] = c;
			}
		}
	}
	return tail;
}

template <class Adj>
static int
bfs_order_t(const Adj & adj, long m, int32_t * order)
{
	if (m == 0)
		return 0;
	std::vector<unsigned char> seen((size_t) m, 0);
	// the far end of a sweep from vertex 0 is (nearly) peripheral: sweeping from there gives thin, long level sets
	long tail = sweep(adj, 0, seen.data(), order, 0);
	const long far = order[tail - 1], first_component = tail;
	for (long k = 0; k < first_component; k++)
		seen[order[k]] = 0;
	tail = sweep(adj, far, seen.data(), order, 0);
	for (long v = 0; v < m && tail < m; v++)
		if (!seen[v])
			tail = sweep(adj, v, seen.data(), order, tail);
	if (tail != m)
	{
		set_error("bfs_order: visited %ld of %ld vertices", tail, m);
		return 1;
	}
	return 0;
}

int
bfs_order(const int32_t * rp, const int32_t * ci, long m, long n, int32_t * order)
{
	if (m != n)
	{
		set_error("bfs_order: the graph partition needs a square matrix (got %ld x %ld)", m, n);
		return 1;
	}
	return bfs_order_t(CsrAdj{rp, ci}, m, order);
}

int
owners_from_order(const int32_t * rp, long m, const int32_t * order, long parts, int32_t * owner)
{
	if (parts < 1)
	{
		set_error("owners_from_order: parts must be >= 1");
		return 1;
	}
	const long total = (long) rp[m] - rp[0];
	long before = 0;
	for (long k = 0; k < m; k++)
	{
		const long v = order[k];
		// weight = non-zeros, one extra per row so that empty rows spread too
		const long p = (total + m) > 0 ? (long) (((__int128) (before + k) * parts) / (total + m)) : 0;
		owner[v] = (int32_t) std::min(p, parts - 1);
		before += rp[v + 1] - rp[v];
	}
	return 0;
}

template <class Adj>
static int
partition_volume_t(const Adj & adj, long m, const int32_t * owner, long parts, long * volume)
{
	const long words = (m + 63) / 64;
	std::vector<unsigned long long> bits((size_t) (parts * words), 0ull);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4096)
	for (long i = 0; i < m; i++)
	{
		const long p = owner[i];
		unsigned long long * b = bits.data() + p * words;
		int32_t buf[64];
		const int32_t * nb;
		const long L = adj.row(i, buf, nb);
		for (long j = 0; j < L; j++)
		{
			const long c = nb[j];
			if (owner[c] != p)
			{
				const unsigned long long bit = 1ull << (c & 63);
				if (!(__atomic_load_n(&b[c >> 6], __ATOMIC_RELAXED) & bit))
					__atomic_fetch_or(&b[c >> 6], bit, __ATOMIC_RELAXED);
			}
		}
	}
	for (long p = 0; p < parts; p++)
	{
		long cnt = 0;
		const unsigned long long * b = bits.data() + p * words;
		#pragma omp parallel for num_threads(spmv::host_threads()) reduction(+ : cnt)
		for (long w = 0; w < words; w++)
			cnt += __builtin_popcountll(b[w]);
		volume[p] = cnt;
	}
	return 0;
}

int
partition_volume(const int32_t * rp, const int32_t * ci, long m, const int32_t * owner, long parts, long * volume)
{
	return partition_volume_t(CsrAdj{rp, ci}, m, owner, parts, volume);
}

// ---- the same two steps for the analytic KKT matrix of edge N, matrix-free
int
kkt_bfs_owner(long N, long parts, int32_t * owner)
{
	KktAdj adj;
	if (make_grid(N, adj.G))
		return 1;
	const long m = adj.G.n1 + adj.G.n2;
	std::vector<int32_t> order((size_t) m), rp((size_t) m + 1);
	long nnz = 0;
	if (gen_kkt_row_ptr(N, rp.data(), nullptr, &nnz) || bfs_order_t(adj, m, order.data()))
		return 1;
	return owners_from_order(rp.data(), m, order.data(), parts, owner);
}

int
kkt_partition_volume(long N, const int32_t * owner, long parts, long * volume)
{
	KktAdj adj;
	if (make_grid(N, adj.G))
		return 1;
	return partition_volume_t(adj, adj.G.n1 + adj.G.n2, owner, parts, volume);
}

int
partition_layout(const int32_t * rp, const int32_t * ci, long m, const int32_t * owner, long parts, int32_t * perm, long * offsets)
{
	// lowest part (other than the owner) that reads x[v]; `parts` = nobody: interior
	std::vector<int32_t> reader((size_t) m, (int32_t) parts);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4096)
	for (long i = 0; i < m; i++)
	{
		const int32_t p = owner[i];
		for (long j = rp[i]; j < rp[i + 1]; j++)
		{
			const long c = ci[j];
			if (owner[c] == p)
				continue;
			int32_t cur = __atomic_load_n(&reader[c], __ATOMIC_RELAXED);
			while (p < cur && !__atomic_compare_exchange_n(&reader[c], &cur, p, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
				;
		}
	}
	// stable counting sort on (owner, reader)
	const long classes = parts + 1;
	std::vector<long> start((size_t) (parts * classes + 1), 0);
	for (long v = 0; v < m; v++)
	{
		if (owner[v] < 0 || owner[v] >= parts)
		{
			set_error("partition_layout: owner[%ld] = %d outside [0,%ld)", v, owner[v], parts);
			return 1;
		}
		start[(size_t) (owner[v] * classes + reader[v]) + 1]++;
	}
	for (long b = 0; b < parts * classes; b++)
		start[b + 1] += start[b];
	for (long p = 0; p <= parts; p++)
		offsets[p] = p < parts ? start[(size_t) (p * classes)] : m;
	for (long v = 0; v < m; v++)
		perm[start[(size_t) (owner[v] * classes + reader[v])]++] = (int32_t) v;
	return 0;
}

int
permuted_block(const int32_t * rp, const int32_t * ci, const double * va, long m, const int32_t * perm, const int32_t * inv,
		long r0, long r1, spmv_host_csr * out)
{
	if (r0 < 0 || r1 < r0 || r1 > m)
	{
		set_error("permuted_block: rows [%ld,%ld) outside [0,%ld)", r0, r1, m);
		return 1;
	}
	const long lm = r1 - r0;
	out->m = lm;
	out->n = m;
	out->row_ptr = (int32_t *) malloc(((size_t) lm + 1) * sizeof(int32_t));
	if (!out->row_ptr)
	{
		set_error("permuted_block: out of memory");
		return 1;
	}
	long nnz = 0;
	out->row_ptr[0] = 0;
	for (long r = 0; r < lm; r++)
	{
		const long o = perm[r0 + r];
		nnz += rp[o + 1] - rp[o];
		if (nnz > 0x7fffffffL)
		{
			free(out->row_ptr);
			out->row_ptr = nullptr;
			set_error("permuted_block: more than 2^31-1 non-zeros in one block");
			return 1;
		}
		out->row_ptr[r + 1] = (int32_t) nnz;
	}
	out->nnz = nnz;
	out->col_idx = (int32_t *) malloc((size_t) std::max<long>(nnz, 1) * sizeof(int32_t));
	out->values = (double *) malloc((size_t) std::max<long>(nnz, 1) * sizeof(double));
	if (!out->col_idx || !out->values)
	{
		free(out->row_ptr);
		free(out->col_idx);
		free(out->values);
		out->row_ptr = out->col_idx = nullptr;
		out->values = nullptr;
		set_error("permuted_block: out of memory for %ld non-zeros", nnz);
		return 1;
	}
	#pragma omp parallel num_threads(spmv::host_threads())
	{
		std::vector<std::pair<int32_t, double>> row;
		#pragma omp for schedule(dynamic, 1024)
		for (long r = 0; r < lm; r++)
		{
			const long o = perm[r0 + r];
			const long L = rp[o + 1] - rp[o];
			row.resize((size_t) L);
			for (long k = 0; k < L; k++)
				row[k] = std::make_pair(inv[ci[rp[o] + k]], va[rp[o] + k]);
			// stable: duplicates of one column keep their input order, as the reference's CSR does (csr_gen.c:99-213)
			std::stable_sort(row.begin(), row.end(), [](const std::pair<int32_t, double> & a, const std::pair<int32_t, double> & b) { return a.first < b.first; });
			int32_t * c = out->col_idx + out->row_ptr[r];
			double * v = out->values + out->row_ptr[r];
			for (long k = 0; k < L; k++)
			{
				c[k] = row[k].first;
				v[k] = row[k].second;
			}
		}
	}
	return 0;
}

}  // namespace spmv_host

namespace spmv_host {

// Halo index lists of one rank under an owner map, in ORIGINAL vertex numbers, each list ascending:
//   send[q] = vertices owned by `rank` that rows owned by q read      (what `rank` packs and sends to q)
//   recv[q] = vertices owned by q that rows owned by `rank` read      (where what q sends is scattered to)
// send[q] of rank p and recv[p] of rank q are the same set by construction, so the packed buffers need no header.
int
halo_lists(const int32_t * rp, const int32_t * ci, long m, const int32_t * owner, long parts, long rank,
		long * send_offsets, int32_t ** send_list, long * recv_offsets, int32_t ** recv_list)
{
	if (rank < 0 || rank >= parts)
	{
		set_error("halo_lists: rank %ld outside [0,%ld)", rank, parts);
		return 1;
	}
	const long words = (m + 63) / 64;
	std::vector<unsigned long long> sbits((size_t) (parts * words), 0ull), rbits((size_t) words, 0ull);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4096)
	for (long i = 0; i < m; i++)
	{
		const long q = owner[i];
		if (q == rank)
		{
			for (long j = rp[i]; j < rp[i + 1]; j++)
			{
				const long c = ci[j];
				if (owner[c] != rank)
				{
					const unsigned long long bit = 1ull << (c & 63);
					if (!(__atomic_load_n(&rbits[c >> 6], __ATOMIC_RELAXED) & bit))
						__atomic_fetch_or(&rbits[c >> 6], bit, __ATOMIC_RELAXED);
				}
			}
		}
		else
		{
			unsigned long long * b = sbits.data() + q * words;
			for (long j = rp[i]; j < rp[i + 1]; j++)
			{
				const long c = ci[j];
				if (owner[c] == rank)
				{
					const unsigned long long bit = 1ull << (c & 63);
					if (!(__atomic_load_n(&b[c >> 6], __ATOMIC_RELAXED) & bit))
						__atomic_fetch_or(&b[c >> 6], bit, __ATOMIC_RELAXED);
				}
			}
		}
	}
	std::vector<int32_t> s, r;
	std::vector<std::vector<int32_t>> rq((size_t) parts);
	for (long w = 0; w < words; w++)
		for (unsigned long long v = rbits[w]; v; v &= v - 1)
		{
			const long c = w * 64 + __builtin_ctzll(v);
			rq[(size_t) owner[c]].push_back((int32_t) c);
		}
	send_offsets[0] = recv_offsets[0] = 0;
	for (long q = 0; q < parts; q++)
	{
		const unsigned long long * b = sbits.data() + q * words;
		for (long w = 0; w < words; w++)
			for (unsigned long long v = b[w]; v; v &= v - 1)
				s.push_back((int32_t) (w * 64 + __builtin_ctzll(v)));
		send_offsets[q + 1] = (long) s.size();
		r.insert(r.end(), rq[(size_t) q].begin(), rq[(size_t) q].end());
		recv_offsets[q + 1] = (long) r.size();
	}
	*send_list = (int32_t *) malloc(std::max<size_t>(s.size(), 1) * sizeof(int32_t));
	*recv_list = (int32_t *) malloc(std::max<size_t>(r.size(), 1) * sizeof(int32_t));
	if (!*send_list || !*recv_list)
	{
		free(*send_list);
		free(*recv_list);
		*send_list = *recv_list = nullptr;
		set_error("halo_lists: out of memory");
		return 1;
	}
	if (!s.empty())
		memcpy(*send_list, s.data(), s.size() * sizeof(int32_t));
	if (!r.empty())
		memcpy(*recv_list, r.data(), r.size() * sizeof(int32_t));
	return 0;
}

}  // namespace spmv_host
