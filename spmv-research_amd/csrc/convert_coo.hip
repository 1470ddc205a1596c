// The column-blocked layout's entry arrays built ON THE GPU (the host builder in build_coo.hip stays as the checker and
// must give the same bytes: tests/test_gpu_parity.py). The host keeps what is O(rows): the row ranges, the rows to split and the deal of
// chunks to workgroups. Everything O(non-zeros) happens here:
//   1. one thread per entry: its workgroup t (owner of its row's chunk, or of its piece of a split row) and LDS slot ->
//      key = t << 47 | column << 15 | slot; 2. one stable radix sort of (key, entry number) = the host's per-workgroup stable sorts by
//      (column, slot); 3. one wave per workgroup walks its sorted entries and cuts them into groups of up to 64 whose columns lie within SPAN of
//      the group's first (what one wave instruction of coo_blocked_kernel gathers): a first walk counts the groups (-> batches per
//      workgroup, rounded up to the kernel's period of three), a second writes entries, values, padding and the groups' base columns in place.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "launch.hpp"

namespace spmv {

namespace {

constexpr int CB = 256;
constexpr int T_SHIFT = 47;                  // key: t << 47 | column << 15 | slot (slot bits = coo_blocked_slot_bits() = 15)

struct Freed {
	std::vector<void *> ptrs;
	~Freed()
	{
		for (void * p : ptrs)
			(void) hipFree(p);
	}
	template <typename P>
	int get(P ** out, size_t bytes)
	{
		void * p = nullptr;
		HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
		ptrs.push_back(p);
		*out = (P *) p;
		return 0;
	}
	template <typename P>
	int put(P ** out, const std::vector<P> & v)
	{
		if (get(out, v.size() * sizeof(P)))
			return 1;
		if (!v.empty())
			HIP_TRY(hipMemcpy(*out, v.data(), v.size() * sizeof(P), hipMemcpyHostToDevice));
		return 0;
	}
};

__global__ __launch_bounds__(CB) void
blocked_key_kernel(const int * __restrict__ rp, const int * __restrict__ ci, long m, long nnz, const int * __restrict__ range_row,
		const int * __restrict__ range_long, const int * __restrict__ long_row, const int * __restrict__ range_chunk0, const int * __restrict__ chunk_owner,
		const int * __restrict__ chunk_ptr, int NR, int WGS, int CH, int slot_bits, unsigned long long * __restrict__ key, unsigned * __restrict__ ids)
{
	const long e = (long) blockIdx.x * CB + threadIdx.x;
	if (e >= nnz)
		return;
	// the row of entry e: the last i with rp[i] <= e
	long lo = 0, hi = m - 1;
	while (lo < hi)
	{
		const long mid = (lo + hi + 1) / 2;
		if (rp[mid] <= e)
			lo = mid;
		else
			hi = mid - 1;
	}
	const long row = lo;
	// its range: the last r with range_row[r] <= row among those that hold rows
	int a = 0, b = NR - 1;
	while (a < b)
	{
		const int mid = (a + b + 1) / 2;
		if (range_row[mid] <= row)
			a = mid;
		else
			b = mid - 1;
	}
	const int r = a;
	// is it one of the range's split rows?
	int la = range_long[r], lb = range_long[r + 1];
	const int l0 = la;
	while (la < lb)
	{
		const int mid = (la + lb) / 2;
		if (long_row[mid] < row)
			la = mid + 1;
		else
			lb = mid;
	}
	long t, slot;
	if (la < range_long[r + 1] && long_row[la] == row)
	{
		// piece j of the row's entries: [len * j / WGS, len * (j + 1) / WGS)
		const long len = rp[row + 1] - rp[row], q = e - rp[row];
		long j = q * WGS / len;
		while (len * (j + 1) / WGS <= q)
			j++;
		while (len * j / WGS > q)
			j--;
		t = (long) r * WGS + j;
		slot = (long) (chunk_ptr[t + 1] - chunk_ptr[t]) * CH + (la - l0);
	}
	else
	{
		const long c = (row - range_row[r]) / CH;
		const int own = chunk_owner[range_chunk0[r] + c];
		t = (long) r * WGS + (own >> 16);
		slot = (long) (own & 0xffff) * CH + (row - (range_row[r] + c * CH));
	}
	key[e] = (unsigned long long) t << T_SHIFT | (unsigned long long) (unsigned) ci[e] << slot_bits | (unsigned long long) slot;
	ids[e] = (unsigned) e;
}

// first sorted entry of every workgroup
__global__ __launch_bounds__(CB) void
blocked_start_kernel(const unsigned long long * __restrict__ key, long nnz, long NT, long * __restrict__ start)
{
	const long t = (long) blockIdx.x * CB + threadIdx.x;
	if (t > NT)
		return;
	const unsigned long long want = (unsigned long long) t << T_SHIFT;
	long lo = 0, hi = nnz;
	while (lo < hi)
	{
		const long mid = (lo + hi) / 2;
		if (key[mid] < want)
			lo = mid + 1;
		else
			hi = mid;
	}
	start[t] = lo;
}

// one wave per workgroup of the layout. FILL = false: count the groups; true: write them.
template <typename T, bool FILL>
__global__ __launch_bounds__(WAVE) void
blocked_walk_kernel(const unsigned long long * __restrict__ key, const unsigned * __restrict__ ids, const double * __restrict__ va, const long * __restrict__ start,
		const int * __restrict__ wg_rows, const int * __restrict__ batch_ptr, long SPAN, long BATCH, int slot_bits, int SPARE, int * __restrict__ groups_out,
		unsigned * __restrict__ ent, T * __restrict__ val, int * __restrict__ batch_base)
{
	const long t = blockIdx.x;
	const int lane = threadIdx.x;
	const long end = start[t + 1];
	const long NW = 1024 / WAVE, KPL = BATCH / 1024, GPB = NW * KPL;
	const unsigned long long slot_mask = (1ull << slot_bits) - 1;
	long g = 0;
	int lastbase = 0;
	long b0 = 0, nslots = 0;
	unsigned spare0 = 0;
	if (FILL)
	{
		b0 = batch_ptr[t];
		nslots = (long) (batch_ptr[t + 1] - batch_ptr[t]) * GPB;
		spare0 = (unsigned) wg_rows[t];
	}
	for (long k = start[t]; k < end; g++)
	{
		const bool have = k + lane < end;
		const unsigned long long kk = have ? key[k + lane] : 0ull;
		const long col = (long) ((kk >> slot_bits) & 0xffffffffull);
		const long base = __shfl(col, 0, WAVE);
		const unsigned long long ok = __ballot(have && col - base < SPAN);
		const int cnt = ~ok == 0ull ? WAVE : __ffsll((unsigned long long) ~ok) - 1;      // entries up to the first that does not fit
		if (FILL)
		{
			const long bt = g / GPB, u = (g % GPB) / NW, w = g % NW;
			const size_t at = (size_t) ((b0 + bt) * BATCH + u * 1024 + w * WAVE);
			if (lane < cnt)
			{
				ent[at + lane] = (unsigned) ((col - base) << slot_bits) | (unsigned) (kk & slot_mask);
				if (val)
					val[at + lane] = (T) va[ids[k + lane]];
			}
			else
				ent[at + lane] = spare0 + (unsigned) (lane % SPARE);          // padding: column offset 0, a spare LDS slot, value 0
			if (lane == 0)
				batch_base[((b0 + bt) * NW + w) * KPL + u] = (int) base;
			lastbase = (int) base;
		}
		k += cnt;
	}
	if (!FILL)
	{
		if (lane == 0)
			groups_out[t] = (int) g;
		return;
	}
	// what is left of the workgroup's last batches
	for (; g < nslots; g++)
	{
		const long bt = g / GPB, u = (g % GPB) / NW, w = g % NW;
		ent[(size_t) ((b0 + bt) * BATCH + u * 1024 + w * WAVE) + lane] = spare0 + (unsigned) (lane % SPARE);
		if (lane == 0)
			batch_base[((b0 + bt) * NW + w) * KPL + u] = lastbase;
	}
}

}  // namespace

// 0 = built (device arrays owned by the caller; the host's batch_ptr filled), 1 = error.
int
blocked_entries_convert_device(bool f32, bool uniform, long m, long nnz, const int * rp_host, const int * ci_host, const double * va_host, long NR, int WGS, long CH,
		const std::vector<int> & range_row, const std::vector<int> & range_long, const std::vector<int> & long_row, const std::vector<int> & chunk_ptr,
		const std::vector<int> & chunk_row, const std::vector<int> & wg_rows, long SPAN, long BATCH, int slot_bits, int SPARE, long ghost_batches,
		std::vector<int> & batch_ptr, unsigned ** d_ent_out, void ** d_val_out, int ** d_batch_base_out)
{
	const long NT = NR * WGS;
	const long NW = 1024 / WAVE, KPL = BATCH / 1024, GPB = NW * KPL;
	// who owns chunk c of range r, and as which of its chunks
	std::vector<int> range_chunk0((size_t) NR + 1, 0);
	for (long r = 0; r < NR; r++)
		range_chunk0[(size_t) r + 1] = range_chunk0[(size_t) r] + (int) ((range_row[(size_t) r + 1] - range_row[(size_t) r] + CH - 1) / CH);
	std::vector<int> chunk_owner((size_t) std::max(range_chunk0[(size_t) NR], 1), 0);
	for (long t = 0; t < NT; t++)
		for (long k = 0; k < chunk_ptr[(size_t) t + 1] - chunk_ptr[(size_t) t]; k++)
		{
			const long r = t / WGS, c = (chunk_row[(size_t) chunk_ptr[(size_t) t] + (size_t) k] - range_row[(size_t) r]) / CH;
			if (k > 0xffff)
			{
				set_error("column-blocked layout: a workgroup with more than 65 535 chunks");
				return 1;
			}
			chunk_owner[(size_t) (range_chunk0[(size_t) r] + c)] = (int) ((t % WGS) << 16 | k);
		}
	Freed tmp, out_guard;
	int * rp, * ci, * d_range_row, * d_range_long, * d_long_row, * d_chunk0, * d_owner, * d_chunk_ptr, * d_wg_rows, * d_groups, * d_batch_ptr;
	double * va = nullptr;
	unsigned long long * key, * key_sorted;
	unsigned * ids, * ids_sorted;
	long * start;
	if (tmp.get(&rp, (size_t) (m + 1) * 4) || tmp.get(&ci, (size_t) nnz * 4) || tmp.put(&d_range_row, range_row) || tmp.put(&d_range_long, range_long) ||
	    tmp.put(&d_long_row, long_row) || tmp.put(&d_chunk0, range_chunk0) || tmp.put(&d_owner, chunk_owner) || tmp.put(&d_chunk_ptr, chunk_ptr) ||
	    tmp.put(&d_wg_rows, wg_rows) || tmp.get(&d_groups, (size_t) NT * 4) || tmp.get(&d_batch_ptr, (size_t) (NT + 1) * 4) || tmp.get(&key, (size_t) nnz * 8) ||
	    tmp.get(&key_sorted, (size_t) nnz * 8) || tmp.get(&ids, (size_t) nnz * 4) || tmp.get(&ids_sorted, (size_t) nnz * 4) || tmp.get(&start, (size_t) (NT + 1) * 8))
		return 1;
	HIP_TRY(hipMemcpy(rp, rp_host, (size_t) (m + 1) * 4, hipMemcpyHostToDevice));
	if (nnz)
		HIP_TRY(hipMemcpy(ci, ci_host, (size_t) nnz * 4, hipMemcpyHostToDevice));
	if (!uniform)
	{
		if (tmp.get(&va, (size_t) nnz * 8))
			return 1;
		if (nnz)
			HIP_TRY(hipMemcpy(va, va_host, (size_t) nnz * 8, hipMemcpyHostToDevice));
	}
	if (nnz)
	{
		hipLaunchKernelGGL(blocked_key_kernel, dim3((unsigned) ((nnz + CB - 1) / CB)), dim3(CB), 0, 0, rp, ci, m, nnz, d_range_row, d_range_long, d_long_row, d_chunk0,
				d_owner, d_chunk_ptr, (int) NR, WGS, (int) CH, slot_bits, key, ids);
		HIP_TRY(hipGetLastError());
		int t_bits = 1;
		while ((1L << t_bits) < NT)
			t_bits++;
		size_t bytes = 0;
		HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, key, key_sorted, ids, ids_sorted, (int) nnz, 0, T_SHIFT + t_bits, (hipStream_t) 0));
		void * sort_tmp;
		if (tmp.get(&sort_tmp, bytes))
			return 1;
		HIP_TRY(hipcub::DeviceRadixSort::SortPairs(sort_tmp, bytes, key, key_sorted, ids, ids_sorted, (int) nnz, 0, T_SHIFT + t_bits, (hipStream_t) 0));
	}
	hipLaunchKernelGGL(blocked_start_kernel, dim3((unsigned) ((NT + 1 + CB - 1) / CB)), dim3(CB), 0, 0, key_sorted, nnz, NT, start);
	hipLaunchKernelGGL((blocked_walk_kernel<double, false>), dim3((unsigned) NT), dim3(WAVE), 0, 0, key_sorted, ids_sorted, va, start, d_wg_rows, d_batch_ptr, SPAN, BATCH,
			slot_bits, SPARE, d_groups, (unsigned *) nullptr, (double *) nullptr, (int *) nullptr);
	HIP_TRY(hipGetLastError());
	std::vector<int> groups((size_t) NT, 0);
	HIP_TRY(hipMemcpy(groups.data(), d_groups, (size_t) NT * 4, hipMemcpyDeviceToHost));
	batch_ptr.assign((size_t) NT + 1, 0);
	for (long t = 0; t < NT; t++)
	{
		long nbt = (groups[(size_t) t] + GPB - 1) / GPB;
		nbt = (nbt + 2) / 3 * 3;                                      // the kernel's loop is unrolled over three rotating register sets
		batch_ptr[(size_t) t + 1] = batch_ptr[(size_t) t] + (int) nbt;
	}
	HIP_TRY(hipMemcpy(d_batch_ptr, batch_ptr.data(), (size_t) (NT + 1) * 4, hipMemcpyHostToDevice));
	const long NB = batch_ptr[(size_t) NT];
	const size_t lext = (size_t) (NB + ghost_batches) * (size_t) BATCH;
	const size_t vbytes = f32 ? 4 : 8;
	unsigned * ent = nullptr;
	void * val = nullptr;
	int * batch_base = nullptr;
	HIP_TRY(hipMalloc(&ent, lext * 4 + 64));
	out_guard.ptrs.push_back(ent);
	HIP_TRY(hipMemset(ent, 0, lext * 4 + 64));
	HIP_TRY(hipMalloc(&batch_base, (lext / WAVE + STREAM_SLACK) * 4));
	out_guard.ptrs.push_back(batch_base);
	HIP_TRY(hipMemset(batch_base, 0, (lext / WAVE + STREAM_SLACK) * 4));
	if (!uniform)
	{
		HIP_TRY(hipMalloc(&val, lext * vbytes + 64));
		out_guard.ptrs.push_back(val);
		HIP_TRY(hipMemset(val, 0, lext * vbytes + 64));
	}
	if (f32)
		hipLaunchKernelGGL((blocked_walk_kernel<float, true>), dim3((unsigned) NT), dim3(WAVE), 0, 0, key_sorted, ids_sorted, va, start, d_wg_rows, d_batch_ptr, SPAN, BATCH,
				slot_bits, SPARE, d_groups, ent, (float *) val, batch_base);
	else
		hipLaunchKernelGGL((blocked_walk_kernel<double, true>), dim3((unsigned) NT), dim3(WAVE), 0, 0, key_sorted, ids_sorted, va, start, d_wg_rows, d_batch_ptr, SPAN, BATCH,
				slot_bits, SPARE, d_groups, ent, (double *) val, batch_base);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipDeviceSynchronize());
	out_guard.ptrs.clear();
	*d_ent_out = ent;
	*d_val_out = val;
	*d_batch_base_out = batch_base;
	return 0;
}

}  // namespace spmv
