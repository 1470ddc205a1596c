// CSR -> SELL-64-sigma-delta ON THE GPU.
//
// The reference converts on the host (sell_sorted.cpp:112-298, sellcs_format.c:137-200; the driver reports it as
// "time convert to format", bench.cpp:600-603) and so did build_sell_delta() in spmv_mi355x.hip, which stays as the
// checker: this path must produce the SAME bytes (tests/test_gpu_parity.py compares the two layouts).
//   1. row lengths; stable sort of the rows of every sigma-window by length, descending (radix_sort.c:103-122 semantics):
//      one segmented radix sort (hipCUB; radix sort is stable, so equal lengths keep their row order);
//   2. one wave per 64-row slice: width (max length; index groups cover it rounded up to 4 steps) and the narrowest index encoding that holds every
//      (step, lane) delta against the step's minimum column;
//   3. exclusive scans -> value / index offsets of the slices;
//   4. one wave per slice fills values (column-major, narrowed to the handle's precision), step bases and packed deltas.
// The CSR arrays are uploaded once and freed afterwards; nothing but the slice offsets returns to the host (they feed the
// XCD tile map).

#include <hipcub/hipcub.hpp>

#include <vector>

#include "launch.hpp"

namespace spmv {

constexpr int CV_BLOCK = 256;

__global__ __launch_bounds__(CV_BLOCK) void
row_length_kernel(const int * __restrict__ rp, int m, int * __restrict__ len, int * __restrict__ ids)
{
	const int i = blockIdx.x * CV_BLOCK + threadIdx.x;
	if (i < m)
	{
		len[i] = rp[i + 1] - rp[i];
		ids[i] = i;
	}
}

__global__ __launch_bounds__(CV_BLOCK) void
window_offsets_kernel(long m, long sigma, long num_windows, int * __restrict__ off)
{
	const long w = (long) blockIdx.x * CV_BLOCK + threadIdx.x;
	if (w <= num_windows)
		off[w] = (int) (w * sigma < m ? w * sigma : m);
}

__device__ __forceinline__ int
wave_min_i(int v)
{
	for (int o = WAVE / 2; o > 0; o >>= 1)
		v = min(v, __shfl_xor(v, o, WAVE));
	return v;
}

__device__ __forceinline__ int
wave_max_i(int v)
{
	for (int o = WAVE / 2; o > 0; o >>= 1)
		v = max(v, __shfl_xor(v, o, WAVE));
	return v;
}

__device__ __forceinline__ long
group_bytes(int md)
{
	return (md == 0 || md == 3) ? 16 : md == 1 ? 272 : md == 2 ? 528 : 1024;
}

// Mode 5 (kernels_sell.hip): every lane of a slice of equally long rows gets the offset off = its first column - the reference lane's
// first column; a lane is REGULAR when column_k = column_k[reference] + off at every step, an EXCEPTION when the difference fits a
// signed byte at every step, HARD otherwise. The reference lanes 0..3 are tried in turn; the first one with at least 48 regular lanes,
// at least one exception and no hard lane is taken. Returns the 64-bit mask of the exception lanes (every lane the same value) and,
// through `hard`, whether some lane needs more than 8 bits.
constexpr int SELL5_MIN_REGULAR = 48;

__device__ __forceinline__ unsigned long long
sell5_exceptions(const int * __restrict__ ci, int start, int maxlen, int r, bool & hard)
{
	int regular = 1, fits = 1, off = 0;
	for (int k = 0; k < maxlen; k++)
	{
		const int c = ci[start + k];
		const int rel = c - __shfl(c, r, WAVE);
		if (k == 0)
			off = rel;
		else
		{
			regular = regular && rel == off;
			fits = fits && rel - off >= -128 && rel - off <= 127;
		}
	}
	hard = __ballot(!fits) != 0ull;
	return ~__ballot(regular);
}

__device__ __forceinline__ void
sell5_reference(const int * __restrict__ ci, int start, int maxlen, int lane, int & ref, int & nex)
{
	(void) lane;
	ref = -1;
	nex = 0;
	for (int r = 0; r < 4 && ref < 0; r++)
	{
		bool hard;
		const int e = __popcll(sell5_exceptions(ci, start, maxlen, r, hard));
		if (WAVE - e >= SELL5_MIN_REGULAR && e > 0 && !hard)
		{
			ref = r;
			nex = e;
		}
	}
}

// pass 1: one wave per slice -> mode, number of value slots, number of index bytes
__global__ __launch_bounds__(CV_BLOCK) void
slice_shape_kernel(const int * __restrict__ rp, const int * __restrict__ ci, const int * __restrict__ row_of_sorted, long m, long n,
		long num_slices, unsigned char * __restrict__ mode, int64_t * __restrict__ val_count, int64_t * __restrict__ idx_count, int modes_off)
{
	const long sl = ((long) blockIdx.x * CV_BLOCK + threadIdx.x) / WAVE;
	const int lane = threadIdx.x % WAVE;
	if (sl >= num_slices)
		return;
	const long i = sl * WAVE + lane;
	int start = 0, len = 0;
	if (i < m)
	{
		const int o = row_of_sorted[i];
		start = rp[o];
		len = rp[o + 1] - start;
	}
	const int maxlen = wave_max_i(len);
	const int width = (maxlen + 3) / 4 * 4;
	int maxdelta = 0;
	// step-invariant lane offsets (modes 0 and 3): a full slice of equally long rows whose step-k columns are
	// c_k[lane 0] + off[lane] with the same off at every step; off = lane is the affine case (mode 0)
	const bool uniform = (sl + 1) * WAVE <= m && n >= WAVE && wave_min_i(len) == maxlen;
	int rowoff = (uniform && maxlen > 0) ? 1 : 0;
	int affine = (uniform && maxlen == 0) ? 1 : 0;
	int off = 0;
	for (int k = 0; k < width; k++)
	{
		const bool ok = k < len;
		const int c = ok ? ci[start + k] : 0;
		const int lo = wave_min_i(ok ? c : 0x7fffffff);
		const int hi = wave_max_i(ok ? c : -1);
		if (hi >= 0)
			maxdelta = max(maxdelta, hi - lo);
		if (rowoff && k < maxlen)
		{
			const int rel = c - __shfl(c, 0, WAVE);
			if (k == 0)
			{
				off = rel;
				affine = wave_min_i(rel == lane ? 1 : 0);
			}
			else
				rowoff = wave_min_i(rel == off ? 1 : 0);
		}
	}
	if (!rowoff)
		affine = (uniform && maxlen == 0) ? 1 : 0;
	// lane offsets with exceptions (mode 5): the slice is not of one pattern, but at least 48 of its 64 rows are — against one of the
	// first four lanes as the reference (a reference that is itself out of line agrees with nobody)
	int ref = -1, nex = 0;
	if (uniform && maxlen > 0 && !rowoff && !(modes_off & 4))
		sell5_reference(ci, start, maxlen, lane, ref, nex);
	if (lane == 0)
	{
		// modes_off (sensitivity experiments, sell_modes_off()): bit 0 forbids the affine mode, bit 1 the per-slice lane offsets,
		// bit 2 the lane offsets with exceptions
		const int md = (affine && !(modes_off & 1)) ? 0 : (rowoff && !(modes_off & 2)) ? 3 : ref >= 0 ? 5 : maxdelta < 256 ? 1 : maxdelta < 65536 ? 2 : 4;
		mode[sl] = (unsigned char) (md == 5 ? (5 | ref << 3) : md);
		val_count[sl] = (int64_t) maxlen * WAVE;             // values: exact width; index groups: rounded up to 4 steps
		idx_count[sl] = md == 5 ? (int64_t) (4 * WAVE + 16) + (int64_t) (width / 4) * (16 + (nex + 3) / 4 * 16)
		                        : (int64_t) (md == 3 ? 4 * WAVE : 0) + (int64_t) (width / 4) * group_bytes(md);
	}
}

// pass 2: one wave per slice fills its values, bases and deltas (layout: kernels_sell.hip)
template <typename T>
__global__ __launch_bounds__(CV_BLOCK) void
slice_fill_kernel(const int * __restrict__ rp, const int * __restrict__ ci, const double * __restrict__ va,
		const int * __restrict__ row_of_sorted, long m, long num_slices, const unsigned char * __restrict__ mode,
		const int64_t * __restrict__ val_ptr, const int64_t * __restrict__ idx_ptr, T * __restrict__ val,
		unsigned char * __restrict__ idx, int64_t * __restrict__ desc)
{
	const long sl = ((long) blockIdx.x * CV_BLOCK + threadIdx.x) / WAVE;
	const int lane = threadIdx.x % WAVE;
	if (sl > num_slices)
		return;
	if (sl == num_slices)                              // terminator entry
	{
		if (lane == 0)
		{
			desc[2 * sl] = val_ptr[sl];
			desc[2 * sl + 1] = idx_ptr[sl] | 4;
		}
		return;
	}
	const int64_t vb = val_ptr[sl];
	const int maxlen = (int) ((val_ptr[sl + 1] - vb) / WAVE);
	const int width = (maxlen + 3) / 4 * 4;
	const int md = mode[sl] & 7, ref = mode[sl] >> 3;
	unsigned char * ib = idx + idx_ptr[sl];
	if (lane == 0)
	{
		desc[2 * sl] = vb;
		desc[2 * sl + 1] = idx_ptr[sl] | md;
	}
	const long i = sl * WAVE + lane;
	int start = 0, len = 0;
	if (i < m)
	{
		const int o = row_of_sorted[i];
		start = rp[o];
		len = rp[o + 1] - start;
	}
	long gbytes = group_bytes(md);
	int pad_base = 0;
	bool ex = false;
	int ex_rank = 0, off5 = 0, nex5 = 0;
	if (md == 5)
	{
		// header: the 64 lane offsets relative to the reference lane's first column, the exception mask, padding
		bool hard;
		const unsigned long long mask = sell5_exceptions(ci, start, maxlen, ref, hard);
		ex = (mask >> lane) & 1ull;
		ex_rank = __popcll(mask & ((1ull << lane) - 1ull));
		const int c_first = ci[start];
		off5 = c_first - __shfl(c_first, ref, WAVE);
		reinterpret_cast<int *>(ib)[lane] = off5;
		pad_base = -wave_min_i(off5);                      // all-padding steps: base + off must stay a valid column
		if (lane == 0)
		{
			reinterpret_cast<unsigned long long *>(ib + 4 * WAVE)[0] = mask;
			reinterpret_cast<unsigned long long *>(ib + 4 * WAVE)[1] = 0ull;
		}
		ib += 4 * WAVE + 16;
		nex5 = __popcll(mask);
		gbytes = 16 + (nex5 + 3) / 4 * 16;
	}
	if (md == 3)
	{
		// header: the 64 lane offsets relative to lane 0's column, then the groups of 4 bases
		const int c_first = ci[start];                     // every row of a mode-3 slice has at least one entry
		const int off = c_first - __shfl(c_first, 0, WAVE);
		reinterpret_cast<int *>(ib)[lane] = off;
		pad_base = -wave_min_i(off);                       // all-padding steps: base + off must stay a valid column
		ib += 4 * WAVE;
	}
	for (int g = 0; g < width / 4; g++)
	{
		unsigned char * gp = ib + g * gbytes;
		unsigned d[4];
		int cc[4];
		#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const int k = g * 4 + u;
			const bool ok = k < len;
			int c = ok ? ci[start + k] : 0x7fffffff;
			int base = wave_min_i(c);
			if (base == 0x7fffffff)
				base = 0;                              // a step that is padding for every lane
			if (md == 3)                               // base + off[lane]: lane 0's column on a real step
				base = ok ? __shfl(c, 0, WAVE) : pad_base;
			if (md == 5)                               // ... the reference lane's (rows of a mode-5 slice are equally long: ok is uniform)
				base = ok ? __shfl(c, ref, WAVE) : pad_base;
			if (!ok)
				c = md == 5 ? base + off5 : base;      // padding: value 0 times a column some lane really uses (mode 5: the pattern's, correction 0)
			cc[u] = md == 5 ? c - (base + off5) : 0;   // mode 5: the correction of this step (0 for a regular lane)
			if (k < maxlen)                            // steps past the slice's longest row exist in the index groups only
				val[vb + sell_pair_pos(k, maxlen, lane)] = ok ? (T) va[start + k] : (T) 0;
			if (md != 4)
			{
				if (lane == 0)
					reinterpret_cast<int *>(gp)[u] = base;
				d[u] = (unsigned) (c - base);               // unused by the affine mode (column = base + lane)
			}
			else
				reinterpret_cast<int *>(gp)[u * WAVE + lane] = c;
		}
		if (md == 5)
		{
			// the exception lanes' four corrections of the group, one signed byte each; the tail of the 16-byte-padded group is zeroed
			if (ex)
				reinterpret_cast<unsigned *>(gp + 16)[ex_rank] = ((unsigned) cc[0] & 255u) | ((unsigned) cc[1] & 255u) << 8 | ((unsigned) cc[2] & 255u) << 16 | ((unsigned) cc[3] & 255u) << 24;
			if (lane >= nex5 && lane < (int) ((gbytes - 16) / 4))             // dword slots of the group behind the last exception's
				reinterpret_cast<unsigned *>(gp + 16)[lane] = 0u;
		}
		if (md == 1)
			reinterpret_cast<unsigned *>(gp + 16)[lane] = d[0] | d[1] << 8 | d[2] << 16 | d[3] << 24;
		else if (md == 2)
		{
			uint2 pk;
			pk.x = d[0] | d[1] << 16;
			pk.y = d[2] | d[3] << 16;
			reinterpret_cast<uint2 *>(gp + 16)[lane] = pk;
		}
	}
}

struct Scratch {
	std::vector<void *> ptrs;
	~Scratch()
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
};

// The conversion proper, on a CSR that already lives in device memory (rp[m+1] from 0, ci[nnz], va[nnz] as fp64).
// Outputs (device, owned by the caller on success): row_of_sorted[m], desc[2*(slices+1)], idx[idx_bytes+1024], val[nnz_ext
// + STREAM_SLACK] of the handle's precision. Host outputs: val_ptr (slices+1, for the tile map), mode counts, sizes.
int
sell_delta_convert_resident(bool f32, long m, long n_cols, long nnz, long sigma, const int * rp, const int * ci,
		const double * va, int ** d_row_of_sorted_out, int64_t ** d_desc_out, unsigned char ** d_idx_out, void ** d_val_out,
		std::vector<int64_t> & val_ptr_host, long mode_counts[4], int64_t * nnz_ext_out, int64_t * idx_bytes_out)
{
	(void) nnz;
	const long num_slices = (m + WAVE - 1) / WAVE;
	const long num_windows = (m + sigma - 1) / sigma;
	Scratch tmp;
	int * len, * len_sorted, * ids, * win_off;
	if (tmp.get(&len, (size_t) m * 4) || tmp.get(&len_sorted, (size_t) m * 4) || tmp.get(&ids, (size_t) m * 4) ||
	    tmp.get(&win_off, (size_t) (num_windows + 1) * 4))
		return 1;
	int * row_of_sorted = nullptr;
	HIP_TRY(hipMalloc(&row_of_sorted, (size_t) std::max<long>(m, 1) * 4 + STREAM_SLACK * 4));
	Scratch out_guard;                                 // frees the outputs if anything below fails
	out_guard.ptrs.push_back(row_of_sorted);
	HIP_TRY(hipMemset(row_of_sorted, 0, (size_t) std::max<long>(m, 1) * 4 + STREAM_SLACK * 4));

	// 1. sigma-window sort by length, descending, stable
	if (m > 0)
	{
		hipLaunchKernelGGL(row_length_kernel, dim3((unsigned) ((m + CV_BLOCK - 1) / CV_BLOCK)), dim3(CV_BLOCK), 0, 0, rp, (int) m, len, ids);
		hipLaunchKernelGGL(window_offsets_kernel, dim3((unsigned) ((num_windows + 1 + CV_BLOCK - 1) / CV_BLOCK)), dim3(CV_BLOCK), 0, 0,
				m, sigma, num_windows, win_off);
		HIP_TRY(hipGetLastError());
		size_t bytes = 0;
		HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, bytes, len, len_sorted, ids, row_of_sorted, (int) m,
				(int) num_windows, win_off, win_off + 1, 0, 32, (hipStream_t) 0));
		void * sort_tmp;
		if (tmp.get(&sort_tmp, bytes))
			return 1;
		HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(sort_tmp, bytes, len, len_sorted, ids, row_of_sorted, (int) m,
				(int) num_windows, win_off, win_off + 1, 0, 32, (hipStream_t) 0));
	}

	// 2. slice shapes   3. offsets
	unsigned char * mode;
	int64_t * val_count, * idx_count, * val_ptr, * idx_ptr;
	if (tmp.get(&mode, (size_t) num_slices + 1) || tmp.get(&val_count, (size_t) (num_slices + 1) * 8) ||
	    tmp.get(&idx_count, (size_t) (num_slices + 1) * 8) || tmp.get(&val_ptr, (size_t) (num_slices + 1) * 8) ||
	    tmp.get(&idx_ptr, (size_t) (num_slices + 1) * 8))
		return 1;
	HIP_TRY(hipMemset(val_count, 0, (size_t) (num_slices + 1) * 8));
	HIP_TRY(hipMemset(idx_count, 0, (size_t) (num_slices + 1) * 8));
	const unsigned slice_grid = (unsigned) (((num_slices + 1) * WAVE + CV_BLOCK - 1) / CV_BLOCK);
	if (num_slices > 0)
	{
		hipLaunchKernelGGL(slice_shape_kernel, dim3(slice_grid), dim3(CV_BLOCK), 0, 0, rp, ci, row_of_sorted, m, n_cols, num_slices, mode,
				val_count, idx_count, sell_modes_off());
		HIP_TRY(hipGetLastError());
	}
	{
		size_t bytes = 0;
		HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, val_count, val_ptr, (int) (num_slices + 1), (hipStream_t) 0));
		void * scan_tmp;
		if (tmp.get(&scan_tmp, bytes))
			return 1;
		HIP_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, bytes, val_count, val_ptr, (int) (num_slices + 1), (hipStream_t) 0));
		HIP_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, bytes, idx_count, idx_ptr, (int) (num_slices + 1), (hipStream_t) 0));
	}
	val_ptr_host.assign((size_t) num_slices + 1, 0);
	HIP_TRY(hipMemcpy(val_ptr_host.data(), val_ptr, (size_t) (num_slices + 1) * 8, hipMemcpyDeviceToHost));
	int64_t idx_bytes = 0;
	HIP_TRY(hipMemcpy(&idx_bytes, idx_ptr + num_slices, 8, hipMemcpyDeviceToHost));
	const int64_t nnz_ext = val_ptr_host[num_slices];
	{
		std::vector<unsigned char> mode_host((size_t) std::max<long>(num_slices, 1));
		if (num_slices)
			HIP_TRY(hipMemcpy(mode_host.data(), mode, (size_t) num_slices, hipMemcpyDeviceToHost));
		mode_counts[0] = mode_counts[1] = mode_counts[2] = mode_counts[3] = 0;
		for (long sl = 0; sl < num_slices; sl++)
			mode_counts[((mode_host[sl] & 7) == 0 || (mode_host[sl] & 7) == 3 || (mode_host[sl] & 7) == 5) ? 3 : (mode_host[sl] & 7) == 1 ? 0 : (mode_host[sl] & 7) == 2 ? 1 : 2]++;
	}

	// 4. fill
	const size_t vbytes = f32 ? 4 : 8;
	void * val = nullptr;
	unsigned char * idx = nullptr;
	int64_t * desc = nullptr;
	const size_t idx_alloc = (size_t) std::max<int64_t>(idx_bytes, 16) + 1024;
	HIP_TRY(hipMalloc(&val, ((size_t) nnz_ext + STREAM_SLACK) * vbytes));
	out_guard.ptrs.push_back(val);
	HIP_TRY(hipMalloc(&idx, idx_alloc));
	out_guard.ptrs.push_back(idx);
	HIP_TRY(hipMalloc(&desc, 2 * ((size_t) num_slices + 1) * 8));
	out_guard.ptrs.push_back(desc);
	HIP_TRY(hipMemset((char *) val + (size_t) nnz_ext * vbytes, 0, STREAM_SLACK * vbytes));
	HIP_TRY(hipMemset(idx + (idx_alloc - 1040), 0, 1040));     // the tail the kernels may read past the last group
	if (idx_bytes < 16)
		HIP_TRY(hipMemset(idx, 0, idx_alloc));
	if (f32)
		hipLaunchKernelGGL((slice_fill_kernel<float>), dim3(slice_grid), dim3(CV_BLOCK), 0, 0, rp, ci, va, row_of_sorted, m, num_slices,
				mode, val_ptr, idx_ptr, (float *) val, idx, desc);
	else
		hipLaunchKernelGGL((slice_fill_kernel<double>), dim3(slice_grid), dim3(CV_BLOCK), 0, 0, rp, ci, va, row_of_sorted, m, num_slices,
				mode, val_ptr, idx_ptr, (double *) val, idx, desc);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipDeviceSynchronize());
	out_guard.ptrs.clear();                            // success: ownership moves to the caller
	*d_row_of_sorted_out = row_of_sorted;
	*d_desc_out = desc;
	*d_idx_out = idx;
	*d_val_out = val;
	*nnz_ext_out = nnz_ext;
	*idx_bytes_out = idx_bytes;
	return 0;
}

// The same from HOST arrays: upload, convert, drop the uploaded copy.
int
sell_delta_convert_device(bool f32, long m, long n_cols, long nnz, long sigma, const int * rp_host, const int * ci_host,
		const double * va_host, int ** d_row_of_sorted_out, int64_t ** d_desc_out, unsigned char ** d_idx_out, void ** d_val_out,
		std::vector<int64_t> & val_ptr_host, long mode_counts[4], int64_t * nnz_ext_out, int64_t * idx_bytes_out)
{
	Scratch up;
	int * rp, * ci;
	double * va;
	if (up.get(&rp, (size_t) (m + 1) * 4) || up.get(&ci, (size_t) nnz * 4) || up.get(&va, (size_t) nnz * 8))
		return 1;
	if (rp_host[0] != 0)
	{
		set_error("sell_delta_convert_device: row_ptr must start at 0");
		return 1;
	}
	HIP_TRY(hipMemcpy(rp, rp_host, (size_t) (m + 1) * 4, hipMemcpyHostToDevice));
	if (nnz)
	{
		HIP_TRY(hipMemcpy(ci, ci_host, (size_t) nnz * 4, hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(va, va_host, (size_t) nnz * 8, hipMemcpyHostToDevice));
	}
	return sell_delta_convert_resident(f32, m, n_cols, nnz, sigma, rp, ci, va, d_row_of_sorted_out, d_desc_out, d_idx_out, d_val_out, val_ptr_host,
			mode_counts, nnz_ext_out, idx_bytes_out);
}

// ---------------------------------------------------------------- the LDS-window layout (build_sell.hip: build_sell_window) on the GPU
// Same bytes as the host builder (tests/test_gpu_parity.py compares the arrays of the two): 1. window (lowest column, width) of every
// group of NS slices: one workgroup per group; 2. stable sort of the group's rows by length, descending (the segmented radix sort of
// the delta conversion, sigma = 64 * NS); 3. slice widths rounded up to whole groups of 4 steps, exclusive scan; 4. one wave per slice
// writes the values column-major and the 16-bit window-relative indices, 4 steps of a lane side by side.

__global__ __launch_bounds__(CV_BLOCK) void
window_group_kernel(const int * __restrict__ rp, const int * __restrict__ ci, long m, long sigma, int NS, long num_slices, int sym, int bytes_per_col,
		long lds_budget_bytes, int * __restrict__ grp, int * __restrict__ flags)
{
	__shared__ int s_lo[CV_BLOCK / WAVE], s_hi[CV_BLOCK / WAVE];
	const long g = blockIdx.x;
	const long r0 = g * sigma, r1 = r0 + sigma < m ? r0 + sigma : m;
	int lo = 0x7fffffff, hi = -1;
	for (long j = (long) rp[r0] + threadIdx.x; j < rp[r1]; j += CV_BLOCK)
	{
		lo = min(lo, ci[j]);
		hi = max(hi, ci[j]);
	}
	lo = wave_min_i(lo);
	hi = wave_max_i(hi);
	if (threadIdx.x % WAVE == 0)
	{
		s_lo[threadIdx.x / WAVE] = lo;
		s_hi[threadIdx.x / WAVE] = hi;
	}
	__syncthreads();
	if (threadIdx.x != 0)
		return;
	for (int w = 1; w < CV_BLOCK / WAVE; w++)
	{
		lo = min(lo, s_lo[w]);
		hi = max(hi, s_hi[w]);
	}
	long llo = lo, lhi = hi;
	if (sym)
	{
		llo = llo < r0 ? llo : r0;
		lhi = lhi > r1 - 1 ? lhi : r1 - 1;
	}
	if (lhi < 0)
		llo = 0;
	const long w = lhi < 0 ? 1 : lhi - llo + 1;
	if (w > 65534 || (w + 1) * (long) bytes_per_col > lds_budget_bytes)
		atomicAdd(&flags[0], 1);
	const int wi = (int) (w < 0x7fffffffL ? w : 0x7fffffffL);
	grp[4 * g] = (int) llo;
	grp[4 * g + 1] = wi;
	grp[4 * g + 2] = (int) (g * NS);
	grp[4 * g + 3] = (int) (NS < num_slices - g * NS ? NS : num_slices - g * NS);
	atomicMax(&flags[1], wi);
}

__global__ __launch_bounds__(CV_BLOCK) void
window_width_kernel(const int * __restrict__ rp, const int * __restrict__ row_of_sorted, long num_slices, int64_t * __restrict__ count)
{
	const long sl = (long) blockIdx.x * CV_BLOCK + threadIdx.x;
	if (sl > num_slices)
		return;
	int64_t c = 0;
	if (sl < num_slices)
	{
		const int o = row_of_sorted[sl * WAVE];                   // the slice's longest row is its first
		c = (int64_t) ((rp[o + 1] - rp[o] + 3) / 4 * 4) * WAVE;
	}
	count[sl] = c;
}

template <typename T>
__global__ __launch_bounds__(CV_BLOCK) void
window_fill_kernel(const int * __restrict__ rp, const int * __restrict__ ci, const double * __restrict__ va, const int * __restrict__ row_of_sorted,
		long m, long num_slices, int NS, int sym, const int * __restrict__ grp, const int64_t * __restrict__ ptr, T * __restrict__ val,
		unsigned short * __restrict__ idx, int64_t * __restrict__ desc)
{
	const long sl = ((long) blockIdx.x * CV_BLOCK + threadIdx.x) / WAVE;
	const int r = threadIdx.x % WAVE;
	if (sl > num_slices)
		return;
	if (r == 0)
		desc[2 * sl] = desc[2 * sl + 1] = ptr[sl];
	if (sl == num_slices)
		return;
	const int lo = grp[4 * (sl / NS)], gw = grp[4 * (sl / NS) + 1];
	const int64_t b = ptr[sl];
	const long width = (ptr[sl + 1] - b) / WAVE;
	const long i = sl * WAVE + r;
	long js = 0, len = 0;
	if (i < m)
	{
		const int o = row_of_sorted[i];
		js = rp[o];
		len = rp[o + 1] - js;
	}
	const unsigned short pad = (len > 0 && !sym) ? (unsigned short) (ci[js + len - 1] - lo) : (unsigned short) gw;
	for (long k0 = 0; k0 < width; k0 += 4)
	{
		unsigned short q[4];
		#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const long k = k0 + u;
			val[b + sellw_val_pos(k, r, sizeof(T) == 4)] = k < len ? (T) va[js + k] : (T) 0;
			q[u] = k < len ? (unsigned short) (ci[js + k] - lo) : pad;
		}
		*(uint2 *) (idx + b + k0 * WAVE + r * 4) = make_uint2((unsigned) q[0] | (unsigned) q[1] << 16, (unsigned) q[2] | (unsigned) q[3] << 16);
	}
}

// From HOST arrays: upload, convert, drop the uploaded copy. Returns 0 = built (device outputs owned by the caller), 1 = error, 2 = a
// group's window is too wide (nothing is returned). Host outputs: the slice descriptors (for the tile map and the sizes), the widest window.
int
sell_window_convert_device(bool f32, long m, long nnz, int NS, bool sym, long lds_budget_bytes, const int * rp_host, const int * ci_host,
		const double * va_host, int ** d_grp_out, int ** d_row_of_sorted_out, int64_t ** d_desc_out, unsigned short ** d_idx_out, void ** d_val_out,
		std::vector<int64_t> & desc_host, int * max_w_out)
{
	const long num_slices = (m + WAVE - 1) / WAVE;
	const long num_groups = (num_slices + NS - 1) / NS;
	const long sigma = (long) NS * WAVE;
	const size_t vbytes = f32 ? 4 : 8;
	if (rp_host[0] != 0)
	{
		set_error("sell_window_convert_device: row_ptr must start at 0");
		return 1;
	}
	Scratch tmp, out_guard;
	int * rp, * ci, * flags;
	double * va;
	if (tmp.get(&rp, (size_t) (m + 1) * 4) || tmp.get(&ci, (size_t) nnz * 4) || tmp.get(&va, (size_t) nnz * 8) || tmp.get(&flags, 8))
		return 1;
	HIP_TRY(hipMemcpy(rp, rp_host, (size_t) (m + 1) * 4, hipMemcpyHostToDevice));
	if (nnz)
	{
		HIP_TRY(hipMemcpy(ci, ci_host, (size_t) nnz * 4, hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(va, va_host, (size_t) nnz * 8, hipMemcpyHostToDevice));
	}
	HIP_TRY(hipMemset(flags, 0, 8));
	// 1. windows
	int * grp = nullptr;
	HIP_TRY(hipMalloc(&grp, ((size_t) std::max<long>(num_groups, 1) * 4 + STREAM_SLACK) * 4));
	out_guard.ptrs.push_back(grp);
	HIP_TRY(hipMemset(grp, 0, ((size_t) std::max<long>(num_groups, 1) * 4 + STREAM_SLACK) * 4));
	if (num_groups > 0)
	{
		hipLaunchKernelGGL(window_group_kernel, dim3((unsigned) num_groups), dim3(CV_BLOCK), 0, 0, rp, ci, m, sigma, NS, num_slices, sym ? 1 : 0,
				(int) (vbytes + (sym ? 8 : 0)), lds_budget_bytes, grp, flags);
		HIP_TRY(hipGetLastError());
	}
	int flags_host[2] = {0, 0};
	HIP_TRY(hipMemcpy(flags_host, flags, 8, hipMemcpyDeviceToHost));
	if (flags_host[0])
		return 2;
	*max_w_out = flags_host[1];
	// 2. rows of a group by length, descending, stable
	int * len, * len_sorted, * ids, * win_off;
	if (tmp.get(&len, (size_t) m * 4) || tmp.get(&len_sorted, (size_t) m * 4) || tmp.get(&ids, (size_t) m * 4) || tmp.get(&win_off, (size_t) (num_groups + 1) * 4))
		return 1;
	int * row_of_sorted = nullptr;
	HIP_TRY(hipMalloc(&row_of_sorted, ((size_t) std::max<long>(m, 1) + STREAM_SLACK) * 4));
	out_guard.ptrs.push_back(row_of_sorted);
	HIP_TRY(hipMemset(row_of_sorted, 0, ((size_t) std::max<long>(m, 1) + STREAM_SLACK) * 4));
	if (m > 0)
	{
		hipLaunchKernelGGL(row_length_kernel, dim3((unsigned) ((m + CV_BLOCK - 1) / CV_BLOCK)), dim3(CV_BLOCK), 0, 0, rp, (int) m, len, ids);
		hipLaunchKernelGGL(window_offsets_kernel, dim3((unsigned) ((num_groups + 1 + CV_BLOCK - 1) / CV_BLOCK)), dim3(CV_BLOCK), 0, 0, m, sigma, num_groups, win_off);
		HIP_TRY(hipGetLastError());
		size_t bytes = 0;
		HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, bytes, len, len_sorted, ids, row_of_sorted, (int) m, (int) num_groups, win_off,
				win_off + 1, 0, 32, (hipStream_t) 0));
		void * sort_tmp;
		if (tmp.get(&sort_tmp, bytes))
			return 1;
		HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(sort_tmp, bytes, len, len_sorted, ids, row_of_sorted, (int) m, (int) num_groups, win_off,
				win_off + 1, 0, 32, (hipStream_t) 0));
	}
	// 3. slice offsets
	int64_t * count, * ptr;
	if (tmp.get(&count, (size_t) (num_slices + 1) * 8) || tmp.get(&ptr, (size_t) (num_slices + 1) * 8))
		return 1;
	hipLaunchKernelGGL(window_width_kernel, dim3((unsigned) ((num_slices + 1 + CV_BLOCK - 1) / CV_BLOCK)), dim3(CV_BLOCK), 0, 0, rp, row_of_sorted, num_slices, count);
	HIP_TRY(hipGetLastError());
	{
		size_t bytes = 0;
		HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, count, ptr, (int) (num_slices + 1), (hipStream_t) 0));
		void * scan_tmp;
		if (tmp.get(&scan_tmp, bytes))
			return 1;
		HIP_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, bytes, count, ptr, (int) (num_slices + 1), (hipStream_t) 0));
	}
	int64_t nnz_ext = 0;
	HIP_TRY(hipMemcpy(&nnz_ext, ptr + num_slices, 8, hipMemcpyDeviceToHost));
	// 4. fill
	void * val = nullptr;
	unsigned short * idx = nullptr;
	int64_t * desc = nullptr;
	const size_t idx_bytes = (size_t) std::max<int64_t>(nnz_ext, 1) * 2;
	HIP_TRY(hipMalloc(&val, ((size_t) nnz_ext + STREAM_SLACK) * vbytes));
	out_guard.ptrs.push_back(val);
	HIP_TRY(hipMalloc(&idx, idx_bytes + 1024));
	out_guard.ptrs.push_back(idx);
	HIP_TRY(hipMalloc(&desc, 2 * ((size_t) num_slices + 1) * 8));
	out_guard.ptrs.push_back(desc);
	HIP_TRY(hipMemset((char *) val + (size_t) nnz_ext * vbytes, 0, STREAM_SLACK * vbytes));
	HIP_TRY(hipMemset((char *) idx + (size_t) nnz_ext * 2, 0, idx_bytes + 1024 - (size_t) nnz_ext * 2));
	const unsigned slice_grid = (unsigned) (((num_slices + 1) * WAVE + CV_BLOCK - 1) / CV_BLOCK);
	if (f32)
		hipLaunchKernelGGL((window_fill_kernel<float>), dim3(slice_grid), dim3(CV_BLOCK), 0, 0, rp, ci, va, row_of_sorted, m, num_slices, NS, sym ? 1 : 0, grp, ptr,
				(float *) val, idx, desc);
	else
		hipLaunchKernelGGL((window_fill_kernel<double>), dim3(slice_grid), dim3(CV_BLOCK), 0, 0, rp, ci, va, row_of_sorted, m, num_slices, NS, sym ? 1 : 0, grp, ptr,
				(double *) val, idx, desc);
	HIP_TRY(hipGetLastError());
	desc_host.assign(2 * ((size_t) num_slices + 1), 0);
	HIP_TRY(hipMemcpy(desc_host.data(), desc, desc_host.size() * 8, hipMemcpyDeviceToHost));
	HIP_TRY(hipDeviceSynchronize());
	out_guard.ptrs.clear();                            // success: ownership moves to the caller
	*d_grp_out = grp;
	*d_row_of_sorted_out = row_of_sorted;
	*d_desc_out = desc;
	*d_idx_out = idx;
	*d_val_out = val;
	return 0;
}

// ---------------------------------------------------------------- the plain column-major layout (build_sell.hip: build_sell; C = 16 / 32 / 64 / 256) on the GPU
// Same sort; a slice's width is its first row's length (the rows of a sigma window are sorted by length and sigma is a multiple of C),
// rounded up to the lanes per row; one thread per (slice, row) writes its row column-major, padding = value 0 times the row's last column.

__global__ __launch_bounds__(CV_BLOCK) void
plain_width_kernel(const int * __restrict__ rp, const int * __restrict__ row_of_sorted, long m, long num_slices, int C, int TPR, int64_t * __restrict__ count)
{
	const long sl = (long) blockIdx.x * CV_BLOCK + threadIdx.x;
	if (sl > num_slices)
		return;
	int64_t c = 0;
	if (sl < num_slices && sl * C < m)
	{
		const int o = row_of_sorted[sl * C];
		const long width = ((long) (rp[o + 1] - rp[o]) + TPR - 1) / TPR * TPR;
		c = width * C;
	}
	count[sl] = c;
}

template <typename T>
__global__ __launch_bounds__(CV_BLOCK) void
plain_fill_kernel(const int * __restrict__ rp, const int * __restrict__ ci, const double * __restrict__ va, const int * __restrict__ row_of_sorted, long m,
		long num_slices, int C, const int64_t * __restrict__ ptr, int * __restrict__ col, T * __restrict__ val)
{
	const long t = (long) blockIdx.x * CV_BLOCK + threadIdx.x;
	const long sl = t / C;
	const int r = (int) (t % C);
	if (sl >= num_slices)
		return;
	const int64_t base = ptr[sl];
	const long width = (ptr[sl + 1] - base) / C;
	const long i = sl * C + r;
	long js = 0, len = 0;
	if (i < m)
	{
		const int o = row_of_sorted[i];
		js = rp[o];
		len = rp[o + 1] - js;
	}
	const int pad_col = len > 0 ? ci[js + len - 1] : 0;
	for (long k = 0; k < width; k++)
	{
		const int64_t p = base + k * C + r;
		col[p] = k < len ? ci[js + k] : pad_col;
		val[p] = k < len ? (T) va[js + k] : (T) 0;
	}
}

// From HOST arrays. Device outputs owned by the caller on success; slice_ptr_host for the tile map and the sizes.
int
sell_plain_convert_device(bool f32, long m, long nnz, int C, int TPR, long sigma, const int * rp_host, const int * ci_host, const double * va_host,
		int64_t ** d_slice_ptr_out, int ** d_col_out, void ** d_val_out, int ** d_row_of_sorted_out, std::vector<int64_t> & slice_ptr_host)
{
	const long num_slices = (m + C - 1) / C;
	const long num_windows = (m + sigma - 1) / sigma;
	const size_t vbytes = f32 ? 4 : 8;
	if (rp_host[0] != 0)
	{
		set_error("sell_plain_convert_device: row_ptr must start at 0");
		return 1;
	}
	Scratch tmp, out_guard;
	int * rp, * ci, * len, * len_sorted, * ids, * win_off;
	double * va;
	int64_t * count;
	if (tmp.get(&rp, (size_t) (m + 1) * 4) || tmp.get(&ci, (size_t) nnz * 4) || tmp.get(&va, (size_t) nnz * 8) || tmp.get(&len, (size_t) m * 4) ||
	    tmp.get(&len_sorted, (size_t) m * 4) || tmp.get(&ids, (size_t) m * 4) || tmp.get(&win_off, (size_t) (num_windows + 1) * 4) ||
	    tmp.get(&count, (size_t) (num_slices + 1) * 8))
		return 1;
	HIP_TRY(hipMemcpy(rp, rp_host, (size_t) (m + 1) * 4, hipMemcpyHostToDevice));
	if (nnz)
	{
		HIP_TRY(hipMemcpy(ci, ci_host, (size_t) nnz * 4, hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(va, va_host, (size_t) nnz * 8, hipMemcpyHostToDevice));
	}
	int * row_of_sorted = nullptr;
	HIP_TRY(hipMalloc(&row_of_sorted, ((size_t) std::max<long>(m, 1) + STREAM_SLACK) * 4));
	out_guard.ptrs.push_back(row_of_sorted);
	HIP_TRY(hipMemset(row_of_sorted, 0, ((size_t) std::max<long>(m, 1) + STREAM_SLACK) * 4));
	if (m > 0)
	{
		hipLaunchKernelGGL(row_length_kernel, dim3((unsigned) ((m + CV_BLOCK - 1) / CV_BLOCK)), dim3(CV_BLOCK), 0, 0, rp, (int) m, len, ids);
		hipLaunchKernelGGL(window_offsets_kernel, dim3((unsigned) ((num_windows + 1 + CV_BLOCK - 1) / CV_BLOCK)), dim3(CV_BLOCK), 0, 0, m, sigma, num_windows, win_off);
		HIP_TRY(hipGetLastError());
		size_t bytes = 0;
		HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, bytes, len, len_sorted, ids, row_of_sorted, (int) m, (int) num_windows, win_off,
				win_off + 1, 0, 32, (hipStream_t) 0));
		void * sort_tmp;
		if (tmp.get(&sort_tmp, bytes))
			return 1;
		HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(sort_tmp, bytes, len, len_sorted, ids, row_of_sorted, (int) m, (int) num_windows, win_off,
				win_off + 1, 0, 32, (hipStream_t) 0));
	}
	int64_t * ptr = nullptr;
	HIP_TRY(hipMalloc(&ptr, ((size_t) num_slices + 1) * 8));
	out_guard.ptrs.push_back(ptr);
	hipLaunchKernelGGL(plain_width_kernel, dim3((unsigned) ((num_slices + 1 + CV_BLOCK - 1) / CV_BLOCK)), dim3(CV_BLOCK), 0, 0, rp, row_of_sorted, m, num_slices, C, TPR, count);
	HIP_TRY(hipGetLastError());
	{
		size_t bytes = 0;
		HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, count, ptr, (int) (num_slices + 1), (hipStream_t) 0));
		void * scan_tmp;
		if (tmp.get(&scan_tmp, bytes))
			return 1;
		HIP_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, bytes, count, ptr, (int) (num_slices + 1), (hipStream_t) 0));
	}
	slice_ptr_host.assign((size_t) num_slices + 1, 0);
	HIP_TRY(hipMemcpy(slice_ptr_host.data(), ptr, ((size_t) num_slices + 1) * 8, hipMemcpyDeviceToHost));
	const int64_t nnz_ext = slice_ptr_host[(size_t) num_slices];
	int * col = nullptr;
	void * val = nullptr;
	HIP_TRY(hipMalloc(&col, ((size_t) nnz_ext + STREAM_SLACK) * 4));
	out_guard.ptrs.push_back(col);
	HIP_TRY(hipMalloc(&val, ((size_t) nnz_ext + STREAM_SLACK) * vbytes));
	out_guard.ptrs.push_back(val);
	HIP_TRY(hipMemset(col + nnz_ext, 0, STREAM_SLACK * 4));
	HIP_TRY(hipMemset((char *) val + (size_t) nnz_ext * vbytes, 0, STREAM_SLACK * vbytes));
	if (num_slices > 0)
	{
		const unsigned grid = (unsigned) ((num_slices * C + CV_BLOCK - 1) / CV_BLOCK);
		if (f32)
			hipLaunchKernelGGL((plain_fill_kernel<float>), dim3(grid), dim3(CV_BLOCK), 0, 0, rp, ci, va, row_of_sorted, m, num_slices, C, ptr, col, (float *) val);
		else
			hipLaunchKernelGGL((plain_fill_kernel<double>), dim3(grid), dim3(CV_BLOCK), 0, 0, rp, ci, va, row_of_sorted, m, num_slices, C, ptr, col, (double *) val);
		HIP_TRY(hipGetLastError());
	}
	HIP_TRY(hipDeviceSynchronize());
	out_guard.ptrs.clear();
	*d_slice_ptr_out = ptr;
	*d_col_out = col;
	*d_val_out = val;
	*d_row_of_sorted_out = row_of_sorted;
	return 0;
}

}  // namespace spmv
