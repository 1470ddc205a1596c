// Device-memory helpers and launch-policy choices shared by the format builders.
#include "handle.hpp"

namespace spmv {

int
dev_alloc_bytes(void ** p, size_t bytes)
{
	*p = nullptr;
	if (bytes == 0)
		bytes = 8;
	HIP_TRY(hipMalloc(p, bytes));
	return 0;
}

void
free_all(spmv_mi355x_matrix * A)
{
	(void) vector_free(A->d_x);            // the handle's vectors may be slices of the device's vector pools (placement.hip)
	(void) vector_free(A->d_y);
	void * ptrs[] = {A->d_row_ptr, A->d_col, A->d_val, A->d_coords, A->d_carry_row, A->d_carry_val, A->d_slice_ptr,
	                 A->d_row_of_sorted, A->d_rowind, A->d_sell_desc, A->d_sell_idx, A->d_win_row, A->d_win_lo,
	                 A->d_win_w, A->d_col16, A->d_coob_wg_rows, A->d_coob_range_row, A->d_coob_chunk_ptr, A->d_coob_chunk_row, A->d_coob_batch_ptr, A->d_coob_batch_base, A->d_coob_ent, A->d_coob_range_long, A->d_coob_long_row, A->d_coob_carry, A->d_sellw_grp};
	for (void * p : ptrs)
		if (p)
			(void) hipFree(p);
	if (A->stream)
		(void) hipStreamDestroy(A->stream);
}

// narrow fp64 reference values to the handle's precision (csr.cpp:72 `a[i] = values[i]`) and upload
int
upload_values(spmv_mi355x_matrix * A, const double * v, size_t count, void ** d_out)
{
	// STREAM_SLACK spare entries: the LDS-DMA row-block copy reads whole 1 KiB chunks (kernels_csr_stream.hip)
	if (dev_alloc_bytes(d_out, (count + STREAM_SLACK) * A->vbytes))
		return 1;
	HIP_TRY(hipMemset((char *) *d_out + count * A->vbytes, 0, STREAM_SLACK * A->vbytes));
	if (count == 0)
		return 0;
	if (!A->f32)
	{
		HIP_TRY(hipMemcpy(*d_out, v, count * sizeof(double), hipMemcpyHostToDevice));
		return 0;
	}
	// chunked narrowing keeps the host staging buffer small for 10^9-entry matrices
	const size_t CH = (size_t) 1 << 26;
	std::vector<float> tmp(std::min(CH, count));
	for (size_t off = 0; off < count; off += CH)
	{
		size_t len = std::min(CH, count - off);
		#pragma omp parallel for num_threads(spmv::host_threads())
		for (long i = 0; i < (long) len; i++)
			tmp[i] = (float) v[off + i];
		HIP_TRY(hipMemcpy((char *) *d_out + off * sizeof(float), tmp.data(), len * sizeof(float), hipMemcpyHostToDevice));
	}
	return 0;
}

int
upload_ints(const int * src, size_t count, int ** d_out)
{
	if (dev_alloc(d_out, count + STREAM_SLACK))
		return 1;
	HIP_TRY(hipMemset(*d_out + count, 0, STREAM_SLACK * sizeof(int)));
	if (count)
		HIP_TRY(hipMemcpy(*d_out, src, count * sizeof(int), hipMemcpyHostToDevice));
	return 0;
}

int
pick_lanes_per_row(double mean)
{
	// measured on the five BASELINE.json twins (profiles/sweep_r01.md): 8..16 lanes win from 5.6 to 64 nnz/row — a
	// wider group only adds idle lanes and butterfly steps, a narrower one serialises the row
	if (mean <= 4) return 4;
	if (mean <= 12) return 8;
	if (mean <= 128) return 16;
	if (mean <= 512) return 32;
	return 64;
}

// auto tile order: contiguous work-balanced ranges keep each XCD's L2 on one window of x (best for small and skewed
// matrices); for many-tile matrices chunks of 64 tiles dealt round-robin balance row-count-bound kernels better
// (nlpkkt240 twin: csr_vector +26 %, csr_stream +12 %, SELL +2 %; pwtk/soc-LiveJournal1 twins prefer the ranges)
int
resolve_remap(int requested, long ntiles)
{
	if (requested >= 0)
		return requested;
	return ntiles >= 8192 ? 2 : 1;
}

int
upload_bytes(const void * src, size_t bytes, size_t slack_bytes, void ** d_out)
{
	if (dev_alloc_bytes(d_out, bytes + slack_bytes))
		return 1;
	if (slack_bytes)
		HIP_TRY(hipMemset((char *) *d_out + bytes, 0, slack_bytes));
	if (bytes)
		HIP_TRY(hipMemcpy(*d_out, src, bytes, hipMemcpyHostToDevice));
	return 0;
}

bool
values_uniform(const spmv_mi355x_matrix * A, const double * va, long nnz, double * v0_out)
{
	if (nnz <= 0)
		return false;
	const bool f32 = A->f32;
	const double v0 = f32 ? (double) (float) va[0] : va[0];
	long differs = 0;
	#pragma omp parallel for num_threads(spmv::host_threads()) reduction(+ : differs)
	for (long j = 0; j < nnz; j++)
		differs += (f32 ? (double) (float) va[j] : va[j]) != v0;
	*v0_out = v0;
	return differs == 0 && v0 == v0;
}

}  // namespace spmv
