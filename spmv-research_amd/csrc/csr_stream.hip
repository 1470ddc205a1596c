// A CSR assembled in DEVICE memory from pieces of rows, and a handle built from it.
//
// spmv_mi355x_create() wants the whole CSR in host memory at once. A rank of a multi-GPU run that generates (or reads) its rows piece
// by piece should not have to: the pieces are appended to device arrays as they come — each checked on the host while it passes
// (row_ptr monotone from 0, columns in range, as build_input.hip checks a whole matrix) — and the SELL-64-sigma-delta layout is then
// converted on the GPU from the resident CSR (convert_sell.hip), exactly as create() converts an uploaded one. One handle, the
// host never holds more than a piece. bench.py --gpus N builds its per-rank handles this way (python/bench_multi.py).
#include <algorithm>
#include <cstring>
#include <vector>

#include "handle.hpp"

struct spmv_mi355x_csr_stream {
	int device = 0;
	long m = 0, n = 0, capacity = 0;
	long rows_done = 0, nnz_done = 0;
	int * d_rp = nullptr;
	int * d_ci = nullptr;
	double * d_va = nullptr;
};

using namespace spmv;

static void
stream_free(spmv_mi355x_csr_stream * s)
{
	if (!s)
		return;
	(void) hipSetDevice(s->device);
	for (void * p : {(void *) s->d_rp, (void *) s->d_ci, (void *) s->d_va})
		if (p)
			(void) hipFree(p);
	delete s;
}

extern "C" {

int
spmv_mi355x_csr_stream_begin(spmv_mi355x_csr_stream ** out, int device, long m, long n, long nnz_capacity)
{
	if (!out)
	{
		set_error("csr_stream_begin: out is NULL");
		return 1;
	}
	*out = nullptr;
	if (m < 0 || n < 0 || nnz_capacity < 0 || m >= 0x7fffffffL || n >= 0x7fffffffL || nnz_capacity >= 0x7fffffffL)
	{
		set_error("csr_stream_begin: sizes out of the int32 index range (m=%ld n=%ld capacity=%ld)", m, n, nnz_capacity);
		return 1;
	}
	if (device < 0)
		HIP_TRY(hipGetDevice(&device));
	HIP_TRY(hipSetDevice(device));
	spmv_mi355x_csr_stream * s = new spmv_mi355x_csr_stream();
	s->device = device;
	s->m = m;
	s->n = n;
	s->capacity = nnz_capacity;
	if (dev_alloc(&s->d_rp, (size_t) m + 1) || dev_alloc(&s->d_ci, (size_t) std::max<long>(nnz_capacity, 1)) ||
	    dev_alloc(&s->d_va, (size_t) std::max<long>(nnz_capacity, 1)))
	{
		stream_free(s);
		return 1;
	}
	if (hipMemset(s->d_rp, 0, 4) != hipSuccess)
	{
		set_error("csr_stream_begin: %s", hipGetErrorString(hipGetLastError()));
		stream_free(s);
		return 1;
	}
	*out = s;
	return 0;
}

// the next `rows` rows: row_ptr[rows + 1] starting at 0, their col_idx / values
int
spmv_mi355x_csr_stream_append(spmv_mi355x_csr_stream * s, long rows, const int32_t * row_ptr, const int32_t * col_idx, const double * values)
{
	if (!s || rows < 0 || !row_ptr)
	{
		set_error("csr_stream_append: bad argument");
		return 1;
	}
	if (row_ptr[0] != 0)
	{
		set_error("csr_stream_append: a piece's row_ptr must start at 0 (got %d)", row_ptr[0]);
		return 1;
	}
	if (s->rows_done + rows > s->m)
	{
		set_error("csr_stream_append: %ld rows appended to %ld of %ld", rows, s->rows_done, s->m);
		return 1;
	}
	long bad_row = -1;
	#pragma omp parallel for num_threads(spmv::host_threads()) reduction(max : bad_row)
	for (long i = 0; i < rows; i++)
		if (row_ptr[i + 1] < row_ptr[i])
			bad_row = std::max(bad_row, i);
	if (bad_row >= 0)
	{
		set_error("csr_stream_append: row_ptr decreases at row %ld of the piece", bad_row);
		return 1;
	}
	const long pnnz = row_ptr[rows];
	if (s->nnz_done + pnnz > s->capacity)
	{
		set_error("csr_stream_append: %ld non-zeros appended to %ld exceed the capacity %ld", pnnz, s->nnz_done, s->capacity);
		return 1;
	}
	if (pnnz > 0 && (!col_idx || !values))
	{
		set_error("csr_stream_append: NULL col_idx / values");
		return 1;
	}
	long bad = -1;
	#pragma omp parallel for num_threads(spmv::host_threads()) reduction(max : bad)
	for (long j = 0; j < pnnz; j++)
		if (col_idx[j] < 0 || col_idx[j] >= s->n)
			bad = std::max(bad, j);
	if (bad >= 0)
	{
		set_error("column index %d out of range [0,%ld) at entry %ld of the piece", col_idx[bad], s->n, bad);
		return 1;
	}
	HIP_TRY(hipSetDevice(s->device));
	std::vector<int32_t> shifted((size_t) rows);
	#pragma omp parallel for num_threads(spmv::host_threads())
	for (long i = 0; i < rows; i++)
		shifted[(size_t) i] = (int32_t) (row_ptr[i + 1] + s->nnz_done);
	if (rows)
		HIP_TRY(hipMemcpy(s->d_rp + s->rows_done + 1, shifted.data(), (size_t) rows * 4, hipMemcpyHostToDevice));
	if (pnnz)
	{
		HIP_TRY(hipMemcpy(s->d_ci + s->nnz_done, col_idx, (size_t) pnnz * 4, hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(s->d_va + s->nnz_done, values, (size_t) pnnz * 8, hipMemcpyHostToDevice));
	}
	s->rows_done += rows;
	s->nnz_done += pnnz;
	return 0;
}

// Builds the handle and releases the stream (on failure too). Formats: SPMV_MI355X_SELL_C_SIGMA (64-row slices, delta layout).
int
spmv_mi355x_create_from_stream(spmv_mi355x_matrix ** out, spmv_mi355x_csr_stream * s, int format, int precision, const spmv_mi355x_opts * opts_in)
{
	if (!out || !s)
	{
		set_error("create_from_stream: NULL argument");
		stream_free(s);
		return 1;
	}
	*out = nullptr;
	spmv_mi355x_opts o;
	memset(&o, 0, sizeof(o));
	o.device = -1;
	if (opts_in)
	{
		const size_t sz = std::min<size_t>(sizeof(o), (size_t) std::max(opts_in->struct_size, 0));
		if (sz < 8)
		{
			set_error("opts->struct_size not set");
			stream_free(s);
			return 1;
		}
		memcpy(&o, opts_in, sz);
	}
	int rc = 0;
	if (format != SPMV_MI355X_SELL_C_SIGMA)
	{
		set_error("create_from_stream: format %d is not built from a device-resident CSR (SELL-C-sigma only; use spmv_mi355x_create)", format);
		rc = 1;
	}
	else if (precision != SPMV_MI355X_F64 && precision != SPMV_MI355X_F32)
	{
		set_error("unknown precision %d", precision);
		rc = 1;
	}
	else if (s->rows_done != s->m)
	{
		set_error("create_from_stream: %ld of %ld rows were appended", s->rows_done, s->m);
		rc = 1;
	}
	else if (o.row_begin || o.row_end || o.col_filter_mode || o.symmetric_input)
	{
		set_error("create_from_stream: row blocks, column filters and symmetric input belong to spmv_mi355x_create");
		rc = 1;
	}
	if (rc || hipSetDevice(s->device) != hipSuccess)
	{
		stream_free(s);
		return 1;
	}
	spmv_mi355x_matrix * A = new spmv_mi355x_matrix();
	init_handle(A, format, precision, s->device, o, s->m, s->n, s->nnz_done);
	rc = build_sell_delta_resident(A, o, s->d_rp, s->d_ci, s->d_va);
	stream_free(s);
	if (rc)
	{
		free_all(A);
		delete A;
		return 1;
	}
	*out = A;
	return 0;
}

int
spmv_mi355x_csr_stream_discard(spmv_mi355x_csr_stream * s)
{
	stream_free(s);
	return 0;
}

}
