// C ABI of the MI355X SpMV engine (include/spmv_mi355x.h): handle management and the spmv entry points. Host code only;
// the kernels live in kernels_*.hip, the format constructors (= the reference's csr_to_format) in build_*.hip. There is
// deliberately NO CPU compute path here: every y comes from a HIP kernel.

#include "handle.hpp"

namespace spmv {

static thread_local char g_err[1024] = "";

void
set_error(const char * fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

// what every constructor of a handle starts from: identity, sizes, launch policy
void
init_handle(spmv_mi355x_matrix * A, int format, int precision, int device, const spmv_mi355x_opts & o, long m, long n, long nnz)
{
	A->format = format;
	A->precision = precision;
	A->f32 = (precision == SPMV_MI355X_F32);
	A->vbytes = A->f32 ? 4 : 8;
	A->device = device;
	A->placement_level = (o.placement == 1 || o.placement == 3 || o.placement == 4) ? o.placement : 0;      // 0 and 2: off
	A->placement_budget_gib = o.placement_budget_gib > 0 ? o.placement_budget_gib : 0;
	A->convert_on_device = o.convert_on != 2 && !getenv("SPMV_MI355X_HOST_CONVERT");      // layouts with a GPU builder: SELL delta, SELL window, column-blocked
	A->n = n;
	A->m = m;
	A->nnz = nnz;
	A->csr_mem_footprint = (double) nnz * (A->vbytes + 4) + (double) (m + 1) * 4;
	A->remap = (o.xcd_remap == 2) ? 0 : (o.xcd_remap == 3) ? 2 : (o.xcd_remap == 1) ? 1 : -1;   // -1 = auto, resolved per kernel
	const double stream_bytes = (double) nnz * (A->vbytes + 4);
	A->cfg.nt = (o.nontemporal == 1) ? 1 : (o.nontemporal == 2) ? 0 : (stream_bytes > 192.0 * 1024 * 1024 ? 1 : 0);
	A->cfg.beta = 0;
}

// the handle's own (zeroed) input vector. No stream of its own here: callers of the device-pointer entry points bring theirs, and with
// several processes on one GPU every extra hardware queue costs (4 gloo ranks on one MI355X: halo exchange 2.4 -> 40 ms per step)
int
ensure_x(spmv_mi355x_matrix * A)
{
	HIP_TRY(hipSetDevice(A->device));
	if (!A->d_x)
	{
		if (dev_alloc_bytes(&A->d_x, (size_t) std::max<long>(A->n, 1) * A->vbytes))
			return 1;
		HIP_TRY(hipMemset(A->d_x, 0, (size_t) std::max<long>(A->n, 1) * A->vbytes));
		HIP_TRY(hipDeviceSynchronize());
	}
	return 0;
}

}  // namespace spmv

using namespace spmv;

// ---------------------------------------------------------------------------------------------------- C ABI

extern "C" {

const char *
spmv_mi355x_last_error(void)
{
	return g_err;
}

int
spmv_mi355x_device_count(int * count_out)
{
	int c = 0;
	hipError_t e = hipGetDeviceCount(&c);
	if (e != hipSuccess)
	{
		(void) hipGetLastError();
		c = 0;
	}
	*count_out = c;
	return 0;
}

int
spmv_mi355x_device_info(int device, char * name_out, long name_n, int * compute_units_out, long * hbm_bytes_out)
{
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (name_out && name_n > 0)
		snprintf(name_out, name_n, "%s (%s)", prop.name, prop.gcnArchName);
	if (compute_units_out)
		*compute_units_out = prop.multiProcessorCount;
	if (hbm_bytes_out)
		*hbm_bytes_out = (long) prop.totalGlobalMem;
	return 0;
}

int
spmv_mi355x_create(spmv_mi355x_matrix ** out, int format, int precision, long m, long n, long nnz,
		const int32_t * row_ptr, const int32_t * col_idx, const double * values, const spmv_mi355x_opts * opts_in)
{
	*out = nullptr;
	spmv_mi355x_opts o;
	memset(&o, 0, sizeof(o));
	o.device = -1;
	if (opts_in)
	{
		size_t sz = std::min<size_t>(sizeof(o), (size_t) std::max(opts_in->struct_size, 0));
		if (sz < 8)
		{
			set_error("opts->struct_size not set");
			return 1;
		}
		memcpy(&o, opts_in, sz);
	}
	if (format < 0 || format >= SPMV_MI355X_NUM_FORMATS)
	{
		set_error("unknown format %d", format);
		return 1;
	}
	if (precision != SPMV_MI355X_F64 && precision != SPMV_MI355X_F32)
	{
		set_error("unknown precision %d", precision);
		return 1;
	}
	if (m < 0 || n < 0 || nnz < 0 || m >= 0x7fffffffL || n >= 0x7fffffffL || nnz >= 0x7fffffffL || m + nnz >= 0x7fffffffL)
	{
		set_error("sizes out of the int32 index range (m=%ld n=%ld nnz=%ld)", m, n, nnz);
		return 1;
	}
	if (!row_ptr || (nnz > 0 && (!col_idx || !values)))
	{
		set_error("NULL input array");
		return 1;
	}
	int ndev = 0;
	spmv_mi355x_device_count(&ndev);
	if (ndev < 1)
	{
		set_error("no HIP device available: this engine has no CPU fallback");
		return 1;
	}
	int device = o.device;
	if (device < 0)
		HIP_TRY(hipGetDevice(&device));
	if (device >= ndev)
	{
		set_error("device %d out of range (%d devices)", device, ndev);
		return 1;
	}
	HIP_TRY(hipSetDevice(device));
	// ---- symmetric storage in + SELL-C-sigma: the stored triangle itself in the LDS-window layout, when the matrix is banded enough for
	// a slice group's window of rows and columns to fit LDS (half the matrix stream; kernels_sell_window.hip). Anything else — and every
	// malformed input, for its error message — goes through the expansion below.
	if (o.symmetric_input && format == SPMV_MI355X_SELL_C_SIGMA && o.sell_window != 2 && o.sell_delta != 1 && (o.sell_c == 0 || o.sell_c == 64) &&
	    o.sell_sigma == 0 && m == n && !o.row_begin && !o.row_end && !o.col_filter_mode && row_ptr[0] == 0 && row_ptr[m] == nnz && nnz >= (1L << 16))
	{
		bool ok = true;
		long expanded = 0;
		#pragma omp parallel for num_threads(spmv::host_threads()) schedule(static, 4096) reduction(&& : ok) reduction(+ : expanded)
		for (long i = 0; i < m; i++)
		{
			ok = ok && row_ptr[i + 1] >= row_ptr[i] && row_ptr[i] >= 0 && row_ptr[i + 1] <= nnz;
			for (long j = row_ptr[i]; j < row_ptr[i + 1] && ok; j++)
			{
				ok = ok && col_idx[j] >= 0 && col_idx[j] < n;
				expanded += col_idx[j] == i ? 1 : 2;
			}
		}
		// Taken on its own only when the EXPANDED matrix would not stay in the 256 MiB Infinity Cache from launch to launch: measured on the
		// cant / pwtk twins (49 / 96 MB, cache-resident), symmetric storage halves the footprint and the bytes moved but takes 14.6 against
		// 9.2 us and 30.9 against 17.3 us — one LDS atomic per entry and a second launch (the y clear) cost more than a stream the caches
		// serve anyway. opts.sell_window = 1 asks for it regardless.
		const bool wanted = o.sell_window == 1 || (double) expanded * ((precision == SPMV_MI355X_F32 ? 4 : 8) + 4) > 256.0 * 1024 * 1024;
		if (ok && wanted && expanded < 0x7fffffffL)
		{
			spmv_mi355x_matrix * S = new spmv_mi355x_matrix();
			init_handle(S, format, precision, device, o, m, n, expanded);
			const int took = build_sell_symmetric(S, o, row_ptr, col_idx, values);
			if (took == 0)
			{
				*out = S;
				return 0;
			}
			free_all(S);
			delete S;
			if (took == 1)
				return 1;
		}
	}
	// ---- input stage: symmetric expansion, row block / column filter, validation (build_input.hip)
	LocalCsr in;
	if (prepare_local_csr(o, m, n, nnz, row_ptr, col_idx, values, in))
		return 1;

	spmv_mi355x_matrix * A = new spmv_mi355x_matrix();
	init_handle(A, format, precision, device, o, in.m, n, in.nnz);

	// ---- the format's constructor (= csr_to_format of the reference's backends)
	int rc;
	switch (format)
	{
		case SPMV_MI355X_SELL_C_SIGMA:
			rc = build_sell_family(A, o, in.rp, in.ci, in.va);
			break;
		case SPMV_MI355X_COO:
			rc = build_coo_family(A, o, in.rp, in.ci, in.va);
			break;
		default:
			rc = build_csr_family(A, o, in.rp, in.ci, in.va);
	}
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
spmv_mi355x_destroy(spmv_mi355x_matrix * A)
{
	if (!A)
		return 0;
	(void) hipSetDevice(A->device);
	free_all(A);
	delete A;
	return 0;
}

const char * spmv_mi355x_format_name(const spmv_mi355x_matrix * A) { return A->format_name; }
double spmv_mi355x_mem_footprint(const spmv_mi355x_matrix * A) { return A->mem_footprint; }
double spmv_mi355x_csr_mem_footprint(const spmv_mi355x_matrix * A) { return A->csr_mem_footprint; }
long spmv_mi355x_rows(const spmv_mi355x_matrix * A) { return A->m; }
long spmv_mi355x_cols(const spmv_mi355x_matrix * A) { return A->n; }
long spmv_mi355x_nnz(const spmv_mi355x_matrix * A) { return A->nnz; }
int spmv_mi355x_precision(const spmv_mi355x_matrix * A) { return A->precision; }
int spmv_mi355x_device(const spmv_mi355x_matrix * A) { return A->device; }

int
spmv_mi355x_spmv_device_async(spmv_mi355x_matrix * A, const void * x, void * y, int beta, void * hip_stream)
{
	hipStream_t st = (hipStream_t) hip_stream;
	// the handle's arrays live on A->device: launching with another device current would hand that device foreign pointers
	int cur = -1;
	HIP_TRY(hipGetDevice(&cur));
	if (cur != A->device)
		HIP_TRY(hipSetDevice(A->device));
	LaunchCfg cfg = A->cfg;
	cfg.beta = beta ? 1 : 0;
	long grid = 0;
	int rc = 1;
	if (A->nnz == 0)
	{
		// a handle without entries (the remote-column half of a row block whose columns are all local): y += 0 is nothing at all,
		// y = 0 is a fill — no kernel walks empty slices and no launch re-reads and re-writes y
		if (!beta && A->m > 0)
			HIP_TRY(hipMemsetAsync(y, 0, (size_t) A->m * A->vbytes, st));
		A->last_grid = 0;
		return 0;
	}
	switch (A->format)
	{
		case SPMV_MI355X_CSR_SCALAR:
			rc = launch_csr_scalar(A->f32, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid);
			break;
		case SPMV_MI355X_CSR_VECTOR:
			rc = launch_csr_vector(A->f32, A->lanes_per_row, A->rows_per_group, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid);
			break;
		case SPMV_MI355X_CSR_STREAM:
			rc = (A->stream_mode == 4)
			     ? launch_csr_window(A->f32, A->lanes_per_row, A->d_row_ptr, A->d_col16 ? (const void *) A->d_col16 : (const void *) A->d_col,
					A->d_col16 ? 1 : 0, A->d_val, x, y, A->d_win_row, A->d_win_lo, A->d_win_w,
					A->win_lds_bytes, cfg, st, &grid)
			     : (A->stream_mode == 3)
			     ? launch_csr_stream_d(A->f32, A->lanes_per_row, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid)
			     : (A->stream_mode == 2)
			     ? launch_csr_stream_t(A->f32, A->lanes_per_row, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid)
			     : launch_csr_stream(A->f32, A->lanes_per_row, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, cfg, st, &grid);
			break;
		case SPMV_MI355X_CSR_MERGE:
			if (A->coob_ranges > 0)
			{
				rc = launch_coo_blocked(A->f32, A->d_coob_wg_rows, A->d_coob_range_row, A->d_coob_chunk_ptr, A->d_coob_chunk_row, A->d_coob_batch_ptr, A->d_coob_batch_base, A->d_coob_range_long,
						A->d_coob_long_row, A->coob_num_long, A->d_coob_ent, A->d_val, x, y, A->d_coob_carry, A->coob_ranges,
						A->coob_chunk_rows, A->coob_lds, cfg, st, &grid);
				break;
			}
			rc = launch_merge(A->f32, A->merge_ipt, A->d_row_ptr, A->d_col, A->d_val, x, y, (int) A->m, (int) A->nnz,
					A->merge_num_tiles, A->d_coords, A->d_carry_row, A->d_carry_val, cfg, st, &grid);
			break;
		case SPMV_MI355X_SELL_C_SIGMA:
			rc = A->sell_sym
			     ? launch_sell_window_sym(A->f32, A->sell_split, A->sellw_ns, A->d_sellw_grp, A->d_sell_desc, (const unsigned short *) A->d_sell_idx, A->d_val,
					A->d_row_of_sorted, x, y, (int) A->m, A->sellw_lds, cfg, st, &grid)
			     : A->sell_window
			     ? launch_sell_window(A->f32, A->sell_split, A->sellw_ns, A->d_sellw_grp, A->d_sell_desc, (const unsigned short *) A->d_sell_idx, A->d_val,
					A->d_row_of_sorted, x, y, (int) A->m, A->sellw_lds, cfg, st, &grid)
			     : A->sell_delta
			     ? launch_sell_delta(A->f32, A->sell_split, A->d_sell_desc, A->d_sell_idx, A->d_val, A->d_row_of_sorted, x, y, (int) A->m,
					(int) A->sell_slices, cfg, st, &grid)
			     : launch_sell(A->f32, A->sell_c, A->d_slice_ptr, A->d_col, A->d_val, A->d_row_of_sorted, x, y, (int) A->m,
					(int) A->sell_slices, cfg, st, &grid);
			break;
		case SPMV_MI355X_COO:
			if (A->coob_ranges > 0)
			{
				rc = launch_coo_blocked(A->f32, A->d_coob_wg_rows, A->d_coob_range_row, A->d_coob_chunk_ptr, A->d_coob_chunk_row, A->d_coob_batch_ptr, A->d_coob_batch_base, A->d_coob_range_long,
						A->d_coob_long_row, A->coob_num_long, A->d_coob_ent, A->d_val, x, y, A->d_coob_carry, A->coob_ranges,
						A->coob_chunk_rows, A->coob_lds, cfg, st, &grid);
				break;
			}
			rc = launch_coo(A->f32, A->coo_k, A->d_rowind, A->d_col, A->d_val, x, y, (int) A->m, A->nnz, A->coo_num_waves,
					A->d_carry_row, A->d_carry_val, cfg, st, &grid);
			break;
		default:
			set_error("bad handle");
	}
	A->last_grid = grid;
	return rc;
}

int
spmv_mi355x_copy_device_async(void * dst, const void * src, long bytes, void * hip_stream)
{
	if (bytes < 0 || (bytes > 0 && (!dst || !src)))
	{
		set_error("copy_device_async: bad argument");
		return 1;
	}
	if (bytes)
		HIP_TRY(hipMemcpyAsync(dst, src, (size_t) bytes, hipMemcpyDeviceToDevice, (hipStream_t) hip_stream));
	return 0;
}

int
spmv_mi355x_time_device(spmv_mi355x_matrix * A, const void * x, void * y, int iters, void * hip_stream, double * ms_out)
{
	hipStream_t st = (hipStream_t) hip_stream;
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipEventRecord(e0, st));
	for (int i = 0; i < iters; i++)
		if (spmv_mi355x_spmv_device_async(A, x, y, 0, hip_stream))
			return 1;
	HIP_TRY(hipEventRecord(e1, st));
	HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	(void) hipEventDestroy(e0);
	(void) hipEventDestroy(e1);
	*ms_out = iters > 0 ? (double) ms / iters : 0;
	return 0;
}

int
spmv_mi355x_kernel_info(const spmv_mi355x_matrix * A, char * name_out, long name_n, long * grid_out, int * block_out)
{
	if (name_out && name_n > 0)
		snprintf(name_out, name_n, "%s", A->kernel_name);
	if (grid_out)
		*grid_out = A->last_grid;
	if (block_out)
		*block_out = A->kernel_block;
	return 0;
}

static int
ensure_xy(spmv_mi355x_matrix * A)
{
	if (ensure_x(A))
		return 1;
	if (!A->stream)
		HIP_TRY(hipStreamCreate(&A->stream));          // a BLOCKING stream: ordered against the null-stream copies / fills of create() and of placement
	if (!A->d_y && tune_placement(A))       // placement.hip: y (zero-filled), plain or from the device's vector pools; x re-homed with it
		return 1;
	return 0;
}

void * spmv_mi355x_x_device(spmv_mi355x_matrix * A) { return ensure_xy(A) ? nullptr : A->d_x; }
void * spmv_mi355x_y_device(spmv_mi355x_matrix * A) { return ensure_xy(A) ? nullptr : A->d_y; }

int
spmv_mi355x_set_always_copy(spmv_mi355x_matrix * A, int on)
{
	A->always_copy = on != 0;
	return 0;
}

int
spmv_mi355x_upload_x(spmv_mi355x_matrix * A, const void * x_host)
{
	if (ensure_xy(A))
		return 1;
	HIP_TRY(hipDeviceSynchronize());        // a host transfer of the handle's vectors: after whatever the caller queued on its own streams
	HIP_TRY(hipMemcpyAsync(A->d_x, x_host, (size_t) A->n * A->vbytes, hipMemcpyHostToDevice, A->stream));
	HIP_TRY(hipStreamSynchronize(A->stream));
	A->cached_x_host = x_host;
	return 0;
}

int
spmv_mi355x_upload_y(spmv_mi355x_matrix * A, const void * y_host)
{
	if (ensure_xy(A))
		return 1;
	HIP_TRY(hipDeviceSynchronize());        // a host transfer of the handle's vectors: after whatever the caller queued on its own streams
	HIP_TRY(hipMemcpyAsync(A->d_y, y_host, (size_t) A->m * A->vbytes, hipMemcpyHostToDevice, A->stream));
	HIP_TRY(hipStreamSynchronize(A->stream));
	return 0;
}

int
spmv_mi355x_download_y(spmv_mi355x_matrix * A, void * y_host)
{
	if (ensure_xy(A))
		return 1;
	HIP_TRY(hipDeviceSynchronize());        // a host transfer of the handle's vectors: after whatever the caller queued on its own streams
	HIP_TRY(hipMemcpyAsync(y_host, A->d_y, (size_t) A->m * A->vbytes, hipMemcpyDeviceToHost, A->stream));
	HIP_TRY(hipStreamSynchronize(A->stream));
	A->y_downloaded = true;
	return 0;
}

// Matrix_Format::spmv with the reference GPU backends' caching convention (csr_rocm_vector.cpp:224-257).
int
spmv_mi355x_spmv(spmv_mi355x_matrix * A, const void * x_host, void * y_host)
{
	if (ensure_xy(A))
		return 1;
	if (A->always_copy || A->cached_x_host != x_host)
	{
		// a new x (new host pointer) must give a new y: the reference backends cache both for the driver's loop over ONE x
		// (csr_rocm_vector.cpp:224-257); the same pointer with changed contents needs set_always_copy
		if (spmv_mi355x_upload_x(A, x_host))
			return 1;
		A->y_downloaded = false;
	}
	if (spmv_mi355x_spmv_device_async(A, A->d_x, A->d_y, 0, A->stream))
		return 1;
	HIP_TRY(hipStreamSynchronize(A->stream));
	if (A->always_copy || !A->y_downloaded)
		if (spmv_mi355x_download_y(A, y_host))
			return 1;
	return 0;
}

// One of the handle's stored arrays as it lies in device memory, copied to a malloc'ed host buffer (free with spmv_mi355x_free): the
// LDS-window SELL layout and the column-blocked layout, whose host and GPU builders must give the same bytes (tests/test_gpu_parity.py).
int
spmv_mi355x_stored_array(const spmv_mi355x_matrix * A, const char * name, void ** out, size_t * bytes_out)
{
	if (!A || !name || !out || !bytes_out)
	{
		set_error("stored_array: NULL argument");
		return 1;
	}
	*out = nullptr;
	*bytes_out = 0;
	struct Arr { const char * name; const void * p; size_t bytes; };
	std::vector<Arr> arrs;
	if (A->format == SPMV_MI355X_SELL_C_SIGMA && A->sell_window)
		arrs = {{"val", A->d_val, (size_t) A->sell_nnz_ext * A->vbytes}, {"idx", A->d_sell_idx, (size_t) A->sell_idx_bytes},
		        {"desc", A->d_sell_desc, 2 * ((size_t) A->sell_slices + 1) * 8}, {"row_of_sorted", A->d_row_of_sorted, (size_t) A->m * 4},
		        {"groups", A->d_sellw_grp, (size_t) A->sellw_groups * 16}};
	else if (A->format == SPMV_MI355X_SELL_C_SIGMA && !A->sell_delta)
		arrs = {{"val", A->d_val, (size_t) A->sell_nnz_ext * A->vbytes}, {"col", A->d_col, (size_t) A->sell_nnz_ext * 4},
		        {"slice_ptr", A->d_slice_ptr, ((size_t) A->sell_slices + 1) * 8}, {"row_of_sorted", A->d_row_of_sorted, (size_t) A->m * 4}};
	else if (A->d_coob_ent)
	{
		const size_t NT = (size_t) A->coob_ranges * spmv::coo_blocked_wgs_per_range(), BATCH = (size_t) spmv::coo_blocked_batch_entries(A->cfg.unit != 0);
		const size_t stored = ((size_t) A->coob_batches + 2) * BATCH;
		arrs = {{"entries", A->d_coob_ent, stored * 4}, {"val", A->d_val, A->cfg.unit ? 0 : stored * A->vbytes},
		        {"batch_base", A->d_coob_batch_base, stored / 64 * 4}, {"batch_ptr", A->d_coob_batch_ptr, (NT + 1) * 4},
		        {"chunk_ptr", A->d_coob_chunk_ptr, (NT + 1) * 4}, {"chunk_row", A->d_coob_chunk_row, (size_t) A->coob_chunks * 4},
		        {"wg_rows", A->d_coob_wg_rows, NT * 4}, {"range_row", A->d_coob_range_row, ((size_t) A->coob_ranges + 1) * 4},
		        {"range_long", A->d_coob_range_long, ((size_t) A->coob_ranges + 1) * 4}, {"long_row", A->d_coob_long_row, (size_t) A->coob_num_long * 4}};
	}
	else
	{
		set_error("stored_array: only for the plain and the LDS-window SELL layouts and the column-blocked layout (this handle: %s)", A->format_name);
		return 1;
	}
	for (const Arr & a : arrs)
		if (!strcmp(a.name, name))
		{
			HIP_TRY(hipSetDevice(A->device));
			void * host = malloc(std::max<size_t>(a.bytes, 1));
			if (!host)
			{
				set_error("stored_array: out of host memory (%zu bytes)", a.bytes);
				return 1;
			}
			if (a.bytes && hipMemcpy(host, a.p, a.bytes, hipMemcpyDeviceToHost) != hipSuccess)
			{
				free(host);
				set_error("stored_array: copy failed: %s", hipGetErrorString(hipGetLastError()));
				return 1;
			}
			*out = host;
			*bytes_out = a.bytes;
			return 0;
		}
	set_error("stored_array: %s has no array '%s'", A->format_name, name);
	return 1;
}

int
spmv_mi355x_sell_layout(const spmv_mi355x_matrix * A, long * C_out, long * sigma_out, long * num_slices_out,
		long * nnz_ext_out, int64_t ** slice_ptr_out, int32_t ** col_out, double ** val_out, int32_t ** row_of_sorted_out)
{
	if (A->format != SPMV_MI355X_SELL_C_SIGMA)
	{
		set_error("not a SELL handle");
		return 1;
	}
	if (A->sell_window)
	{
		set_error("sell_layout: not available for the LDS-window layout (16-bit window-relative indices)");
		return 1;
	}
	HIP_TRY(hipSetDevice(A->device));
	if (C_out) *C_out = A->sell_c;
	if (sigma_out) *sigma_out = A->sell_sigma;
	if (num_slices_out) *num_slices_out = A->sell_slices;
	if (nnz_ext_out) *nnz_ext_out = A->sell_nnz_ext;
	std::vector<int64_t> h_desc;
	std::vector<unsigned char> h_idx;
	if (A->sell_delta)
	{
		h_desc.resize(2 * ((size_t) A->sell_slices + 1));
		HIP_TRY(hipMemcpy(h_desc.data(), A->d_sell_desc, h_desc.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
		h_idx.resize((size_t) std::max<long>(A->sell_idx_bytes, 1));
		HIP_TRY(hipMemcpy(h_idx.data(), A->d_sell_idx, (size_t) A->sell_idx_bytes, hipMemcpyDeviceToHost));
	}
	if (slice_ptr_out)
	{
		*slice_ptr_out = (int64_t *) malloc(((size_t) A->sell_slices + 1) * sizeof(int64_t));
		if (A->sell_delta)
			for (long sl = 0; sl <= A->sell_slices; sl++)
				(*slice_ptr_out)[sl] = h_desc[2 * sl];
		else
			HIP_TRY(hipMemcpy(*slice_ptr_out, A->d_slice_ptr, ((size_t) A->sell_slices + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
	}
	size_t ne = (size_t) std::max<long>(A->sell_nnz_ext, 1);
	if (col_out)
	{
		*col_out = (int32_t *) malloc(ne * sizeof(int32_t));
		if (A->sell_delta)
		{
			// decode the compressed indices back to the plain column-major layout
			for (long sl = 0; sl < A->sell_slices; sl++)
			{
				const int64_t vb = h_desc[2 * sl];
				const long width = (h_desc[2 * sl + 2] - vb) / 64;
				const int md = (int) (h_desc[2 * sl + 1] & 7);
				const unsigned char * ib = h_idx.data() + (h_desc[2 * sl + 1] & ~(int64_t) 15);
				const int * offs = reinterpret_cast<const int *>(ib);
				unsigned long long exmask = 0;
				if (md == 3)
					ib += 4 * 64;
				if (md == 5)
				{
					exmask = *reinterpret_cast<const unsigned long long *>(ib + 4 * 64);
					ib += 4 * 64 + 16;
				}
				const long gbytes = md == 5 ? 16 + (__builtin_popcountll(exmask) + 3) / 4 * 16 : (md == 0 || md == 3) ? 16 : md == 1 ? 272 : md == 2 ? 528 : 1024;
				for (long k = 0; k < width; k++)
				{
					const unsigned char * gp = ib + (k / 4) * gbytes;
					const long u = k % 4;
					for (int r = 0; r < 64; r++)
					{
						int c;
						if (md == 0)
							c = reinterpret_cast<const int *>(gp)[u] + r;
						else if (md == 3)
							c = reinterpret_cast<const int *>(gp)[u] + offs[r];
						else if (md == 5)
							c = reinterpret_cast<const int *>(gp)[u] + offs[r] +
							    (((exmask >> r) & 1ull) ? (int) reinterpret_cast<const signed char *>(gp + 16)[4 * __builtin_popcountll(exmask & ((1ull << r) - 1ull)) + u] : 0);
						else if (md == 1)
							c = reinterpret_cast<const int *>(gp)[u] + gp[16 + r * 4 + u];
						else if (md == 2)
							c = reinterpret_cast<const int *>(gp)[u] + reinterpret_cast<const unsigned short *>(gp + 16)[r * 4 + u];
						else
							c = reinterpret_cast<const int *>(gp)[u * 64 + r];
						(*col_out)[vb + k * 64 + r] = c;
					}
				}
			}
		}
		else
			HIP_TRY(hipMemcpy(*col_out, A->d_col, (size_t) A->sell_nnz_ext * sizeof(int32_t), hipMemcpyDeviceToHost));
	}
	if (val_out)
	{
		*val_out = (double *) malloc(ne * sizeof(double));
		if (!A->f32)
			HIP_TRY(hipMemcpy(*val_out, A->d_val, (size_t) A->sell_nnz_ext * sizeof(double), hipMemcpyDeviceToHost));
		else
		{
			std::vector<float> tmp(ne);
			HIP_TRY(hipMemcpy(tmp.data(), A->d_val, (size_t) A->sell_nnz_ext * sizeof(float), hipMemcpyDeviceToHost));
			for (size_t i = 0; i < (size_t) A->sell_nnz_ext; i++)
				(*val_out)[i] = tmp[i];
		}
		if (A->sell_delta)
		{
			// the delta layout keeps a lane's steps in pairs (launch.hpp: sell_pair_pos): back to plain column-major
			std::vector<double> slice;
			for (long sl = 0; sl < A->sell_slices; sl++)
			{
				const int64_t vb = h_desc[2 * sl];
				const long width = (h_desc[2 * sl + 2] - vb) / 64;
				slice.assign(*val_out + vb, *val_out + vb + width * 64);
				for (long k = 0; k < width; k++)
					for (long r = 0; r < 64; r++)
						(*val_out)[vb + k * 64 + r] = slice[(size_t) spmv::sell_pair_pos(k, width, r)];
			}
		}
	}
	if (row_of_sorted_out)
	{
		*row_of_sorted_out = (int32_t *) malloc((size_t) std::max<long>(A->m, 1) * sizeof(int32_t));
		HIP_TRY(hipMemcpy(*row_of_sorted_out, A->d_row_of_sorted, (size_t) A->m * sizeof(int32_t), hipMemcpyDeviceToHost));
	}
	return 0;
}

int
spmv_mi355x_merge_tiles(const spmv_mi355x_matrix * A, long * num_tiles_out, long * tile_items_out, int32_t ** coords_out)
{
	if (A->format != SPMV_MI355X_CSR_MERGE)
	{
		set_error("not a merge handle");
		return 1;
	}
	HIP_TRY(hipSetDevice(A->device));
	if (num_tiles_out) *num_tiles_out = A->merge_num_tiles;
	if (tile_items_out) *tile_items_out = A->merge_tile;
	if (coords_out)
	{
		size_t cnt = 2 * ((size_t) A->merge_num_tiles + 1);
		*coords_out = (int32_t *) malloc(cnt * sizeof(int32_t));
		HIP_TRY(hipMemcpy(*coords_out, A->d_coords, cnt * sizeof(int32_t), hipMemcpyDeviceToHost));
	}
	return 0;
}

void
spmv_mi355x_free(void * p)
{
	free(p);
}

}  // extern "C"
