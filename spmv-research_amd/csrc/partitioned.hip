// Row-partitioned SpMV over several GPUs of one node behind the C ABI (include/spmv_mi355x.h: spmv_mi355x_*_partitioned) — the
// form the reference's single-process driver can call (bench.cpp:600-603 hands ONE csr_to_format the whole matrix).
//
// The reference has no multi-GPU path; its nearest analogue is the per-thread row partition of the CPU CSR backend
// (csr.cpp:140 -> loop_partitioner_balance_prefix_sums, lib/parallel_util.h:156-184), which is applied here with one worker
// per GPU. Part p owns the nnz-balanced contiguous row block [offsets[p], offsets[p+1]) of A, the same block of y and the same
// slice of x. x lives on every device as `nparts` slices padded to one length, so that ONE equal-sized allgather fills it in
// place; column c owned by part q is renumbered q*padded + (c - offsets[q]) once, at create().
//
// Per SpMV, on every device:   local columns:  y  = A_loc x   (needs only the own slice of x)        stream `comp`
//                              exchange:       allgather(x)                                         stream `comm`
//                              remote columns: y += A_rem x   after the exchange                     stream `comp`
// so the exchange overlaps the local part. Each row is summed on one GPU as local + remote, in a fixed order.
//
// The exchange sits behind a small interface with two back ends:
//   * RCCL (ncclAllGather inside one group call, one communicator per device; xGMI between the GPUs) — bound at run time
//     from librccl.so.1, so that the library also loads where RCCL is absent or already loaded by the caller's framework;
//   * peer copies (hipMemcpyPeerAsync, or plain device-to-device copies when several parts share a device) — what a box
//     with one GPU can run, and the fall-back when RCCL refuses the device list.

#include <dlfcn.h>
#include <chrono>
#include <string>

#include <rccl/rccl.h>

#include "handle.hpp"

using namespace spmv;

namespace {

struct RcclApi {
	void * lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	const char * (*GetErrorString)(ncclResult_t) = nullptr;
	bool ok = false;
};

RcclApi &
rccl()
{
	static RcclApi api = [] {
		RcclApi a;
		for (const char * name : {"librccl.so.1", "librccl.so"})
			if ((a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
				break;
		if (!a.lib)
			return a;
		a.CommInitAll = (decltype(a.CommInitAll)) dlsym(a.lib, "ncclCommInitAll");
		a.CommDestroy = (decltype(a.CommDestroy)) dlsym(a.lib, "ncclCommDestroy");
		a.GroupStart = (decltype(a.GroupStart)) dlsym(a.lib, "ncclGroupStart");
		a.GroupEnd = (decltype(a.GroupEnd)) dlsym(a.lib, "ncclGroupEnd");
		a.AllGather = (decltype(a.AllGather)) dlsym(a.lib, "ncclAllGather");
		a.GetErrorString = (decltype(a.GetErrorString)) dlsym(a.lib, "ncclGetErrorString");
		a.ok = a.CommInitAll && a.CommDestroy && a.GroupStart && a.GroupEnd && a.AllGather && a.GetErrorString;
		return a;
	}();
	return api;
}

struct Part {
	int device = 0;
	long r0 = 0, r1 = 0;
	spmv_mi355x_matrix * loc = nullptr;    // columns inside the own slice of x
	spmv_mi355x_matrix * rem = nullptr;    // columns owned by the other parts
	void * x_full = nullptr;               // nparts padded slices; x_full and y are the own (engine-placed) x / y pair of `loc`
	void * y = nullptr;                    // the part's rows of y (+64 slack, as the driver's y)
	hipStream_t comp = nullptr, comm = nullptr;
	hipEvent_t x_ready = nullptr, done = nullptr;
};

// index in [lo,hi] whose value is closest to target (lib/macros/macrolib.h:537-590, as parallel_util.h:156-184 uses it)
long
closest(const int32_t * A, long lo, long hi, long target)
{
	if (target < A[lo])
		return lo;
	if (target > A[hi])
		return hi;
	long s = lo, e = hi;
	for (;;)
	{
		const long mid = (s + e) / 2;
		if (mid == s || mid == e)
			break;
		if (target > A[mid])
			s = mid;
		else
			e = mid;
	}
	if (target == A[s])
		return s;
	if (target == A[e])
		return e;
	return (labs(target - (long) A[s]) < labs(target - (long) A[e])) ? s : e;
}

// every partitioned entry point walks the devices with hipSetDevice: the caller's current device is put back on every way out
struct DeviceRestore {
	int saved = -1;
	DeviceRestore() { if (hipGetDevice(&saved) != hipSuccess) { (void) hipGetLastError(); saved = -1; } }
	~DeviceRestore() { if (saved >= 0) (void) hipSetDevice(saved); }
};

}  // namespace

struct spmv_mi355x_partitioned {
	int nparts = 0;
	long m = 0, n = 0, nnz = 0, padded = 0;
	bool f32 = false;
	size_t vbytes = 8;
	std::vector<Part> parts;
	std::vector<long> offsets;
	int exchange = 0;                      // 1 = RCCL allgather, 2 = peer copies
	std::vector<ncclComm_t> comms;
	const void * cached_x_host = nullptr;
	bool y_downloaded = false, always_copy = false;
	double mem_footprint = 0;
	char format_name[160] = "";
	char exchange_name[64] = "";
};

static void
destroy_parts(spmv_mi355x_partitioned * P)
{
	for (ncclComm_t c : P->comms)
		if (c)
			(void) rccl().CommDestroy(c);
	P->comms.clear();
	for (Part & p : P->parts)
	{
		(void) hipSetDevice(p.device);
		if (p.loc) spmv_mi355x_destroy(p.loc);
		if (p.rem) spmv_mi355x_destroy(p.rem);
		if (p.comp) (void) hipStreamDestroy(p.comp);
		if (p.comm) (void) hipStreamDestroy(p.comm);
		if (p.x_ready) (void) hipEventDestroy(p.x_ready);
		if (p.done) (void) hipEventDestroy(p.done);
	}
	P->parts.clear();
}

// The exchange of one SpMV: afterwards every device holds all slices in its x_full. Enqueued on the parts' `comm` streams.
// RCCL: one ncclAllGather per device inside ONE group call (the single-thread use of a single-process communicator set);
// copies: every part PULLS the other parts' slices into its own x_full — it only ever writes its own buffer, so the ordering
// against its own kernels is all that is needed.
static int
exchange_x(spmv_mi355x_partitioned * P)
{
	const size_t slice_bytes = (size_t) P->padded * P->vbytes;
	if (P->exchange == 1)
	{
		RcclApi & R = rccl();
		ncclResult_t rc = R.GroupStart();
		for (int p = 0; p < P->nparts && rc == ncclSuccess; p++)
		{
			Part & a = P->parts[p];
			// in place: the send buffer is the own slice inside the receive buffer
			rc = R.AllGather((const char *) a.x_full + (size_t) p * slice_bytes, a.x_full, (size_t) P->padded, P->f32 ? ncclFloat32 : ncclFloat64,
					P->comms[p], a.comm);
		}
		const ncclResult_t rc2 = R.GroupEnd();
		if (rc != ncclSuccess || rc2 != ncclSuccess)
		{
			set_error("RCCL allgather(x): %s", R.GetErrorString(rc != ncclSuccess ? rc : rc2));
			return 1;
		}
		return 0;
	}
	for (int p = 0; p < P->nparts; p++)
	{
		Part & a = P->parts[p];
		HIP_TRY(hipSetDevice(a.device));
		for (int q = 0; q < P->nparts; q++)
		{
			if (q == p)
				continue;
			const Part & b = P->parts[q];
			// b's own slice was put in place on b's comm stream (upload); x_ready is its latest marker on that stream
			HIP_TRY(hipStreamWaitEvent(a.comm, b.x_ready, 0));
			char * dst = (char *) a.x_full + (size_t) q * slice_bytes;
			const char * src = (const char *) b.x_full + (size_t) q * slice_bytes;
			if (a.device == b.device)
				HIP_TRY(hipMemcpyAsync(dst, src, slice_bytes, hipMemcpyDeviceToDevice, a.comm));
			else
				HIP_TRY(hipMemcpyPeerAsync(dst, a.device, src, b.device, slice_bytes, a.comm));
		}
	}
	return 0;
}

// One y = A x with everything device-resident: enqueue only, from the calling thread (about ten API calls per device and step:
// at 8 devices that is of the order of the GPUs' own ~0.2 ms — if the driver's run shows the host as the limit, the per-device
// sequences are the thing to capture in hipGraphs).
static int
step(spmv_mi355x_partitioned * P)
{
	for (Part & a : P->parts)
	{
		HIP_TRY(hipSetDevice(a.device));
		// the exchange overwrites the remote slices of x_full: it must not overtake the previous step's remote-column kernel
		HIP_TRY(hipStreamWaitEvent(a.comm, a.done, 0));
		if (spmv_mi355x_spmv_device_async(a.loc, a.x_full, a.y, 0, a.comp))
			return 1;
	}
	if ((P->nparts > 1 || P->exchange == 1) && exchange_x(P))
		return 1;
	for (Part & a : P->parts)
	{
		HIP_TRY(hipSetDevice(a.device));
		HIP_TRY(hipEventRecord(a.x_ready, a.comm));
		HIP_TRY(hipStreamWaitEvent(a.comp, a.x_ready, 0));
		if (spmv_mi355x_spmv_device_async(a.rem, a.x_full, a.y, 1, a.comp))
			return 1;
		HIP_TRY(hipEventRecord(a.done, a.comp));
	}
	return 0;
}

static int
sync_all(spmv_mi355x_partitioned * P)
{
	for (Part & a : P->parts)
	{
		HIP_TRY(hipSetDevice(a.device));
		HIP_TRY(hipStreamSynchronize(a.comm));
		HIP_TRY(hipStreamSynchronize(a.comp));
	}
	return 0;
}

extern "C" {

int
spmv_mi355x_create_partitioned(spmv_mi355x_partitioned ** out, int nparts, const int * devices, int exchange, int format, int precision,
		long m, long n, long nnz, const int32_t * row_ptr, const int32_t * col_idx, const double * values, const spmv_mi355x_opts * opts_in)
{
	DeviceRestore restore;
	*out = nullptr;
	if (nparts < 1 || nparts > 64)
	{
		set_error("create_partitioned: nparts = %d (1..64)", nparts);
		return 1;
	}
	if (m != n)
	{
		set_error("create_partitioned: the x slices follow the row blocks, the matrix must be square (m=%ld n=%ld)", m, n);
		return 1;
	}
	if (!row_ptr || row_ptr[0] != 0 || (long) row_ptr[m] != nnz || (nnz > 0 && (!col_idx || !values)))
	{
		set_error("create_partitioned: bad CSR arrays (row_ptr[0] must be 0, row_ptr[m] must equal nnz)");
		return 1;
	}
	for (long i = 0; i < m; i++)
		if (row_ptr[i + 1] < row_ptr[i])
		{
			set_error("row_ptr is not monotone at row %ld", i);
			return 1;
		}
	spmv_mi355x_opts o;
	memset(&o, 0, sizeof(o));
	o.struct_size = sizeof(o);
	if (opts_in)
		memcpy(&o, opts_in, std::min<size_t>(sizeof(o), (size_t) std::max(opts_in->struct_size, 0)));
	if (o.symmetric_input || o.row_begin || o.row_end || o.col_filter_mode)
	{
		set_error("create_partitioned: symmetric_input / row blocks / column filters are set by the partition itself");
		return 1;
	}
	int ndev = 0;
	spmv_mi355x_device_count(&ndev);
	if (ndev < 1)
	{
		set_error("no HIP device available: this engine has no CPU fallback");
		return 1;
	}
	spmv_mi355x_partitioned * P = new spmv_mi355x_partitioned();
	P->nparts = nparts;
	P->m = m;
	P->n = n;
	P->nnz = nnz;
	P->f32 = precision == SPMV_MI355X_F32;
	P->vbytes = P->f32 ? 4 : 8;
	// ---- nnz-balanced contiguous row blocks: worker p of W gets [closest(total*p/W), closest(total*(p+1)/W)) (parallel_util.h:156-184)
	P->offsets.assign((size_t) nparts + 1, 0);
	for (int p = 1; p < nparts; p++)
		P->offsets[p] = m > 0 ? closest(row_ptr, 0, m - 1, (nnz * p) / nparts) : 0;
	P->offsets[nparts] = m;
	for (int p = 1; p <= nparts; p++)
		P->offsets[p] = std::max(P->offsets[p], P->offsets[p - 1]);
	long widest = 0;
	for (int p = 0; p < nparts; p++)
		widest = std::max(widest, P->offsets[p + 1] - P->offsets[p]);
	P->padded = (widest + 63) / 64 * 64;
	if (P->padded * nparts >= 0x7fffffffL)
	{
		set_error("create_partitioned: padded x of %ld entries exceeds the int32 index range", P->padded * nparts);
		delete P;
		return 1;
	}
	const long n_x = P->padded * nparts;
	P->parts.resize((size_t) nparts);
	bool distinct = true;
	for (int p = 0; p < nparts; p++)
	{
		P->parts[p].device = devices ? devices[p] : p % ndev;
		if (P->parts[p].device < 0 || P->parts[p].device >= ndev)
		{
			set_error("create_partitioned: device %d out of range (%d devices)", P->parts[p].device, ndev);
			delete P;
			return 1;
		}
		for (int q = 0; q < p; q++)
			distinct = distinct && P->parts[q].device != P->parts[p].device;
	}
	// owner of a column = the part whose row block holds it; its position in the padded layout
	auto remap = [&](int c) {
		const long q = (long) (std::upper_bound(P->offsets.begin() + 1, P->offsets.end(), (long) c) - P->offsets.begin()) - 1;
		return (int) (q * P->padded + (c - P->offsets[q]));
	};
	int rc = 0;
	for (int p = 0; p < nparts && !rc; p++)
	{
		Part & a = P->parts[p];
		a.r0 = P->offsets[p];
		a.r1 = P->offsets[p + 1];
		const long lm = a.r1 - a.r0, e0 = row_ptr[a.r0], lnnz = row_ptr[a.r1] - e0;
		std::vector<int> l_rp((size_t) lm + 1), l_ci((size_t) std::max<long>(lnnz, 1));
		#pragma omp parallel for num_threads(spmv::host_threads())
		for (long i = 0; i <= lm; i++)
			l_rp[(size_t) i] = row_ptr[a.r0 + i] - (int) e0;
		long bad = -1;
		#pragma omp parallel for num_threads(spmv::host_threads()) reduction(max : bad)
		for (long j = 0; j < lnnz; j++)
		{
			const int c = col_idx[e0 + j];
			if (c < 0 || c >= n)
				bad = std::max(bad, e0 + j);
			else
				l_ci[(size_t) j] = remap(c);
		}
		if (bad >= 0)
		{
			set_error("column index %d out of range [0,%ld) at entry %ld", col_idx[bad], n, bad);
			rc = 1;
			break;
		}
		if (hipSetDevice(a.device) != hipSuccess)
		{
			set_error("hipSetDevice(%d) failed", a.device);
			rc = 1;
			break;
		}
		spmv_mi355x_opts po = o;
		po.device = a.device;
		po.col_begin = (long) p * P->padded;
		po.col_end = po.col_begin + lm;
		po.col_filter_mode = 1;
		rc = spmv_mi355x_create(&a.loc, format, precision, lm, n_x, lnnz, l_rp.data(), l_ci.data(), values + e0, &po);
		po.col_filter_mode = 2;
		rc = rc || spmv_mi355x_create(&a.rem, format, precision, lm, n_x, lnnz, l_rp.data(), l_ci.data(), values + e0, &po);
		if (!rc)
		{
			// the local-column handle has n_x columns and lm rows: its own x / y pair (placed relative to its arrays,
			// placement.hip) is exactly this part's x_full and y
			a.x_full = spmv_mi355x_x_device(a.loc);
			a.y = spmv_mi355x_y_device(a.loc);
			rc = !a.x_full || !a.y;
		}
		if (rc)
			break;
		if (hipMemset(a.x_full, 0, (size_t) n_x * P->vbytes) != hipSuccess || hipStreamCreate(&a.comp) != hipSuccess ||
		    hipStreamCreate(&a.comm) != hipSuccess || hipEventCreateWithFlags(&a.x_ready, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&a.done, hipEventDisableTiming) != hipSuccess || hipEventRecord(a.done, a.comp) != hipSuccess ||
		    hipEventRecord(a.x_ready, a.comm) != hipSuccess)
		{
			set_error("create_partitioned: stream / event / buffer setup failed on device %d", a.device);
			rc = 1;
			break;
		}
		P->mem_footprint += spmv_mi355x_mem_footprint(a.loc) + spmv_mi355x_mem_footprint(a.rem);
	}
	// ---- exchange back end
	if (!rc && (nparts > 1 || exchange == 1))          // one part with the RCCL back end forced: the library calls, on one device
	{
		if (exchange != 2 && distinct && rccl().ok)
		{
			std::vector<int> devs;
			for (const Part & a : P->parts)
				devs.push_back(a.device);
			P->comms.assign((size_t) nparts, nullptr);
			const ncclResult_t r = rccl().CommInitAll(P->comms.data(), nparts, devs.data());
			if (r == ncclSuccess)
			{
				P->exchange = 1;
				snprintf(P->exchange_name, sizeof(P->exchange_name), "RCCL allgather");
			}
			else
			{
				P->comms.clear();
				if (exchange == 1)
				{
					set_error("create_partitioned: ncclCommInitAll failed: %s", rccl().GetErrorString(r));
					rc = 1;
				}
			}
		}
		else if (exchange == 1)
		{
			set_error("create_partitioned: the RCCL exchange needs %s", !rccl().ok ? "librccl.so.1" : "distinct devices");
			rc = 1;
		}
		if (!rc && P->exchange == 0)
		{
			P->exchange = 2;
			snprintf(P->exchange_name, sizeof(P->exchange_name), distinct ? "peer copies" : "device copies (parts share a device)");
			for (const Part & a : P->parts)
				for (const Part & b : P->parts)
					if (a.device != b.device)
					{
						(void) hipSetDevice(a.device);
						const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
						if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
							(void) hipGetLastError();          // hipMemcpyPeerAsync stages through the host then
					}
		}
	}
	else if (!rc)
		snprintf(P->exchange_name, sizeof(P->exchange_name), "none (one part)");
	if (rc)
	{
		destroy_parts(P);
		delete P;
		return 1;
	}
	snprintf(P->format_name, sizeof(P->format_name), "MI355X_PART%d_%s", nparts, spmv_mi355x_format_name(P->parts[0].loc));
	*out = P;
	return 0;
}

int
spmv_mi355x_destroy_partitioned(spmv_mi355x_partitioned * P)
{
	if (!P)
		return 0;
	DeviceRestore restore;
	(void) sync_all(P);
	destroy_parts(P);
	delete P;
	return 0;
}

int spmv_mi355x_partitioned_parts(const spmv_mi355x_partitioned * P) { return P->nparts; }
const char * spmv_mi355x_partitioned_format_name(const spmv_mi355x_partitioned * P) { return P->format_name; }
const char * spmv_mi355x_partitioned_exchange(const spmv_mi355x_partitioned * P) { return P->exchange_name; }
double spmv_mi355x_partitioned_mem_footprint(const spmv_mi355x_partitioned * P) { return P->mem_footprint; }

int
spmv_mi355x_partitioned_offsets(const spmv_mi355x_partitioned * P, long * offsets_out)
{
	for (int p = 0; p <= P->nparts; p++)
		offsets_out[p] = P->offsets[p];
	return 0;
}

int
spmv_mi355x_partitioned_set_always_copy(spmv_mi355x_partitioned * P, int on)
{
	P->always_copy = on != 0;
	return 0;
}

// Matrix_Format::spmv(x, y) on host buffers, with the reference GPU backends' caching convention (csr_rocm_vector.cpp:224-257)
int
spmv_mi355x_spmv_partitioned(spmv_mi355x_partitioned * P, const void * x_host, void * y_host)
{
	DeviceRestore restore;
	const size_t slice_bytes = (size_t) P->padded * P->vbytes;
	if (P->always_copy || P->cached_x_host != x_host)
	{
		for (int p = 0; p < P->nparts; p++)
		{
			Part & a = P->parts[p];
			HIP_TRY(hipSetDevice(a.device));
			// every device gets its OWN slice; the exchange delivers the others
			HIP_TRY(hipMemcpyAsync((char *) a.x_full + (size_t) p * slice_bytes, (const char *) x_host + (size_t) a.r0 * P->vbytes,
					(size_t) (a.r1 - a.r0) * P->vbytes, hipMemcpyHostToDevice, a.comm));
			HIP_TRY(hipEventRecord(a.x_ready, a.comm));
			HIP_TRY(hipStreamWaitEvent(a.comp, a.x_ready, 0));
		}
		P->cached_x_host = x_host;
		P->y_downloaded = false;
	}
	if (step(P) || sync_all(P))
		return 1;
	if (P->always_copy || !P->y_downloaded)
	{
		for (Part & a : P->parts)
		{
			HIP_TRY(hipSetDevice(a.device));
			HIP_TRY(hipMemcpyAsync((char *) y_host + (size_t) a.r0 * P->vbytes, a.y, (size_t) (a.r1 - a.r0) * P->vbytes, hipMemcpyDeviceToHost, a.comp));
		}
		if (sync_all(P))
			return 1;
		P->y_downloaded = true;
	}
	return 0;
}

// `iters` SpMVs back to back with x resident (uploaded by an earlier spmv_partitioned call), the exchange forced every time;
// wall-clock per iteration between two all-device synchronisations (no single stream spans the devices)
int
spmv_mi355x_time_partitioned(spmv_mi355x_partitioned * P, int iters, double * ms_per_iter_out)
{
	DeviceRestore restore;
	if (sync_all(P))
		return 1;
	const auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < iters; i++)
		if (step(P))
			return 1;
	if (sync_all(P))
		return 1;
	const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	*ms_per_iter_out = iters > 0 ? ms / iters : 0;
	return 0;
}

}  // extern "C"
