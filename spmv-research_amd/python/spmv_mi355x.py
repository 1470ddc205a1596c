"""ctypes binding of the C ABI in include/spmv_mi355x.h (libspmv_mi355x.so, hand-written HIP for gfx950).

This is test/bench plumbing above the C ABI: the same entry points the reference harness would bind through
host/spmv_kernel_mi355x.cpp. There is no fallback: if the shared object is missing or no GPU is usable, the
calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("SPMV_MI355X_LIB") or os.path.join(PKG_ROOT, "lib", "libspmv_mi355x.so")     # the override is for kernel experiments

CSR_SCALAR, CSR_VECTOR, CSR_MERGE, SELL_C_SIGMA, COO, CSR_STREAM = range(6)
FORMATS = {"csr_scalar": CSR_SCALAR, "csr_vector": CSR_VECTOR, "csr_merge": CSR_MERGE,
           "sell_c_sigma": SELL_C_SIGMA, "coo": COO, "csr_stream": CSR_STREAM}
F64, F32 = 0, 1


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("device", C.c_int), ("lanes_per_row", C.c_int),
                ("sell_split", C.c_int), ("sell_c", C.c_int), ("sell_sigma", C.c_int),
                ("merge_items", C.c_int), ("xcd_remap", C.c_int), ("nontemporal", C.c_int),
                ("stream_mode", C.c_int),
                ("row_begin", C.c_long), ("row_end", C.c_long), ("col_begin", C.c_long), ("col_end", C.c_long),
                ("col_filter_mode", C.c_int), ("sell_delta", C.c_int), ("convert_on", C.c_int),
                ("symmetric_input", C.c_int), ("rows_per_group", C.c_int), ("col_blocks", C.c_int),
                ("sell_window", C.c_int), ("kahan", C.c_int), ("sell_group", C.c_int), ("placement", C.c_int),
                ("placement_budget_gib", C.c_int)]


# every symbol declared in include/spmv_mi355x.h (checked by tests/test_abi.py)
SYMBOLS = [
    "spmv_mi355x_last_error", "spmv_mi355x_device_count", "spmv_mi355x_device_info", "spmv_mi355x_create",
    "spmv_mi355x_destroy", "spmv_mi355x_format_name", "spmv_mi355x_mem_footprint", "spmv_mi355x_csr_mem_footprint",
    "spmv_mi355x_rows", "spmv_mi355x_cols", "spmv_mi355x_nnz", "spmv_mi355x_spmv", "spmv_mi355x_set_always_copy",
    "spmv_mi355x_upload_x", "spmv_mi355x_download_y", "spmv_mi355x_spmv_device_async", "spmv_mi355x_time_device",
    "spmv_mi355x_kernel_info", "spmv_mi355x_x_device", "spmv_mi355x_y_device", "spmv_mi355x_sell_layout", "spmv_mi355x_stored_array",
    "spmv_mi355x_merge_tiles", "spmv_mi355x_free", "spmv_mi355x_precision", "spmv_mi355x_device",
    "spmv_mi355x_pcg", "spmv_mi355x_pbicgstab", "spmv_mi355x_pcg_dist", "spmv_mi355x_pbicgstab_dist",
    "spmv_mi355x_copy_device_async",
    "spmv_mi355x_create_partitioned", "spmv_mi355x_destroy_partitioned", "spmv_mi355x_spmv_partitioned",
    "spmv_mi355x_partitioned_set_always_copy", "spmv_mi355x_time_partitioned", "spmv_mi355x_partitioned_parts",
    "spmv_mi355x_partitioned_offsets", "spmv_mi355x_partitioned_format_name", "spmv_mi355x_partitioned_exchange",
    "spmv_mi355x_partitioned_mem_footprint",
    "spmv_mi355x_upload_y", "spmv_mi355x_output_alloc", "spmv_mi355x_input_alloc", "spmv_mi355x_output_free", "spmv_mi355x_placement_release", "spmv_mi355x_placement_info", "spmv_mi355x_place_arrays",
    "spmv_mi355x_csr_stream_begin", "spmv_mi355x_csr_stream_append", "spmv_mi355x_create_from_stream", "spmv_mi355x_csr_stream_discard",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C spmv-research_amd` "
                               "(__graft_entry__.build()). There is no CPU fallback.")
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 and refuses to initialise ("No HIP GPUs
        # are available") when the system copy was loaded first. Loading torch's first lets both share it.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        L.spmv_mi355x_last_error.restype = C.c_char_p
        L.spmv_mi355x_format_name.restype = C.c_char_p
        L.spmv_mi355x_mem_footprint.restype = C.c_double
        L.spmv_mi355x_csr_mem_footprint.restype = C.c_double
        for f in ("spmv_mi355x_rows", "spmv_mi355x_cols", "spmv_mi355x_nnz"):
            getattr(L, f).restype = C.c_long
        L.spmv_mi355x_partitioned_format_name.restype = C.c_char_p
        L.spmv_mi355x_partitioned_exchange.restype = C.c_char_p
        L.spmv_mi355x_partitioned_mem_footprint.restype = C.c_double
        L.spmv_mi355x_x_device.restype = C.c_void_p
        L.spmv_mi355x_y_device.restype = C.c_void_p
        _lib = L
    return _lib


class SpmvError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise SpmvError(lib().spmv_mi355x_last_error().decode())


def placement_release(device=-1):
    """Free the vector pools of a device (-1: all) — no vector of them may be live (include/spmv_mi355x.h)."""
    _check(lib().spmv_mi355x_placement_release(C.c_int(device)))


def placement_info(device=0):
    """What the one walk of a device found (include/spmv_mi355x.h): dict(state, candidates, walked_gib, pools, us_per_pool)."""
    st, cand, gib, npool, us = C.c_int(), C.c_int(), C.c_long(), C.c_int(), (C.c_double * 4)()
    _check(lib().spmv_mi355x_placement_info(C.c_int(device), C.byref(st), C.byref(cand), C.byref(gib), C.byref(npool), us))
    return dict(state={0: "no walk", 1: "pools kept", 2: "no contrast: plain allocations"}[st.value], candidates=cand.value, walked_gib=gib.value,
                pools=npool.value, us_per_pool=[round(us[k], 1) for k in range(npool.value)])


def device_count():
    c = C.c_int()
    _check(lib().spmv_mi355x_device_count(C.byref(c)))
    return c.value


def device_info(device=0):
    name = C.create_string_buffer(256)
    cu = C.c_int()
    mem = C.c_long()
    _check(lib().spmv_mi355x_device_info(device, name, C.c_long(256), C.byref(cu), C.byref(mem)))
    return dict(name=name.value.decode(), compute_units=cu.value, hbm_bytes=mem.value)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OutputVector:
    """A device vector a handle's SpMV writes, allocated by the engine (include/spmv_mi355x.h "output vectors placed by the
    engine"): placed by timing the handle's own kernel on it. `.ptr` goes to spmv_device(); `.torch()` is a zero-copy torch
    view (CUDA array interface) for callers that fill / check it with torch."""

    def __init__(self, matrix, count, is_input=False):
        self.count, self.dtype = int(count), np.dtype(matrix.dtype)
        self.nbytes = max(self.count, 1) * self.dtype.itemsize
        out = C.c_void_p()
        alloc = lib().spmv_mi355x_input_alloc if is_input else lib().spmv_mi355x_output_alloc
        _check(alloc(matrix.h, C.c_size_t(self.nbytes), C.byref(out)))
        self.ptr = out.value
        self._view = None

    @property
    def __cuda_array_interface__(self):
        return dict(shape=(self.count,), typestr=self.dtype.str, data=(self.ptr, False), version=2, strides=None)

    def torch(self):
        import torch
        if self._view is None:
            self._view = torch.as_tensor(self, device="cuda")
        return self._view

    def free(self):
        if getattr(self, "ptr", None):
            self._view = None
            lib().spmv_mi355x_output_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _make_opts(opts):
    o = Opts()
    o.struct_size = C.sizeof(Opts)
    o.device = -1
    for k, v in opts.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k}")
        setattr(o, k, v)
    return o


class CsrStream:
    """A CSR assembled in device memory from pieces of consecutive rows (include/spmv_mi355x.h "a handle from a CSR that arrives in
    pieces"): append(row_ptr, col_idx, values) per piece, finish(fmt, dtype, **opts) -> Matrix. The host never holds more than a
    piece. SELL-C-sigma (64-row slices, delta layout) only."""

    def __init__(self, m, n, nnz_capacity, device=-1):
        self.s = C.c_void_p()
        _check(lib().spmv_mi355x_csr_stream_begin(C.byref(self.s), C.c_int(device), C.c_long(m), C.c_long(n), C.c_long(nnz_capacity)))

    def append(self, row_ptr, col_idx, values):
        row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        col_idx = np.ascontiguousarray(col_idx, np.int32)
        values = np.ascontiguousarray(values, np.float64)
        _check(lib().spmv_mi355x_csr_stream_append(self.s, C.c_long(len(row_ptr) - 1), _p(row_ptr), _p(col_idx), _p(values)))

    def finish(self, fmt="sell_c_sigma", dtype=np.float64, **opts):
        o = _make_opts(opts)
        h = C.c_void_p()
        s, self.s = self.s, None                       # consumed, whatever happens
        fmt_id = FORMATS[fmt] if isinstance(fmt, str) else fmt
        _check(lib().spmv_mi355x_create_from_stream(C.byref(h), s, fmt_id, F64 if np.dtype(dtype) == np.float64 else F32, C.byref(o)))
        return Matrix(None, None, None, 0, 0, fmt, dtype, _handle=h)

    def discard(self):
        if getattr(self, "s", None):
            lib().spmv_mi355x_csr_stream_discard(self.s)
            self.s = None

    def __del__(self):
        try:
            self.discard()
        except Exception:
            pass


class SolverInfo(C.Structure):
    """spmv_mi355x_solver_info (include/spmv_mi355x.h)"""
    _fields_ = [("struct_size", C.c_uint), ("iterations", C.c_long), ("error", C.c_double), ("error_best", C.c_double),
                ("eps", C.c_double), ("eps_counter", C.c_double), ("restarts", C.c_long), ("spmv_calls", C.c_long),
                ("seconds", C.c_double)]


SPMV_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)


class DistOps(C.Structure):
    """spmv_mi355x_dist_ops (include/spmv_mi355x.h)"""
    _fields_ = [("struct_size", C.c_uint), ("row_offset", C.c_long), ("spmv", SPMV_CB), ("allreduce_sum", ALLREDUCE_CB),
                ("reduce_buf_dev", C.c_void_p), ("ctx", C.c_void_p)]


def solve_distributed(method, ops, dtype, m_local, row_ptr, col_idx, values, b, max_iterations, history=True):
    """spmv_mi355x_pcg_dist / spmv_mi355x_pbicgstab_dist: returns the same dict as Matrix.pcg for the LOCAL slice of x."""
    dtype = np.dtype(dtype)
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    values = np.ascontiguousarray(values, np.float64)
    b = np.ascontiguousarray(b, dtype)
    x = np.zeros(max(m_local, 1), dtype)
    hist = np.zeros((max(max_iterations, 1), 3), np.float64) if history else None
    info = SolverInfo()
    info.struct_size = C.sizeof(SolverInfo)
    fn = lib().spmv_mi355x_pcg_dist if method == "pcg" else lib().spmv_mi355x_pbicgstab_dist
    _check(fn(C.byref(ops), F64 if dtype == np.float64 else F32, C.c_long(m_local), _p(row_ptr), _p(col_idx), _p(values),
              _p(b), _p(x), C.c_long(max_iterations), _p(hist) if history else None, C.byref(info)))
    out = {k: getattr(info, k) for k, _ in SolverInfo._fields_ if k != "struct_size"}
    out["x"] = x[:m_local]
    out["history"] = hist[:info.iterations] if history else None
    return out


class PartitionedMatrix:
    """One matrix cut into nnz-balanced row blocks over several GPUs of one node behind ONE handle
    (spmv_mi355x_create_partitioned): what the reference's single-process driver would hold as its Matrix_Format."""

    def __init__(self, row_ptr, col_idx, values, m, n, nparts, fmt="sell_c_sigma", dtype=np.float64, devices=None, exchange=0, **opts):
        row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        col_idx = np.ascontiguousarray(col_idx, np.int32)
        values = np.ascontiguousarray(values, np.float64)
        self.dtype = np.dtype(dtype)
        o = Opts()
        o.struct_size = C.sizeof(Opts)
        o.device = -1
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k}")
            setattr(o, k, v)
        dev = None if devices is None else np.ascontiguousarray(devices, np.int32)
        self.h = C.c_void_p()
        fmt_id = FORMATS[fmt] if isinstance(fmt, str) else fmt
        _check(lib().spmv_mi355x_create_partitioned(C.byref(self.h), C.c_int(nparts), None if dev is None else _p(dev), C.c_int(exchange),
                                                    fmt_id, F64 if self.dtype == np.float64 else F32, C.c_long(m), C.c_long(n),
                                                    C.c_long(len(col_idx)), _p(row_ptr), _p(col_idx), _p(values), C.byref(o)))
        L = lib()
        self.m, self.n, self.nparts = m, n, L.spmv_mi355x_partitioned_parts(self.h)
        self.format_name = L.spmv_mi355x_partitioned_format_name(self.h).decode()
        self.exchange = L.spmv_mi355x_partitioned_exchange(self.h).decode()
        self.mem_footprint = L.spmv_mi355x_partitioned_mem_footprint(self.h)
        off = np.zeros(self.nparts + 1, np.int64)
        _check(L.spmv_mi355x_partitioned_offsets(self.h, _p(off)))
        self.offsets = off

    def spmv(self, x, always_copy=True):
        x = np.ascontiguousarray(x, self.dtype)
        assert x.shape[0] == self.n
        y = np.ones(self.m + 64, self.dtype)
        lib().spmv_mi355x_partitioned_set_always_copy(self.h, 1 if always_copy else 0)
        _check(lib().spmv_mi355x_spmv_partitioned(self.h, _p(x), _p(y)))
        return y[:self.m].copy()

    def time(self, iters):
        ms = C.c_double()
        _check(lib().spmv_mi355x_time_partitioned(self.h, C.c_int(iters), C.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            lib().spmv_mi355x_destroy_partitioned(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Matrix:
    """One converted matrix on one GPU = the reference's `struct Matrix_Format` instance."""

    def __init__(self, row_ptr, col_idx, values, m, n, fmt="csr_vector", dtype=np.float64, _handle=None, **opts):
        if _handle is not None:                        # CsrStream.finish(): the handle exists already
            self.dtype = np.dtype(dtype)
            self.h = _handle
            self._describe()
            return
        row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        col_idx = np.ascontiguousarray(col_idx, np.int32)
        values = np.ascontiguousarray(values, np.float64)
        self.dtype = np.dtype(dtype)
        o = Opts()
        o.struct_size = C.sizeof(Opts)
        o.device = -1
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k}")
            setattr(o, k, v)
        self.h = C.c_void_p()
        fmt_id = FORMATS[fmt] if isinstance(fmt, str) else fmt
        _check(lib().spmv_mi355x_create(C.byref(self.h), fmt_id, F64 if self.dtype == np.float64 else F32,
                                        C.c_long(m), C.c_long(n), C.c_long(len(col_idx)),
                                        _p(row_ptr), _p(col_idx), _p(values), C.byref(o)))
        self._describe()

    def _describe(self):
        L = lib()
        self.m = L.spmv_mi355x_rows(self.h)
        self.n = L.spmv_mi355x_cols(self.h)
        self.nnz = L.spmv_mi355x_nnz(self.h)
        self.format_name = L.spmv_mi355x_format_name(self.h).decode()
        self.mem_footprint = L.spmv_mi355x_mem_footprint(self.h)
        self.csr_mem_footprint = L.spmv_mi355x_csr_mem_footprint(self.h)

    # Matrix_Format::spmv(x, y) on host buffers; y gets the driver's +64 slack and 1.0 canary (bench_spmv.cpp:606-609)
    def spmv(self, x, always_copy=True):
        x = np.ascontiguousarray(x, self.dtype)
        assert x.shape[0] == self.n
        y = np.ones(self.m + 64, self.dtype)
        lib().spmv_mi355x_set_always_copy(self.h, 1 if always_copy else 0)
        _check(lib().spmv_mi355x_spmv(self.h, _p(x), _p(y)))
        return y[:self.m].copy()

    def spmv_raw(self, x, y):
        """Exact reference call shape: caller-owned buffers, reference caching semantics."""
        _check(lib().spmv_mi355x_spmv(self.h, _p(x), _p(y)))

    def set_always_copy(self, on):
        lib().spmv_mi355x_set_always_copy(self.h, 1 if on else 0)

    def spmv_device(self, x_ptr, y_ptr, beta=0, stream=0):
        _check(lib().spmv_mi355x_spmv_device_async(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), C.c_int(beta),
                                                   C.c_void_p(stream)))

    def time_device(self, x_ptr, y_ptr, iters, stream=0):
        ms = C.c_double()
        _check(lib().spmv_mi355x_time_device(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), C.c_int(iters),
                                             C.c_void_p(stream), C.byref(ms)))
        return ms.value

    def kernel_info(self):
        name = C.create_string_buffer(128)
        grid = C.c_long()
        block = C.c_int()
        _check(lib().spmv_mi355x_kernel_info(self.h, name, C.c_long(128), C.byref(grid), C.byref(block)))
        return dict(name=name.value.decode(), grid=grid.value, block=block.value)

    def x_device(self):
        return lib().spmv_mi355x_x_device(self.h)

    def y_device(self):
        return lib().spmv_mi355x_y_device(self.h)

    def upload_x(self, x):
        x = np.ascontiguousarray(x, self.dtype)
        _check(lib().spmv_mi355x_upload_x(self.h, _p(x)))

    def output_vector(self, count=None):
        """An engine-placed device vector for this handle's SpMV to write (rows + 64 values by default)."""
        return OutputVector(self, self.m + 64 if count is None else count)

    def input_vector(self, count=None):
        """An engine-placed device vector for this handle's SpMV to READ as x (cols values by default); `.torch()` gives the zero-copy
        tensor a collective can write into."""
        return OutputVector(self, self.n if count is None else count, is_input=True)

    def upload_y(self, y):
        y = np.ascontiguousarray(y, self.dtype)
        assert y.shape[0] >= self.m
        _check(lib().spmv_mi355x_upload_y(self.h, _p(y)))

    def download_y(self):
        y = np.zeros(self.m, self.dtype)
        _check(lib().spmv_mi355x_download_y(self.h, _p(y)))
        return y

    def sell_layout(self):
        Cc, sig, ns, ne = C.c_long(), C.c_long(), C.c_long(), C.c_long()
        sp = C.POINTER(C.c_int64)()
        col = C.POINTER(C.c_int32)()
        val = C.POINTER(C.c_double)()
        ros = C.POINTER(C.c_int32)()
        _check(lib().spmv_mi355x_sell_layout(self.h, C.byref(Cc), C.byref(sig), C.byref(ns), C.byref(ne),
                                             C.byref(sp), C.byref(col), C.byref(val), C.byref(ros)))
        out = dict(C=Cc.value, sigma=sig.value, num_slices=ns.value, nnz_ext=ne.value,
                   slice_ptr=np.ctypeslib.as_array(sp, shape=(ns.value + 1,)).copy(),
                   col=np.ctypeslib.as_array(col, shape=(max(ne.value, 1),))[:ne.value].copy(),
                   val=np.ctypeslib.as_array(val, shape=(max(ne.value, 1),))[:ne.value].copy(),
                   row_of_sorted=np.ctypeslib.as_array(ros, shape=(max(self.m, 1),))[:self.m].copy())
        for p in (sp, col, val, ros):
            lib().spmv_mi355x_free(p)
        return out

    def place_arrays(self, x_ptr, y_ptr):
        """The search over the handle's matrix arrays (opts.placement = 3) for a caller's device vectors; y is overwritten."""
        _check(lib().spmv_mi355x_place_arrays(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def stored_array(self, name, dtype=np.uint8):
        """One stored array of the LDS-window SELL / column-blocked layout as it lies in device memory (include/spmv_mi355x.h)."""
        out, nb = C.c_void_p(), C.c_size_t()
        _check(lib().spmv_mi355x_stored_array(self.h, name.encode(), C.byref(out), C.byref(nb)))
        a = np.frombuffer(C.string_at(out.value, nb.value), dtype=np.uint8).copy().view(dtype) if nb.value else np.zeros(0, dtype)
        lib().spmv_mi355x_free(out)
        return a

    def merge_tiles(self):
        nt, ti = C.c_long(), C.c_long()
        co = C.POINTER(C.c_int32)()
        _check(lib().spmv_mi355x_merge_tiles(self.h, C.byref(nt), C.byref(ti), C.byref(co)))
        coords = np.ctypeslib.as_array(co, shape=(2 * (nt.value + 1),)).copy().reshape(-1, 2)
        lib().spmv_mi355x_free(co)
        return dict(num_tiles=nt.value, tile_items=ti.value, coords=coords)

    def _solve(self, fn, row_ptr, col_idx, values, b, max_iterations, history):
        row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        col_idx = np.ascontiguousarray(col_idx, np.int32)
        values = np.ascontiguousarray(values, np.float64)
        b = np.ascontiguousarray(b, self.dtype)
        assert b.shape[0] == self.m and len(row_ptr) == self.m + 1
        x = np.zeros(max(self.n, 1), self.dtype)
        hist = np.zeros((max(max_iterations, 1), 3), np.float64) if history else None
        info = SolverInfo()
        info.struct_size = C.sizeof(SolverInfo)
        _check(fn(self.h, _p(row_ptr), _p(col_idx), _p(values), _p(b), _p(x), C.c_long(max_iterations),
                  _p(hist) if history else None, C.byref(info)))
        out = {k: getattr(info, k) for k, _ in SolverInfo._fields_ if k != "struct_size"}
        out["x"] = x[:self.n]
        out["history"] = hist[:info.iterations] if history else None
        return out

    def pcg(self, row_ptr, col_idx, values, b, max_iterations, history=True):
        """preconditioned_cg() of bench_cg.cpp:93-322 with every vector resident in HBM."""
        return self._solve(lib().spmv_mi355x_pcg, row_ptr, col_idx, values, b, max_iterations, history)

    def pbicgstab(self, row_ptr, col_idx, values, b, max_iterations, history=True):
        """preconditioned_bicgstab() of bench_bicg.cpp:149-459 with every vector resident in HBM."""
        return self._solve(lib().spmv_mi355x_pbicgstab, row_ptr, col_idx, values, b, max_iterations, history)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            lib().spmv_mi355x_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
