"""TEST INFRASTRUCTURE — ctypes binding of oracle/_build/liboracle.so (see oracle/oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

METRIC_NAMES = ("mae", "max_ae", "mse", "mape", "smape", "lnQ_error", "mlare", "gmare")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("spmv_oracle.c", "mtx_oracle.c", "oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Coo(C.Structure):
    _fields_ = [("m", C.c_long), ("n", C.c_long), ("nnz", C.c_long), ("nnz_sym", C.c_long),
                ("nnz_diag", C.c_long), ("nnz_non_diag", C.c_long),
                ("symmetric", C.c_int), ("skew", C.c_int), ("hermitian", C.c_int),
                ("field", C.c_char * 16),
                ("R", C.POINTER(C.c_int32)), ("C", C.POINTER(C.c_int32)), ("V", C.POINTER(C.c_double))]


class _Sell(C.Structure):
    _fields_ = [("m", C.c_long), ("nnz", C.c_long), ("C", C.c_long), ("num_slices", C.c_long),
                ("nnz_ext", C.c_long),
                ("slice_ptr", C.POINTER(C.c_int32)), ("ja", C.POINTER(C.c_int32)), ("a", C.POINTER(C.c_double)),
                ("permutation", C.POINTER(C.c_int32)), ("rev_permutation", C.POINTER(C.c_int32)),
                ("mem_footprint", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_check_accuracy.restype = C.c_double
        _lib.orc_time_csr_spmv_f64.restype = C.c_double
    return _lib


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def partition_prefix_sums(num_workers, worker_pos, sums, N, total):
    s, e = C.c_long(), C.c_long()
    sums = _i32(sums)
    lib().orc_partition_prefix_sums(C.c_long(num_workers), C.c_long(worker_pos), _p(sums), C.c_long(N),
                                    C.c_long(total), C.byref(s), C.byref(e))
    return s.value, e.value


def partition_iterations(num_workers, worker_pos, start, end):
    s, e = C.c_long(), C.c_long()
    lib().orc_partition_iterations(C.c_long(num_workers), C.c_long(worker_pos), C.c_long(start), C.c_long(end),
                                   C.byref(s), C.byref(e))
    return s.value, e.value


def mtx_read(path):
    coo = _Coo()
    err = C.create_string_buffer(1000)
    rc = lib().orc_mtx_read(os.fsencode(path), 1, C.byref(coo), err, C.c_long(1000))
    if rc != 0:
        raise ValueError(err.value.decode())
    nnz = coo.nnz
    R = np.ctypeslib.as_array(coo.R, shape=(max(nnz, 1),))[:nnz].copy()
    Cc = np.ctypeslib.as_array(coo.C, shape=(max(nnz, 1),))[:nnz].copy()
    V = np.ctypeslib.as_array(coo.V, shape=(max(nnz, 1),))[:nnz].copy()
    info = dict(m=coo.m, n=coo.n, nnz=nnz, nnz_sym=coo.nnz_sym, nnz_diag=coo.nnz_diag,
                nnz_non_diag=coo.nnz_non_diag, symmetric=coo.symmetric, skew=coo.skew,
                hermitian=coo.hermitian, field=coo.field.decode())
    lib().orc_coo_free(C.byref(coo))
    return info, R, Cc, V


def coo_to_csr(R, Cc, V, m, n):
    R, Cc = _i32(R), _i32(Cc)
    V = np.ascontiguousarray(V, np.float64)
    nnz = len(R)
    row_ptr = np.zeros(m + 1, np.int32)
    col_idx = np.zeros(max(nnz, 1), np.int32)
    values = np.zeros(max(nnz, 1), np.float64)
    lib().orc_coo_to_csr(_p(R), _p(Cc), _p(V), C.c_long(m), C.c_long(n), C.c_long(nnz),
                         _p(row_ptr), _p(col_idx), _p(values))
    return row_ptr, col_idx[:nnz], values[:nnz]


def mtx_to_csr(path):
    info, R, Cc, V = mtx_read(path)
    return (info,) + coo_to_csr(R, Cc, V, info["m"], info["n"])


def _prep(row_ptr, col_idx, a, x, dtype):
    row_ptr, col_idx = _i32(row_ptr), _i32(col_idx)
    a = np.ascontiguousarray(a, dtype)     # narrowing from the fp64 reference values, as csr.cpp:72 does
    x = np.ascontiguousarray(x, dtype)
    m = len(row_ptr) - 1
    y = np.ones(m + 64, dtype)             # driver canary (bench_spmv.cpp:606-609)
    return row_ptr, col_idx, a, x, y, m


def csr_spmv(row_ptr, col_idx, a, x, dtype=np.float64, num_threads=1):
    row_ptr, col_idx, a, x, y, m = _prep(row_ptr, col_idx, a, x, dtype)
    f = lib().orc_csr_spmv_f64 if dtype == np.float64 else lib().orc_csr_spmv_f32
    f(_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), _p(x), _p(y), C.c_int(num_threads))
    return y[:m].copy()


def csr_kahan_spmv(row_ptr, col_idx, a, x):
    row_ptr, col_idx, a, x, y, m = _prep(row_ptr, col_idx, a, x, np.float64)
    lib().orc_csr_kahan_spmv_f64(_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), _p(x), _p(y))
    return y[:m].copy()


def csr_sym_spmv(row_ptr, col_idx, a, x, dtype=np.float64):
    """y = (L + L^T - diag) x from the stored (lower) triangle as csr_sym.cpp:191-267 computes it with one thread."""
    row_ptr, col_idx, a, x, y, m = _prep(row_ptr, col_idx, a, x, dtype)
    f = lib().orc_csr_sym_spmv_f64 if dtype == np.float64 else lib().orc_csr_sym_spmv_f32
    f(_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), _p(x), _p(y))
    return y[:m].copy()


def csr_vec_spmv(row_ptr, col_idx, a, x, vec_len, dtype=np.float64):
    row_ptr, col_idx, a, x, y, m = _prep(row_ptr, col_idx, a, x, dtype)
    f = lib().orc_csr_vec_spmv_f64 if dtype == np.float64 else lib().orc_csr_vec_spmv_f32
    f(_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), _p(x), _p(y), C.c_int(vec_len))
    return y[:m].copy()


def merge_spmv(row_ptr, col_idx, a, x, num_threads, dtype=np.float64):
    row_ptr, col_idx, a, x, y, m = _prep(row_ptr, col_idx, a, x, dtype)
    f = lib().orc_merge_spmv_f64 if dtype == np.float64 else lib().orc_merge_spmv_f32
    f(_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), C.c_long(len(col_idx)), _p(x), _p(y), C.c_int(num_threads))
    return y[:m].copy()


def merge_path_search(diagonal, row_end_offsets, a_len, b_len):
    xo, yo = C.c_long(), C.c_long()
    reo = _i32(row_end_offsets)
    lib().orc_merge_path_search(C.c_long(diagonal), _p(reo), C.c_long(a_len), C.c_long(b_len), C.byref(xo), C.byref(yo))
    return xo.value, yo.value


class Sell:
    """SELL-C-sigma built the way BENCH/spmv_kernels/sell_sorted.cpp builds it."""

    def __init__(self, row_ptr, col_idx, values, C_rows, num_threads, dtype=np.float64):
        row_ptr, col_idx = _i32(row_ptr), _i32(col_idx)
        values = np.ascontiguousarray(values, np.float64)
        self.dtype = dtype
        self.s = _Sell()
        m = len(row_ptr) - 1
        lib().orc_sell_sorted_build(_p(row_ptr), _p(col_idx), _p(values), C.c_long(m), C.c_long(len(col_idx)),
                                    C.c_int(C_rows), C.c_int(num_threads), C.c_int(np.dtype(dtype).itemsize),
                                    C.byref(self.s))
        s = self.s
        self.m, self.C, self.num_slices, self.nnz_ext = s.m, s.C, s.num_slices, s.nnz_ext
        self.mem_footprint = s.mem_footprint
        self.slice_ptr = np.ctypeslib.as_array(s.slice_ptr, shape=(s.num_slices + 1,)).copy()
        self.ja = np.ctypeslib.as_array(s.ja, shape=(max(s.nnz_ext, 1),))[:s.nnz_ext].copy()
        self.a = np.ctypeslib.as_array(s.a, shape=(max(s.nnz_ext, 1),))[:s.nnz_ext].copy()
        self.permutation = np.ctypeslib.as_array(s.permutation, shape=(max(m, 1),))[:m].copy()
        self.rev_permutation = np.ctypeslib.as_array(s.rev_permutation, shape=(max(m, 1),))[:m].copy()

    def spmv(self, x):
        x = np.ascontiguousarray(x, self.dtype)
        y = np.ones(self.m + 64, self.dtype)
        f = lib().orc_sell_spmv_f64 if self.dtype == np.float64 else lib().orc_sell_spmv_f32
        f(C.byref(self.s), _p(x), _p(y))
        return y[:self.m].copy()

    def __del__(self):
        try:
            lib().orc_sell_free(C.byref(self.s))
        except Exception:
            pass


def sellcs_layout(row_ptr, col_idx, a, C_rows, sigma):
    """The BSC SELL-C-sigma library's layout (orc_sellcs_layout): dict(row_order, widths, slice_ptr, col, val)."""
    row_ptr, col_idx = _i32(row_ptr), _i32(col_idx)
    a = np.ascontiguousarray(a, np.float64)
    m = len(row_ptr) - 1
    ns = (m + C_rows - 1) // C_rows
    order = np.zeros(max(m, 1), np.int32)
    widths = np.zeros(max(ns, 1), np.int32)
    sp = np.zeros(ns + 1, np.int64)
    L = lib()
    L.orc_sellcs_layout.restype = C.c_long
    args = (_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), C.c_long(C_rows), C.c_long(sigma), _p(order), _p(widths), _p(sp))
    total = L.orc_sellcs_layout(*args, None, None)
    col = np.zeros(max(total, 1), np.int32)
    val = np.zeros(max(total, 1), np.float64)
    L.orc_sellcs_layout(*args, _p(col), _p(val))
    return dict(row_order=order[:m], widths=widths[:ns], slice_ptr=sp, col=col[:total], val=val[:total])


def csr_to_coo_rows(row_ptr):
    row_ptr = _i32(row_ptr)
    m = len(row_ptr) - 1
    rowind = np.zeros(max(int(row_ptr[m]), 1), np.int32)
    lib().orc_csr_to_coo_rows(_p(row_ptr), C.c_long(m), _p(rowind))
    return rowind[:int(row_ptr[m])]


def coo_spmv(rowind, colind, val, m, x, dtype=np.float64):
    rowind, colind = _i32(rowind), _i32(colind)
    val = np.ascontiguousarray(val, dtype)
    x = np.ascontiguousarray(x, dtype)
    y = np.ones(m + 64, dtype)
    f = lib().orc_coo_spmv_f64 if dtype == np.float64 else lib().orc_coo_spmv_f32
    f(_p(rowind), _p(colind), _p(val), C.c_long(m), C.c_long(len(rowind)), _p(x), _p(y))
    return y[:m].copy()


def gold_spmv(row_ptr, col_idx, a, x):
    row_ptr, col_idx = _i32(row_ptr), _i32(col_idx)
    a = np.ascontiguousarray(a, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    m = len(row_ptr) - 1
    y = np.zeros(max(m, 1), np.float64)
    lib().orc_gold_spmv(_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), _p(x), _p(y))
    return y[:m]


def check_accuracy(row_ptr, col_idx, a, x, y_test, is_double=True):
    """Returns (maxDiff, dict of the 8 CSV error metrics) as BENCH/bench_spmv.cpp:108-235 computes them."""
    row_ptr, col_idx = _i32(row_ptr), _i32(col_idx)
    a = np.ascontiguousarray(a, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    y_test = np.ascontiguousarray(y_test, np.float64)
    m = len(row_ptr) - 1
    met = np.zeros(8, np.float64)
    md = lib().orc_check_accuracy(_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), _p(x), _p(y_test),
                                  C.c_int(1 if is_double else 0), _p(met))
    return md, dict(zip(METRIC_NAMES, met.tolist()))


def time_csr_spmv(row_ptr, col_idx, a, x, num_threads, min_loops=64, min_runtime=2.0):
    row_ptr, col_idx, a, x, y, m = _prep(row_ptr, col_idx, a, x, np.float64)
    loops, tmin, tmax = C.c_long(), C.c_double(), C.c_double()
    med = lib().orc_time_csr_spmv_f64(_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), _p(x), _p(y),
                                      C.c_int(num_threads), C.c_long(min_loops), C.c_double(min_runtime),
                                      C.byref(loops), C.byref(tmin), C.byref(tmax))
    return dict(median=med, min=tmin.value, max=tmax.value, loops=loops.value, y=y[:m].copy())


def _solve(fn, row_ptr, col_idx, a, b, max_iterations):
    row_ptr, col_idx = _i32(row_ptr), _i32(col_idx)
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    m = len(row_ptr) - 1
    x = np.zeros(max(m, 1), np.float64)
    hist = np.zeros((max(max_iterations, 1), 3), np.float64)
    info = np.zeros(4, np.float64)
    fn.restype = C.c_long
    k = fn(_p(row_ptr), _p(col_idx), _p(a), C.c_long(m), C.c_long(len(b)), _p(b), _p(x), C.c_long(max_iterations),
           _p(hist), _p(info))
    if k == -1:
        raise ValueError("bad K, zero in diagonal")
    if k == -2:
        raise ValueError("the matrix must be square")
    return dict(x=x[:m], iterations=int(k), history=hist[:k], eps=info[0], eps_counter=info[1], err_best=info[2],
                restarts=int(info[3]))


def pcg(row_ptr, col_idx, a, b, max_iterations):
    """Jacobi-preconditioned CG as bench_cg.cpp:93-322 (one thread). PARITY UNPINNED, see solver_oracle.c."""
    return _solve(lib().orc_pcg_f64, row_ptr, col_idx, a, b, max_iterations)


def pbicgstab(row_ptr, col_idx, a, b, max_iterations):
    """Jacobi-preconditioned BiCGSTAB as bench_bicg.cpp:149-459 (one thread). PARITY UNPINNED."""
    return _solve(lib().orc_pbicgstab_f64, row_ptr, col_idx, a, b, max_iterations)
