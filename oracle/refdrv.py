"""TEST INFRASTRUCTURE — ctypes driver for the genuine reference build under oracle/_ref/.

Only usable where oracle/_ref/<flavour>/libref_*.so exist (they are produced by `make -C oracle ref`
in the build container from /root/reference; see oracle/Makefile). Used by oracle/gen_golden.py to
produce tests/golden/ fixtures, by tests to cross-check the restatement live when the libs are
present, and by bench.py's cpu_baseline leg (kind "reference").
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def ref_lib_path(backend, prec, flavour="native"):
    return os.path.join(_HERE, "_ref", flavour, f"libref_{backend}_{prec}.so")


def _cpu_has_avx512():
    try:
        with open("/proc/cpuinfo") as f:
            return "avx512f" in f.read()
    except OSError:
        return False


def available(backend="csr", prec="d", flavour="native"):
    if flavour == "native" and not os.path.exists("/root/reference"):
        return False   # -march=native objects are only trusted on the machine that built them
    return os.path.exists(ref_lib_path(backend, prec, flavour))


class RefBackend:
    """One reference backend TU (csr, csr_kahan, csr_vec, sell_sorted) in one precision."""

    def __init__(self, backend, prec="d", flavour="native", threads=None):
        self.lib = C.CDLL(ref_lib_path(backend, prec, flavour), mode=C.RTLD_LOCAL)
        L = self.lib
        L.ref_format_name.restype = C.c_char_p
        L.ref_mem_footprint.restype = C.c_double
        L.ref_csr_mem_footprint.restype = C.c_double
        L.ref_time_spmv.restype = C.c_double
        self.dtype = np.float64 if L.ref_sizeof_value() == 8 else np.float32
        if threads is not None:
            L.ref_set_threads(int(threads))
        self.threads = L.ref_max_threads()
        self._keep = None

    def mtx_to_csr(self, path):
        L = self.lib
        m, n, nnz, sym, nd, nnd = (C.c_long() for _ in range(6))
        ia = C.POINTER(C.c_int32)()
        ja = C.POINTER(C.c_int32)()
        a = C.POINTER(C.c_double)()
        L.ref_mtx_to_csr(os.fsencode(path), C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym),
                         C.byref(nd), C.byref(nnd), C.byref(ia), C.byref(ja), C.byref(a))
        row_ptr = np.ctypeslib.as_array(ia, shape=(m.value + 1,)).copy()
        col_idx = np.ctypeslib.as_array(ja, shape=(max(nnz.value, 1),))[:nnz.value].copy()
        values = np.ctypeslib.as_array(a, shape=(max(nnz.value, 1),))[:nnz.value].copy()
        for p in (ia, ja, a):
            L.ref_free(p)
        info = dict(m=m.value, n=n.value, nnz=nnz.value, symmetric=sym.value,
                    nnz_diag=nd.value, nnz_non_diag=nnd.value)
        return info, row_ptr, col_idx, values

    def mtx_to_csr_keep_symmetry(self, path):
        """KEEP_SYMMETRY loading (bench.cpp:131-136,180-192): the file's own entries only."""
        L = self.lib
        m, n, nnz, sym = (C.c_long() for _ in range(4))
        ia = C.POINTER(C.c_int32)()
        ja = C.POINTER(C.c_int32)()
        a = C.POINTER(C.c_double)()
        L.ref_mtx_to_csr_keep_symmetry(os.fsencode(path), C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym),
                                       C.byref(ia), C.byref(ja), C.byref(a))
        row_ptr = np.ctypeslib.as_array(ia, shape=(m.value + 1,)).copy()
        col_idx = np.ctypeslib.as_array(ja, shape=(max(nnz.value, 1),))[:nnz.value].copy()
        values = np.ctypeslib.as_array(a, shape=(max(nnz.value, 1),))[:nnz.value].copy()
        for p in (ia, ja, a):
            L.ref_free(p)
        return dict(m=m.value, n=n.value, nnz=nnz.value, symmetric=sym.value), row_ptr, col_idx, values

    def coo_to_csr(self, R, Cc, V, m, n):
        R = np.ascontiguousarray(R, np.int32)
        Cc = np.ascontiguousarray(Cc, np.int32)
        V = np.ascontiguousarray(V, np.float64)
        nnz = len(R)
        ia = np.zeros(m + 1, np.int32)
        ja = np.zeros(max(nnz, 1), np.int32)
        a = np.zeros(max(nnz, 1), np.float64)
        self.lib.ref_coo_to_csr(R.ctypes, Cc.ctypes, V.ctypes, C.c_long(m), C.c_long(n), C.c_long(nnz),
                                ia.ctypes, ja.ctypes, a.ctypes)
        return ia, ja[:nnz], a[:nnz]

    def csr_to_format(self, row_ptr, col_idx, values, m, n, symmetric_unexpanded=False):
        ia = np.ascontiguousarray(row_ptr, np.int32)
        ja = np.ascontiguousarray(col_idx, np.int32)
        a = np.ascontiguousarray(values, np.float64)
        self.m, self.n, self.nnz = m, n, len(ja)
        fn = self.lib.ref_csr_to_format_symmetric if symmetric_unexpanded else self.lib.ref_csr_to_format
        rc = fn(ia.ctypes, ja.ctypes, a.ctypes, C.c_long(m), C.c_long(n), C.c_long(len(ja)))
        assert rc == 0
        self.format_name = self.lib.ref_format_name().decode()
        self.mem_footprint = self.lib.ref_mem_footprint()
        self.csr_mem_footprint = self.lib.ref_csr_mem_footprint()

    def spmv(self, x):
        x = np.ascontiguousarray(x, self.dtype)
        y = np.ones(self.m + 64, self.dtype)     # driver canary: bench_spmv.cpp:606-609
        self.lib.ref_spmv(x.ctypes, y.ctypes)
        return y[:self.m].copy()

    def time_spmv(self, x, min_loops=64, min_runtime=2.0):
        x = np.ascontiguousarray(x, self.dtype)
        y = np.ones(self.m + 64, self.dtype)
        loops = C.c_long()
        tmin = C.c_double()
        tmax = C.c_double()
        med = self.lib.ref_time_spmv(x.ctypes, y.ctypes, C.c_long(min_loops), C.c_double(min_runtime),
                                     C.byref(loops), C.byref(tmin), C.byref(tmax))
        return dict(median=med, min=tmin.value, max=tmax.value, loops=loops.value)


def sellcs_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "v3", "libref_sellcs.so"))


def ref_sellcs_layout(row_ptr, col_idx, a, n_cols, C_rows, sigma):
    """The reference's own SELL-C-sigma FORMAT code (sell-C-s/RISC-V/sellcs_format.c, radix_sort.c, sellcs_utils.c compiled in
    place into oracle/_ref/v3/libref_sellcs.so) run on a CSR: dict(row_order, widths, slice_ptr, col, val)."""
    L = C.CDLL(os.path.join(_HERE, "_ref", "v3", "libref_sellcs.so"), mode=C.RTLD_LOCAL)
    L.ref_sellcs_convert.restype = C.c_long
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    a = np.ascontiguousarray(a, np.float64)
    m = len(row_ptr) - 1
    ns = (m + C_rows - 1) // C_rows
    order = np.zeros(max(m, 1), np.int32)
    widths = np.zeros(max(ns, 1), np.int32)
    sp = np.zeros(ns + 1, np.int64)
    cap = int(C_rows) * int(np.sort(np.diff(row_ptr))[::-1][:max(ns, 1)].sum()) + C_rows       # sum of the ns longest rows bounds it
    col = np.zeros(max(cap, 1), np.int32)
    val = np.zeros(max(cap, 1), np.float64)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    total = L.ref_sellcs_convert(C.c_int32(m), C.c_int32(n_cols), p(row_ptr), p(col_idx), p(a), C.c_long(C_rows), C.c_long(sigma),
                                 p(order), p(widths), p(sp), p(col), p(val), C.c_long(cap))
    assert total <= cap
    return dict(row_order=order[:m], widths=widths[:ns], slice_ptr=sp, col=col[:total], val=val[:total])
