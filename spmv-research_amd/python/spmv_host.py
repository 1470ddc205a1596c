"""ctypes binding of include/spmv_host.h (libspmv_host.so): Matrix-Market reader, COO->CSR, partitioners and the
synthetic matrix generators. Host logic only — no SpMV arithmetic."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libspmv_host.so")

SYMBOLS = [
    "spmv_host_last_error", "spmv_host_free", "spmv_host_mtx_read", "spmv_host_coo_free", "spmv_host_coo_to_csr",
    "spmv_host_mtx_write_csr", "spmv_host_partition_iterations", "spmv_host_partition_prefix_sums",
    "spmv_host_csr_free", "spmv_host_gen_twin", "spmv_host_gen_named", "spmv_host_gen_kkt", "spmv_host_csr_features",
    "spmv_host_gen_kkt_row_ptr", "spmv_host_gen_kkt_block", "spmv_host_remap_columns", "spmv_host_column_ranges",
    "spmv_host_bfs_order", "spmv_host_owners_from_order", "spmv_host_partition_volume", "spmv_host_partition_layout",
    "spmv_host_permuted_block", "spmv_host_halo_lists", "spmv_host_gen_kkt_rows", "spmv_host_jitter_columns",
    "spmv_host_kkt_bfs_owner", "spmv_host_kkt_partition_volume", "spmv_host_csr_am_stats", "spmv_host_gen_kkt_rows_into", "spmv_host_gen_kkt_rows_filtered",
]


class _Coo(C.Structure):
    _fields_ = [("m", C.c_long), ("n", C.c_long), ("nnz", C.c_long), ("nnz_sym", C.c_long),
                ("nnz_diag", C.c_long), ("nnz_non_diag", C.c_long),
                ("symmetric", C.c_int), ("skew", C.c_int), ("hermitian", C.c_int), ("pad_", C.c_int),
                ("field", C.c_char * 16),
                ("R", C.POINTER(C.c_int32)), ("C", C.POINTER(C.c_int32)), ("V", C.POINTER(C.c_double))]


class _Csr(C.Structure):
    _fields_ = [("m", C.c_long), ("n", C.c_long), ("nnz", C.c_long),
                ("row_ptr", C.POINTER(C.c_int32)), ("col_idx", C.POINTER(C.c_int32)), ("values", C.POINTER(C.c_double))]


_lib = None


class HostError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `make -C spmv-research_amd` (__graft_entry__.build())")
        _lib = C.CDLL(LIB_PATH)
        _lib.spmv_host_last_error.restype = C.c_char_p
    return _lib


def _check(rc):
    if rc != 0:
        raise HostError(lib().spmv_host_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mtx_read(path):
    coo = _Coo()
    _check(lib().spmv_host_mtx_read(os.fsencode(path), C.byref(coo)))
    nnz = coo.nnz
    R = np.ctypeslib.as_array(coo.R, shape=(max(nnz, 1),))[:nnz].copy()
    Cc = np.ctypeslib.as_array(coo.C, shape=(max(nnz, 1),))[:nnz].copy()
    V = np.ctypeslib.as_array(coo.V, shape=(max(nnz, 1),))[:nnz].copy()
    info = dict(m=coo.m, n=coo.n, nnz=nnz, nnz_sym=coo.nnz_sym, nnz_diag=coo.nnz_diag,
                nnz_non_diag=coo.nnz_non_diag, symmetric=coo.symmetric, skew=coo.skew,
                hermitian=coo.hermitian, field=coo.field.decode())
    lib().spmv_host_coo_free(C.byref(coo))
    return info, R, Cc, V


def coo_to_csr(R, Cc, V, m, n):
    R = np.ascontiguousarray(R, np.int32)
    Cc = np.ascontiguousarray(Cc, np.int32)
    V = np.ascontiguousarray(V, np.float64)
    nnz = len(R)
    row_ptr = np.zeros(m + 1, np.int32)
    col_idx = np.zeros(max(nnz, 1), np.int32)
    values = np.zeros(max(nnz, 1), np.float64)
    _check(lib().spmv_host_coo_to_csr(_p(R), _p(Cc), _p(V), C.c_long(m), C.c_long(n), C.c_long(nnz),
                                      _p(row_ptr), _p(col_idx), _p(values)))
    return row_ptr, col_idx[:nnz], values[:nnz]


def mtx_to_csr(path):
    info, R, Cc, V = mtx_read(path)
    return (info,) + coo_to_csr(R, Cc, V, info["m"], info["n"])


def mtx_write_csr(path, row_ptr, col_idx, values, m, n):
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    values = np.ascontiguousarray(values, np.float64)
    _check(lib().spmv_host_mtx_write_csr(os.fsencode(path), _p(row_ptr), _p(col_idx), _p(values), C.c_long(m), C.c_long(n)))


def partition_iterations(num_workers, worker_pos, start, end):
    s, e = C.c_long(), C.c_long()
    _check(lib().spmv_host_partition_iterations(C.c_long(num_workers), C.c_long(worker_pos), C.c_long(start),
                                                C.c_long(end), C.byref(s), C.byref(e)))
    return s.value, e.value


def partition_prefix_sums(num_workers, worker_pos, sums, N, total):
    sums = np.ascontiguousarray(sums, np.int32)
    s, e = C.c_long(), C.c_long()
    _check(lib().spmv_host_partition_prefix_sums(C.c_long(num_workers), C.c_long(worker_pos), _p(sums), C.c_long(N),
                                                 C.c_long(total), C.byref(s), C.byref(e)))
    return s.value, e.value


def _take_csr(csr):
    m, nnz = csr.m, csr.nnz
    # zero-copy views would dangle after free; matrices up to ~10 GB are copied once
    rp = np.ctypeslib.as_array(csr.row_ptr, shape=(m + 1,)).copy()
    ci = np.ctypeslib.as_array(csr.col_idx, shape=(max(nnz, 1),))[:nnz].copy()
    va = np.ctypeslib.as_array(csr.values, shape=(max(nnz, 1),))[:nnz].copy()
    n = csr.n
    lib().spmv_host_csr_free(C.byref(csr))
    return dict(m=m, n=n, nnz=nnz, row_ptr=rp, col_idx=ci, values=va)


def gen_twin(nr_rows, nr_cols, avg, std, bw_scaled, skew, neigh, crs, seed=14, pattern=False):
    csr = _Csr()
    _check(lib().spmv_host_gen_twin(C.c_long(nr_rows), C.c_long(nr_cols), C.c_double(avg), C.c_double(std),
                                    C.c_double(bw_scaled), C.c_double(skew), C.c_double(neigh), C.c_double(crs),
                                    C.c_ulong(seed), C.c_int(1 if pattern else 0), C.byref(csr)))
    return _take_csr(csr)


def gen_named(name, scale=1.0):
    csr = _Csr()
    _check(lib().spmv_host_gen_named(name.encode(), C.c_double(scale), C.byref(csr)))
    return _take_csr(csr)


def gen_kkt(N, seed=14):
    csr = _Csr()
    _check(lib().spmv_host_gen_kkt(C.c_long(N), C.c_ulong(seed), C.byref(csr)))
    return _take_csr(csr)


def kkt_size(N):
    m = C.c_long()
    _check(lib().spmv_host_gen_kkt_row_ptr(C.c_long(N), None, C.byref(m), None))
    return m.value


def gen_kkt_row_ptr(N):
    m = kkt_size(N)
    rp = np.zeros(m + 1, np.int32)
    nnz = C.c_long()
    _check(lib().spmv_host_gen_kkt_row_ptr(C.c_long(N), _p(rp), None, C.byref(nnz)))
    return rp


def gen_kkt_block(N, r0, r1, seed=14):
    csr = _Csr()
    _check(lib().spmv_host_gen_kkt_block(C.c_long(N), C.c_ulong(seed), C.c_long(r0), C.c_long(r1), C.byref(csr)))
    return _take_csr(csr)


def gen_kkt_rows(N, rows, seed=14):
    """The rows `rows` (ascending or not) of the KKT matrix as a local CSR with global column indices."""
    rows = np.ascontiguousarray(rows, np.int32)
    csr = _Csr()
    _check(lib().spmv_host_gen_kkt_rows(C.c_long(N), C.c_ulong(seed), _p(rows), C.c_long(len(rows)), C.byref(csr)))
    return _take_csr(csr)


def gen_kkt_rows_into(N, nnz, rows=None, r0=0, count=None, values=True, seed=14):
    """Rows of the KKT matrix written straight into numpy arrays of the right size (nnz = their total length, known from the global
    row_ptr): the CSR dict without the second copy gen_kkt_rows makes. values=False: structure only (values is None)."""
    if rows is not None:
        rows = np.ascontiguousarray(rows, np.int32)
        count = len(rows)
    rp = np.zeros(count + 1, np.int32)
    ci = np.empty(max(nnz, 1), np.int32)
    va = np.empty(max(nnz, 1), np.float64) if values else None
    _check(lib().spmv_host_gen_kkt_rows_into(C.c_long(N), C.c_ulong(seed), None if rows is None else _p(rows), C.c_long(r0), C.c_long(count),
                                             _p(rp), _p(ci), None if va is None else _p(va), C.c_long(nnz)))
    assert int(rp[count]) == nnz
    return dict(m=count, n=kkt_size(N), nnz=nnz, row_ptr=rp, col_idx=ci[:nnz], values=None if va is None else va[:nnz])


def gen_kkt_rows_filtered(N, col_lo, col_hi, keep_inside, rows=None, r0=0, count=None, values=True, seed=14):
    """Rows of the KKT matrix with only the columns inside (keep_inside) / outside [col_lo, col_hi): two calls — the filtered row
    lengths first, then the entries into arrays of exactly that size."""
    if rows is not None:
        rows = np.ascontiguousarray(rows, np.int32)
        count = len(rows)
    rp = np.zeros(count + 1, np.int32)
    args = (C.c_long(N), C.c_ulong(seed), None if rows is None else _p(rows), C.c_long(r0), C.c_long(count), C.c_long(col_lo), C.c_long(col_hi),
            C.c_int(1 if keep_inside else 0), _p(rp))
    _check(lib().spmv_host_gen_kkt_rows_filtered(*args, None, None, C.c_long(0)))
    nnz = int(rp[count])
    ci = np.empty(max(nnz, 1), np.int32)
    va = np.empty(max(nnz, 1), np.float64) if values else None
    _check(lib().spmv_host_gen_kkt_rows_filtered(*args, _p(ci), None if va is None else _p(va), C.c_long(nnz)))
    return dict(m=count, n=kkt_size(N), nnz=nnz, row_ptr=rp, col_idx=ci[:nnz], values=None if va is None else va[:nnz])


def jitter_columns(A, frac, span=3, seed=14):
    """In place on a CSR dict: perturb the off-diagonal columns of a fraction of the rows (bench.py --jitter)."""
    assert A["col_idx"].dtype == np.int32 and A["values"].dtype == np.float64 and A["row_ptr"].dtype == np.int32
    _check(lib().spmv_host_jitter_columns(C.c_long(A["m"]), C.c_long(A["n"]), _p(A["row_ptr"]), _p(A["col_idx"]), _p(A["values"]),
                                          C.c_double(frac), C.c_long(span), C.c_ulong(seed)))
    return A


def kkt_bfs_owner(N, parts):
    """owner[v] of the KKT matrix's breadth-first slab partition, computed without building the matrix."""
    owner = np.zeros(kkt_size(N), np.int32)
    _check(lib().spmv_host_kkt_bfs_owner(C.c_long(N), C.c_long(parts), _p(owner)))
    return owner


def kkt_partition_volume(N, owner, parts):
    vol = np.zeros(parts, np.int64)
    _check(lib().spmv_host_kkt_partition_volume(C.c_long(N), _p(_i32(owner)), C.c_long(parts), _p(vol)))
    return vol


def remap_columns(col_idx, offsets, padded):
    """In place: x is kept as len(offsets)-1 slices padded to `padded` entries."""
    assert col_idx.dtype == np.int32 and col_idx.flags.c_contiguous
    offsets = np.ascontiguousarray(offsets, np.int64)
    _check(lib().spmv_host_remap_columns(_p(col_idx), C.c_long(len(col_idx)), _p(offsets), C.c_long(len(offsets) - 1),
                                         C.c_long(padded)))
    return col_idx


def column_ranges(col_idx, padded, parts):
    assert col_idx.dtype == np.int32 and col_idx.flags.c_contiguous
    lo = np.zeros(parts, np.int64)
    hi = np.zeros(parts, np.int64)
    _check(lib().spmv_host_column_ranges(_p(col_idx), C.c_long(len(col_idx)), C.c_long(padded), C.c_long(parts), _p(lo), _p(hi)))
    hi = np.maximum(hi, lo * 0)
    lo = np.where(hi > lo, lo, 0)
    hi = np.where(hi > lo, hi, 0)
    return lo, hi


def _i32(a):
    a = np.ascontiguousarray(a)
    assert a.dtype == np.int32
    return a


def bfs_order(row_ptr, col_idx, m, n):
    """Vertices of the matrix graph in breadth-first order from a pseudo-peripheral vertex (graph_partition.cpp)."""
    order = np.zeros(m, np.int32)
    _check(lib().spmv_host_bfs_order(_p(_i32(row_ptr)), _p(_i32(col_idx)), C.c_long(m), C.c_long(n), _p(order)))
    return order


def owners_from_order(row_ptr, order, parts):
    owner = np.zeros(len(order), np.int32)
    _check(lib().spmv_host_owners_from_order(_p(_i32(row_ptr)), C.c_long(len(order)), _p(_i32(order)), C.c_long(parts), _p(owner)))
    return owner


def partition_volume(row_ptr, col_idx, owner, parts):
    """volume[p] = distinct x entries part p reads from other parts."""
    vol = np.zeros(parts, np.int64)
    _check(lib().spmv_host_partition_volume(_p(_i32(row_ptr)), _p(_i32(col_idx)), C.c_long(len(owner)), _p(_i32(owner)),
                                            C.c_long(parts), _p(vol)))
    return vol


def partition_layout(row_ptr, col_idx, owner, parts):
    """(perm new->old, offsets[parts+1]): boundary vertices first inside every part, original order inside a group."""
    perm = np.zeros(len(owner), np.int32)
    offsets = np.zeros(parts + 1, np.int64)
    _check(lib().spmv_host_partition_layout(_p(_i32(row_ptr)), _p(_i32(col_idx)), C.c_long(len(owner)), _p(_i32(owner)),
                                            C.c_long(parts), _p(perm), _p(offsets)))
    return perm, offsets


def permuted_block(row_ptr, col_idx, values, perm, inv, r0, r1):
    """Rows [r0,r1) of P A P^T as a local CSR (columns in the new numbering, ascending)."""
    values = np.ascontiguousarray(values, np.float64)
    csr = _Csr()
    assert 0 <= r0 <= r1 <= len(perm)                # perm may be just a row list (then inv still spans every column)
    _check(lib().spmv_host_permuted_block(_p(_i32(row_ptr)), _p(_i32(col_idx)), _p(values), C.c_long(len(inv)), _p(_i32(perm)),
                                          _p(_i32(inv)), C.c_long(r0), C.c_long(r1), C.byref(csr)))
    return _take_csr(csr)


def halo_lists(row_ptr, col_idx, owner, parts, rank):
    """(send, recv): per part q the ascending original vertex numbers `rank` sends to / receives from q."""
    so = np.zeros(parts + 1, np.int64)
    ro = np.zeros(parts + 1, np.int64)
    sl = C.POINTER(C.c_int32)()
    rl = C.POINTER(C.c_int32)()
    _check(lib().spmv_host_halo_lists(_p(_i32(row_ptr)), _p(_i32(col_idx)), C.c_long(len(owner)), _p(_i32(owner)), C.c_long(parts),
                                      C.c_long(rank), _p(so), C.byref(sl), _p(ro), C.byref(rl)))
    s = np.ctypeslib.as_array(sl, shape=(max(int(so[-1]), 1),))[:int(so[-1])].copy()
    r = np.ctypeslib.as_array(rl, shape=(max(int(ro[-1]), 1),))[:int(ro[-1])].copy()
    lib().spmv_host_free(sl)
    lib().spmv_host_free(rl)
    return ([s[so[q]:so[q + 1]] for q in range(parts)], [r[ro[q]:ro[q + 1]] for q in range(parts)])


FEATURES = ("avg_nnz_per_row", "std_nnz_per_row", "avg_bw_scaled", "skew", "avg_num_neighbours",
            "cross_row_similarity", "max_nnz_per_row")


def csr_features(row_ptr, col_idx, m, n):
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    out = np.zeros(7)
    _check(lib().spmv_host_csr_features(_p(row_ptr), _p(col_idx), C.c_long(m), C.c_long(n), _p(out)))
    return dict(zip(FEATURES, out.tolist()))


AM_STATS = ("density", "mem_footprint", "avg_nnz_per_row", "std_nnz_per_row", "avg_bw", "std_bw", "avg_bw_scaled", "std_bw_scaled",
            "avg_sc", "std_sc", "avg_sc_scaled", "std_sc_scaled", "skew", "avg_num_neighbours", "cross_row_similarity")


def csr_am_stats(row_ptr, col_idx, m, n):
    """The matrix statistics of the reference's artificial-matrix CSV row (bench_spmv.cpp:489-563)."""
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    out = np.zeros(15)
    buf = C.create_string_buffer(32)
    _check(lib().spmv_host_csr_am_stats(_p(row_ptr), _p(col_idx), C.c_long(m), C.c_long(n), _p(out), buf, C.c_long(32)))
    d = dict(zip(AM_STATS, out.tolist()))
    d["mem_range"] = buf.value.decode()
    return d
