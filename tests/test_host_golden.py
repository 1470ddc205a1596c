"""Host side of the product (libspmv_host.so: Matrix-Market reader, COO->CSR, partitioners, generators) against the
golden vectors of the genuine reference build and against the oracle. CPU only."""
import os

import numpy as np
import pytest

from conftest import CASES, GOLDEN, load_case


@pytest.fixture(scope="module")
def H():
    import spmv_host
    spmv_host.lib()
    return spmv_host


@pytest.mark.parametrize("case", CASES)
def test_reader_and_converter_bit_exact(H, case):
    info, g = load_case(case)
    hinfo, rp, ci, va = H.mtx_to_csr(os.path.join(GOLDEN, case + ".mtx"))
    for k in ("m", "n", "nnz", "symmetric", "nnz_diag", "nnz_non_diag"):
        assert hinfo[k] == info[k], k
    np.testing.assert_array_equal(rp, g["row_ptr"])
    np.testing.assert_array_equal(ci, g["col_idx"])
    np.testing.assert_array_equal(va, g["values"])


@pytest.mark.parametrize("case", CASES)
def test_reader_matches_oracle_coo_order(H, oracle, case):
    """Entry order of the expanded COO (file entries, then mirrored off-diagonals in file order) is part of the contract."""
    path = os.path.join(GOLDEN, case + ".mtx")
    hi, hr, hc, hv = H.mtx_read(path)
    oi, orr, oc, ov = oracle.mtx_read(path)
    assert {k: hi[k] for k in oi} == oi
    np.testing.assert_array_equal(hr, orr)
    np.testing.assert_array_equal(hc, oc)
    np.testing.assert_array_equal(hv, ov)


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_reader_error_behaviour(H, oracle, tmp_path):
    """Same accept/reject decisions as the reference loader (matrix_market.c:185-186,197-217,239-240)."""
    bad = {
        "count.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n",           # fewer lines than nnz
        "extra.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1 1.0\n2 2 1.0\n",   # more lines than nnz
        "sym.mtx": "%%MatrixMarket matrix coordinate real Symmetric\n3 3 1\n1 1 1.0\n",            # case-sensitive keyword
        "obj.mtx": "%%MatrixMarket vector coordinate real general\n3 3 1\n1 1 1.0\n",
        "field.mtx": "%%MatrixMarket matrix coordinate quaternion general\n3 3 1\n1 1 1.0\n",
        "array.mtx": "%%MatrixMarket matrix array real general\n1 1\n1.0\n",                       # not usable on the SpMV path
        "size.mtx": "%%MatrixMarket matrix coordinate real general\n3 3\n1 1 1.0\n",
    }
    for name, text in bad.items():
        p = _write(tmp_path, name, text)
        with pytest.raises(H.HostError):
            H.mtx_read(p)
        with pytest.raises(ValueError):
            oracle.mtx_read(p)
    with pytest.raises(H.HostError):
        H.mtx_read(str(tmp_path / "does_not_exist.mtx"))
    # accepted oddities: blank lines, comments after the banner, '+' signs, exponent forms, tabs, pattern+skew (= +1.0)
    ok = _write(tmp_path, "ok.mtx", "%%MatrixMarket matrix coordinate pattern skew-symmetric\n% c\n\n3 3 2\n2 1\n\n3\t2\n")
    hi, r, c, v = H.mtx_read(ok)
    oi, orr, oc, ov = oracle.mtx_read(ok)
    assert hi["nnz"] == 4 and np.all(v == 1.0)
    np.testing.assert_array_equal(v, ov)
    np.testing.assert_array_equal(r, orr)
    ok2 = _write(tmp_path, "ok2.mtx", "%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 +1.5e+0\n 2  2\t-2.5E-1\n1 2 .5\n")
    _, r, c, v = H.mtx_read(ok2)
    np.testing.assert_array_equal(v, [1.5, -0.25, 0.5])
    np.testing.assert_array_equal(v, oracle.mtx_read(ok2)[3])


def test_coo_to_csr_duplicates_kept_and_rejects_bad_indices(H, oracle):
    R = np.array([1, 0, 1, 1, 0], np.int32)
    C = np.array([2, 1, 0, 2, 1], np.int32)
    V = np.array([1., 2., 3., 4., 5.])
    rp, ci, va = H.coo_to_csr(R, C, V, 2, 3)
    np.testing.assert_array_equal(rp, [0, 2, 5])
    np.testing.assert_array_equal(ci, [1, 1, 0, 2, 2])
    np.testing.assert_array_equal(va, [2., 5., 3., 1., 4.])           # duplicates kept, input order
    orp, oci, ova = oracle.coo_to_csr(R, C, V, 2, 3)
    np.testing.assert_array_equal(rp, orp)
    np.testing.assert_array_equal(ci, oci)
    np.testing.assert_array_equal(va, ova)
    with pytest.raises(H.HostError):
        H.coo_to_csr(np.array([2], np.int32), np.array([0], np.int32), np.array([1.0]), 2, 3)


def test_coo_to_csr_random_large(H, oracle):
    rng = np.random.default_rng(14)
    m, n, nnz = 5000, 4000, 200000
    R = rng.integers(0, m, nnz).astype(np.int32)
    C = rng.integers(0, n, nnz).astype(np.int32)
    V = rng.uniform(-1, 1, nnz)
    R[:300] = 17                                  # one long row with many duplicate columns
    C[:300] = rng.integers(0, 20, 300)
    a = H.coo_to_csr(R, C, V, m, n)
    b = oracle.coo_to_csr(R, C, V, m, n)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)


def test_partitioners_match_oracle(H, oracle):
    rng = np.random.default_rng(5)
    for trial in range(200):
        m = int(rng.integers(1, 400))
        lens = rng.integers(0, 30, m)
        if trial % 5 == 0:
            lens[rng.integers(0, m)] = 5000
        rp = np.zeros(m + 1, np.int32)
        rp[1:] = np.cumsum(lens)
        W = int(rng.integers(1, 20))
        prev_e = 0
        for w in range(W):
            got = H.partition_prefix_sums(W, w, rp, m, int(rp[m]))
            assert got == oracle.partition_prefix_sums(W, w, rp, m, int(rp[m]))
            s, e = int(rng.integers(0, 50)), int(rng.integers(0, 500))
            if e >= s:
                assert H.partition_iterations(W, w, s, e) == oracle.partition_iterations(W, w, s, e)
        # contiguous cover of [0,m) in worker order (what the multi-GPU row split relies on)
        bounds = [H.partition_prefix_sums(W, w, rp, m, int(rp[m])) for w in range(W)]
        assert bounds[0][0] == 0 and bounds[-1][1] == m


def test_generators_are_deterministic_sorted_and_described(H):
    for name, scale in (("cant", 0.2), ("scircuit", 0.2), ("pwtk", 0.1), ("soc-LiveJournal1", 0.02), ("nlpkkt240", 0.0005)):
        A = H.gen_named(name, scale)
        B = H.gen_named(name, scale)
        for k in ("row_ptr", "col_idx", "values"):
            np.testing.assert_array_equal(A[k], B[k])
        rp, ci = A["row_ptr"], A["col_idx"]
        assert rp[0] == 0 and rp[-1] == A["nnz"] and np.all(np.diff(rp) >= 0)
        assert ci.min() >= 0 and ci.max() < A["n"]
        inner = np.ones(A["nnz"], bool)
        inner[rp[1:-1][rp[1:-1] < A["nnz"]]] = False
        inner[0] = False
        assert np.all(np.diff(ci)[inner[1:]] > 0), "columns strictly ascending inside every row"
        f = H.csr_features(rp, ci, A["m"], A["n"])
        assert abs(f["avg_nnz_per_row"] - A["nnz"] / A["m"]) < 1e-9
    # feature targets of the twins at full size (config.sh:402-449): mean within 2 %, std within 35 %
    A = H.gen_named("scircuit", 1.0)
    f = H.csr_features(A["row_ptr"], A["col_idx"], A["m"], A["n"])
    assert A["m"] == 170998 and abs(f["avg_nnz_per_row"] - 5.6078784547) / 5.6 < 0.02
    assert abs(f["std_nnz_per_row"] - 4.39) / 4.39 < 0.35 and abs(f["skew"] - 61.95) / 61.95 < 0.05
    A = H.gen_named("soc-LiveJournal1", 0.05)
    assert np.all(A["values"] == 1.0)            # pattern matrix


def test_kkt_structure_and_row_blocks(H):
    import scipy.sparse as sp
    N = 10
    A = H.gen_kkt(N)
    m = 2 * N ** 3 + 6 * N ** 2
    assert A["m"] == A["n"] == m == H.kkt_size(N)
    M = sp.csr_matrix((A["values"], A["col_idx"], A["row_ptr"]), shape=(m, m))
    assert abs(M - M.T).max() == 0.0                       # symmetric KKT matrix
    assert M[N ** 3:, N ** 3:].nnz == 0                    # zero (2,2) block
    assert 20 < A["nnz"] / m < 30
    rp = H.gen_kkt_row_ptr(N)
    np.testing.assert_array_equal(rp, A["row_ptr"])
    for r0, r1 in ((0, 17), (500, 1500), (m - 33, m), (100, 100)):
        B = H.gen_kkt_block(N, r0, r1)
        s, e = rp[r0], rp[r1]
        np.testing.assert_array_equal(B["row_ptr"], rp[r0:r1 + 1] - s)
        np.testing.assert_array_equal(B["col_idx"], A["col_idx"][s:e])
        np.testing.assert_array_equal(B["values"], A["values"][s:e])


def test_padded_column_layout(H):
    import spmv_dist as D
    A = H.gen_kkt(8)
    for world in (2, 3, 8):
        off = D.row_partition(A["row_ptr"], world)
        assert off[0] == 0 and off[-1] == A["m"]
        nnz_blocks = np.diff(A["row_ptr"][off])
        assert nnz_blocks.max() <= 1.3 * A["nnz"] / world           # nnz-balanced
        padded = D.padded_len(off)
        assert padded % 64 == 0 and padded >= np.diff(off).max()
        c = A["col_idx"].copy()
        D.to_padded_columns(c, off, padded)
        np.testing.assert_array_equal(D.padded_to_global(c, off, padded), A["col_idx"])
        x = np.arange(A["n"], dtype=np.float64)
        xp = D.scatter_x_padded(x, off, padded)
        np.testing.assert_array_equal(xp[c], x[A["col_idx"]])
    with pytest.raises(H.HostError):
        H.remap_columns(A["col_idx"].copy(), np.array([0, 10, A["m"]]), 64)   # slice longer than the padding


def test_mtx_writer_round_trip(H, tmp_path):
    A = H.gen_named("cant", 0.02)
    p = str(tmp_path / "w.mtx")
    H.mtx_write_csr(p, A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"])
    info, rp, ci, va = H.mtx_to_csr(p)
    np.testing.assert_array_equal(rp, A["row_ptr"])
    np.testing.assert_array_equal(ci, A["col_idx"])
    np.testing.assert_array_equal(va, A["values"])           # %.17g round-trips fp64 exactly


# ---- compressed / archived inputs (reference: lib/parallel_io.c:28-130 shells out to zstdmt / tar) ----------

def _zstd_compress(data, frames=1):
    import ctypes as C
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    out = b""
    step = -(-len(data) // frames)
    for i in range(0, len(data), step):
        part = data[i:i + step]
        buf = C.create_string_buffer(z.ZSTD_compressBound(len(part)))
        n = z.ZSTD_compress(buf, len(buf), part, len(part), 3)
        assert not z.ZSTD_isError(n)
        out += buf.raw[:n]
    return out


@pytest.mark.parametrize("case", ["general_real", "symmetric_real", "pattern_general"])
def test_reader_accepts_compressed_and_archived_inputs(case, tmp_path):
    import gzip
    import io
    import tarfile
    import spmv_host as H
    from conftest import GOLDEN
    src = os.path.join(GOLDEN, case + ".mtx")
    text = open(src, "rb").read()
    want = H.mtx_read(src)

    def same(path):
        got = H.mtx_read(str(path))
        assert got[0] == want[0], path
        for a, b in zip(got[1:], want[1:]):
            assert np.array_equal(a, b), path

    (tmp_path / "a.mtx.gz").write_bytes(gzip.compress(text))
    same(tmp_path / "a.mtx.gz")
    half = len(text) // 2                                           # two gzip members back to back (pigz / cat a.gz b.gz)
    (tmp_path / "b.mtx.gz").write_bytes(gzip.compress(text[:half]) + gzip.compress(text[half:]))
    same(tmp_path / "b.mtx.gz")
    (tmp_path / "c.mtx.zst").write_bytes(_zstd_compress(text))
    same(tmp_path / "c.mtx.zst")
    (tmp_path / "d.mtx.zst").write_bytes(_zstd_compress(text, frames=3))
    same(tmp_path / "d.mtx.zst")
    # SuiteSparse layout: <name>.tar.gz holding <name>/<name>.mtx (a directory entry first)
    bio = io.BytesIO()
    with tarfile.open(fileobj=bio, mode="w") as tf:
        d = tarfile.TarInfo("m")
        d.type = tarfile.DIRTYPE
        tf.addfile(d)
        ti = tarfile.TarInfo("m/m.mtx")
        ti.size = len(text)
        tf.addfile(ti, io.BytesIO(text))
    (tmp_path / "m.tar").write_bytes(bio.getvalue())
    same(tmp_path / "m.tar")
    (tmp_path / "m.tar.gz").write_bytes(gzip.compress(bio.getvalue()))
    same(tmp_path / "m.tar.gz")
    (tmp_path / "m.tgz").write_bytes(gzip.compress(bio.getvalue()))
    same(tmp_path / "m.tgz")
    # the reference sends .gz through zstdmt too, which sniffs the format: a gzip stream named .zst must still load
    (tmp_path / "e.mtx.zst").write_bytes(gzip.compress(text))
    same(tmp_path / "e.mtx.zst")


def test_tar_member_with_an_impossible_size_is_refused(tmp_path):
    """A tar header whose octal size field is astronomically large (a crafted or torn archive) must be an error, not a pointer
    that wraps past the end of the buffer (advisor finding, round 1)."""
    import io
    import tarfile
    import spmv_host as H
    from conftest import GOLDEN
    text = open(os.path.join(GOLDEN, "general_real.mtx"), "rb").read()
    bio = io.BytesIO()
    with tarfile.open(fileobj=bio, mode="w", format=tarfile.USTAR_FORMAT) as tf:
        ti = tarfile.TarInfo("m.mtx")
        ti.size = len(text)
        tf.addfile(ti, io.BytesIO(text))
    raw = bytearray(bio.getvalue())
    raw[124:136] = b"77777777777\0"                  # size = 8 GiB - 1 for a member of a few KB
    (tmp_path / "bad.tar").write_bytes(bytes(raw))
    with pytest.raises(H.HostError):
        H.mtx_read(str(tmp_path / "bad.tar"))
    raw[124:136] = b"7" * 12                         # no terminator: strtoull reads 12 octal digits (64 GiB)
    (tmp_path / "bad2.tar").write_bytes(bytes(raw))
    with pytest.raises(H.HostError):
        H.mtx_read(str(tmp_path / "bad2.tar"))


def test_reader_reports_corrupt_compressed_input(tmp_path):
    import gzip
    import spmv_host as H
    from conftest import GOLDEN
    text = open(os.path.join(GOLDEN, "general_real.mtx"), "rb").read()
    z = gzip.compress(text)
    (tmp_path / "t.mtx.gz").write_bytes(z[:len(z) // 2])
    with pytest.raises(Exception, match="corrupt or truncated"):
        H.mtx_read(str(tmp_path / "t.mtx.gz"))
    (tmp_path / "e.tar").write_bytes(b"\0" * 1024)
    with pytest.raises(Exception, match="no regular file"):
        H.mtx_read(str(tmp_path / "e.tar"))


def test_reader_keeps_the_file_entries_first_for_keep_symmetry_callers():
    """KEEP_SYMMETRY builds (bench.cpp:131-136,186-192) pass only the file's own entries on: they are the first nnz_sym
    entries of the expanded COO. coo_to_csr of that prefix must equal the reference's un-expanded CSR."""
    import spmv_host as H
    n_checked = 0
    for case in CASES:
        info, z = load_case(case)
        if "sym_row_ptr" not in z:
            continue
        coo, R, Cc, V = H.mtx_read(os.path.join(GOLDEN, case + ".mtx"))
        k = coo["nnz_sym"]
        assert k == info["sym_nnz"]
        rp, ci, a = H.coo_to_csr(R[:k], Cc[:k], V[:k], coo["m"], coo["n"])
        assert np.array_equal(rp, z["sym_row_ptr"]) and np.array_equal(ci, z["sym_col_idx"]) and np.array_equal(a, z["sym_values"])
        n_checked += 1
    assert n_checked >= 5


def test_artificial_matrix_statistics():
    """The statistics of the artificial-matrix CSV row (bench_spmv.cpp:532-554) on a matrix small enough to do by hand, with the
    definitions the tree still states (csr_util_gen.c:437-449: bandwidth = col_max - col_min, scatter = degree / bandwidth)."""
    import spmv_host as H
    rp = np.array([0, 3, 3, 5, 6], np.int32)
    ci = np.array([0, 2, 6, 1, 2, 7], np.int32)           # rows: {0,2,6}, {}, {1,2}, {7}
    s = H.csr_am_stats(rp, ci, 4, 8)
    bw = np.array([6.0, 0.0, 1.0, 0.0])
    sc = np.array([3 / 6, 0.0, 2 / 1, 0.0])
    assert s["density"] == pytest.approx(6 / 32 * 100) and s["mem_footprint"] == pytest.approx((96 * 6 + 32 * 5) / (8 * 1024 * 1024))
    assert s["avg_nnz_per_row"] == pytest.approx(1.5) and s["std_nnz_per_row"] == pytest.approx(np.std([3, 0, 2, 1]))
    assert s["avg_bw"] == pytest.approx(bw.mean()) and s["std_bw"] == pytest.approx(bw.std())
    assert s["avg_bw_scaled"] == pytest.approx(bw.mean() / 8) and s["avg_sc"] == pytest.approx(sc.mean()) and s["std_sc"] == pytest.approx(sc.std())
    assert s["skew"] == pytest.approx((3 - 1.5) / 1.5) and s["mem_range"] == "[-]"            # far below the 4 MiB class
    A = H.gen_named("cant", 1.0)
    t = H.csr_am_stats(A["row_ptr"], A["col_idx"], A["m"], A["n"])
    f = H.csr_features(A["row_ptr"], A["col_idx"], A["m"], A["n"])
    assert t["mem_range"] == "[32-64]" and t["avg_bw_scaled"] == pytest.approx(f["avg_bw_scaled"]) and t["skew"] == pytest.approx(f["skew"])
