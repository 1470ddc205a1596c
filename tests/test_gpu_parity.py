"""GPU parity tests proper (-m gpu): every HIP kernel, through the C ABI (ctypes -> libspmv_mi355x.so), against
the CPU oracle and the committed golden vectors of the genuine reference build.

Bars (BASELINE.json north_star):
  * index order: y[i] belongs to global row i, every row written (empty rows = 0, driver canary 1.0 overwritten);
  * csr_scalar and SELL with C = 64 walk a row left to right with one FMA per element -> BIT-EXACT vs the reference
    CPU kernel (csr.cpp:334-350) in fp64 and fp32;
  * the reordering kernels (csr_vector, csr_merge, coo, SELL C<64): |y - y_ref| <= tol * sum_j |a_ij x_j| with
    tol = 1e-12 (fp64) / 1e-5 (fp32), and <= tol relative to |y_ref| on rows without cancellation.
"""
import numpy as np
import pytest

from conftest import CASES, MANIFEST, load_case

pytestmark = pytest.mark.gpu

TOL = {np.float64: 1e-12, np.float32: 1e-5}

VARIANTS = [
    ("csr_scalar", {}, True),
    ("csr_vector", {"lanes_per_row": 2}, False),
    ("csr_vector", {"lanes_per_row": 4}, False),
    ("csr_vector", {"lanes_per_row": 8}, False),
    ("csr_vector", {"lanes_per_row": 16}, False),
    ("csr_vector", {"lanes_per_row": 32}, False),
    ("csr_vector", {"lanes_per_row": 64}, False),      # the literal "one wavefront per row" of config 2
    ("csr_vector", {}, False),                          # auto
    ("csr_vector", {"lanes_per_row": 16, "rows_per_group": 4}, False),   # several rows of a lane group in flight
    ("csr_vector", {"lanes_per_row": 8, "rows_per_group": 2}, False),
    ("csr_vector", {"lanes_per_row": 8, "rows_per_group": 4}, False),
    ("csr_vector", {"lanes_per_row": 64, "rows_per_group": 2}, False),
    ("csr_stream", {}, False),
    ("csr_stream", {"lanes_per_row": 64}, False),                         # LDS-DMA, lane per row (exact only while the block fits the strip)
    ("csr_stream", {"lanes_per_row": 32}, False),
    ("csr_stream", {"lanes_per_row": 16}, False),
    ("csr_stream", {"lanes_per_row": 8}, False),
    ("csr_stream", {"lanes_per_row": 4}, False),
    ("csr_stream", {"stream_mode": 2, "lanes_per_row": 64}, True),        # register-staged (val,col) in LDS
    ("csr_stream", {"stream_mode": 2, "lanes_per_row": 32}, False),
    ("csr_stream", {"stream_mode": 2, "lanes_per_row": 8}, False),
    ("csr_stream", {"stream_mode": 1, "lanes_per_row": 4}, False),        # products in LDS
    ("csr_stream", {"stream_mode": 1, "lanes_per_row": 16}, False),
    ("csr_stream", {"stream_mode": 1, "lanes_per_row": 64}, False),
    ("csr_stream", {"stream_mode": 4}, False),                            # x window of the row block in LDS
    ("csr_stream", {"stream_mode": 4, "lanes_per_row": 8, "merge_items": 1}, False),
    ("csr_stream", {"stream_mode": 4, "lanes_per_row": 32, "merge_items": 3}, False),
    ("csr_stream", {"stream_mode": 4, "lanes_per_row": 64}, False),
    ("csr_merge", {}, False),
    ("csr_merge", {"merge_items": 5}, False),
    ("csr_merge", {"merge_items": 13}, False),
    ("sell_c_sigma", {"sell_c": 64, "sell_split": 1}, True),             # delta-compressed indices, one wave per slice
    ("sell_c_sigma", {"sell_c": 64, "sell_sigma": 64, "sell_split": 1}, True),
    ("sell_c_sigma", {}, False),                                           # auto: C = 64, delta, waves per slice by size
    ("sell_c_sigma", {"sell_split": 2}, False),
    ("sell_c_sigma", {"sell_split": 4, "sell_sigma": 128}, False),
    ("sell_c_sigma", {"sell_c": 64, "sell_delta": 2}, True),             # plain int32 indices
    ("sell_c_sigma", {"sell_c": 64, "sell_delta": 1, "sell_sigma": 1024, "sell_split": 1}, True),
    ("sell_c_sigma", {"sell_c": 32, "sell_sigma": 256}, False),
    ("sell_c_sigma", {"sell_c": 16, "sell_sigma": 16384}, False),
    ("sell_c_sigma", {"sell_c": 256}, True),                             # the BSC library's shape: C = 256, sigma = 16384 (sell_c_s.cpp:58-60)
    ("sell_c_sigma", {"sell_c": 256, "sell_sigma": 512}, True),
    ("sell_c_sigma", {"sell_window": 1, "sell_split": 1}, True),          # x window of a slice group in LDS, 16-bit indices
    ("sell_c_sigma", {"sell_window": 1, "sell_split": 1, "sell_group": 16}, True),
    ("sell_c_sigma", {"sell_window": 1, "sell_split": 2, "sell_group": 4}, False),
    ("sell_c_sigma", {"sell_window": 1, "sell_split": 4, "sell_group": 1}, False),
    ("sell_c_sigma", {"sell_window": 1}, False),
    ("coo", {}, False),
    ("coo", {"merge_items": 2}, False),
    ("coo", {"merge_items": 8}, False),
    ("coo", {"col_blocks": -1}, False),                  # column-blocked COO: segments of rows in LDS, entries by column block
    ("coo", {"col_blocks": 3}, False),
    ("coo", {"col_blocks": 64}, False),
    ("csr_merge", {"col_blocks": -1}, False),           # the same layout with merge-path-balanced row ranges
    ("csr_merge", {"col_blocks": 5}, False),
    ("csr_merge", {"col_blocks": -2}, False),           # forced CSR-order merge path
]
IDS = [f"{f}-{'-'.join(f'{k}{v}' for k, v in o.items()) or 'default'}" for f, o, _ in VARIANTS]


@pytest.fixture(scope="module")
def eng():
    import spmv_mi355x as eng
    assert eng.device_count() >= 1, "no GPU visible: the -m gpu tests need an MI355X"
    return eng


def check(y, y_ref, absrow, dtype, exact, what):
    assert y.shape == y_ref.shape
    assert np.all(np.isfinite(y)), what
    if exact:
        np.testing.assert_array_equal(y, y_ref, err_msg=what)
        return
    tol = TOL[dtype]
    err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
    bound = tol * np.maximum(absrow, np.finfo(np.float64).tiny)
    bad = np.nonzero(err > bound)[0]
    assert bad.size == 0, f"{what}: {bad.size} rows beyond {tol}*sum|a x|; worst row {bad[:5]} err {err[bad[:5]]}"
    well = np.abs(y_ref) > 0.1 * absrow          # rows without cancellation: plain relative error must hold too
    if well.any():
        rel = err[well] / np.abs(y_ref[well].astype(np.float64))
        assert rel.max() <= tol, f"{what}: max relative error {rel.max()}"


@pytest.mark.parametrize("variant", VARIANTS, ids=IDS)
@pytest.mark.parametrize("case", CASES)
def test_golden_cases(eng, oracle, case, variant):
    fmt, opts, exact = variant
    info, g = load_case(case)
    rp, ci, a = g["row_ptr"], g["col_idx"], g["values"]
    m, n = info["m"], info["n"]
    absrow = oracle.csr_spmv(rp, ci, np.abs(a), np.abs(g["x_rand"]))
    absrow1 = oracle.csr_spmv(rp, ci, np.abs(a), np.ones(n))
    for dtype, pk in ((np.float64, "d"), (np.float32, "f")):
        A = eng.Matrix(rp, ci, a, m, n, fmt, dtype, **opts)
        assert A.m == m and A.n == n and A.nnz == info["nnz"]
        for xn, x, ar in (("ones", np.ones(n), absrow1), ("rand", g["x_rand"], absrow)):
            y = A.spmv(x)
            # golden vector from the genuine reference CPU kernel, and the oracle's restatement of it
            check(y, g[f"y_csr_{pk}_{xn}"], ar, dtype, exact, f"{case}/{fmt}{opts}/{pk}/{xn} vs golden")
            check(y, oracle.csr_spmv(rp, ci, a, x, dtype), ar, dtype, exact, f"{case}/{fmt}{opts}/{pk}/{xn} vs oracle")
        A.close()


def synth(rng, m, n, kind):
    """Seeded synthetic CSR covering the reference-tested edge cases at sizes the oracle finishes in seconds."""
    if kind == "powerlaw":        # soc-LiveJournal1-like: mean ~14, a few rows in the tens of thousands
        lens = np.minimum((rng.pareto(1.3, m) * 4).astype(np.int64), n)
        lens[rng.integers(0, m, 3)] = min(n, 40000)
    elif kind == "short":         # scircuit-like: mean 5.6 with many empty rows
        lens = rng.poisson(5.6, m)
        lens[rng.random(m) < 0.2] = 0
    elif kind == "regular":       # nlpkkt/pwtk-like: ~27 per row, tiny variance
        lens = np.clip(rng.normal(27.7, 2, m).round().astype(np.int64), 0, n)
    elif kind == "one_row":       # a single row holding everything
        lens = np.zeros(m, np.int64)
        lens[m // 2] = n
    elif kind == "empty":
        lens = np.zeros(m, np.int64)
    else:
        raise ValueError(kind)
    lens = np.minimum(lens, n)
    rp = np.zeros(m + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    nnz = int(rp[-1])
    ci = np.empty(nnz, np.int32)
    for i in np.nonzero(lens)[0]:
        L = lens[i]
        if L > n // 2:
            ci[rp[i]:rp[i + 1]] = np.sort(rng.permutation(n)[:L])
        else:
            c = np.unique(rng.integers(0, n, int(L * 1.3) + 8))
            while len(c) < L:
                c = np.unique(np.concatenate([c, rng.integers(0, n, L)]))
            ci[rp[i]:rp[i + 1]] = np.sort(rng.permutation(c)[:L])
    a = rng.uniform(-1, 1, nnz)
    return rp.astype(np.int32), ci, a


SYNTH = [("powerlaw", 60000, 60000), ("short", 100000, 100000), ("regular", 50000, 50000),
         ("one_row", 1000, 70000), ("empty", 777, 555), ("regular", 4099, 257)]


@pytest.mark.parametrize("kind,m,n", SYNTH, ids=[f"{k}-{m}x{n}" for k, m, n in SYNTH])
def test_synthetic_all_formats(eng, oracle, kind, m, n):
    rng = np.random.default_rng(MANIFEST["seed"])
    rp, ci, a = synth(rng, m, n, kind)
    x = rng.uniform(-1, 1, n)
    for dtype in (np.float64, np.float32):
        y_ref = oracle.csr_spmv(rp, ci, a, x, dtype, num_threads=4)
        absrow = oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x))
        for fmt, opts, exact in VARIANTS:
            try:
                A = eng.Matrix(rp, ci, a, m, n, fmt, dtype, **opts)
            except eng.SpmvError as e:
                # a FORCED LDS-window layout refuses matrices whose slice groups span more than 65 536 columns or 128 KiB of x
                assert opts.get("sell_window") == 1 and "sell_window" in str(e) and n * np.dtype(dtype).itemsize > 128 * 1024, str(e)
                continue
            y = A.spmv(x)
            check(y, y_ref, absrow, dtype, exact, f"{kind}/{fmt}{opts}/{np.dtype(dtype).name}")
            # x changes between calls (CG/BiCG callers): always_copy path must pick the new vector up
            y2 = A.spmv(2 * x)
            check(y2, (2 * y_ref).astype(dtype), 2 * absrow, dtype, exact, f"{kind}/{fmt}{opts} second x")
            A.close()


def test_kahan_variant_is_bit_identical_to_the_reference_kahan_build(eng, oracle):
    """Row a3': the CUSTOM_KAHAN variant of the reference CPU kernel (csr.cpp:353-373) on the device: against the golden vectors of
    the genuine reference build (y_csr_kahan_d_*) and the oracle's restatement, bit for bit."""
    for case in CASES:
        info, g = load_case(case)
        rp, ci, a = g["row_ptr"], g["col_idx"], g["values"]
        A = eng.Matrix(rp, ci, a, info["m"], info["n"], "csr_scalar", np.float64, kahan=1)
        assert A.format_name == "MI355X_CSR_SCALAR_KAHAN_d"
        for xn, x in (("ones", np.ones(info["n"])), ("rand", g["x_rand"])):
            y = A.spmv(x)
            if f"y_csr_kahan_d_{xn}" in g:
                np.testing.assert_array_equal(y, g[f"y_csr_kahan_d_{xn}"], err_msg=f"{case}/{xn} vs the reference build")
            np.testing.assert_array_equal(y, oracle.csr_kahan_spmv(rp, ci, a, x), err_msg=f"{case}/{xn} vs the oracle")
        A.close()
    rng = np.random.default_rng(21)
    rp, ci, a = synth(rng, 20000, 20000, "powerlaw")
    x = rng.uniform(-1, 1, 20000)
    A = eng.Matrix(rp, ci, a, 20000, 20000, "csr_scalar", np.float64, kahan=1)
    np.testing.assert_array_equal(A.spmv(x), oracle.csr_kahan_spmv(rp, ci, a, x))
    A.close()


def test_reference_caching_semantics(eng, oracle):
    """GPU backends of the reference upload x once and download y once (csr_rocm_vector.cpp:224-257)."""
    info, g = load_case("general_real")
    rp, ci, a = g["row_ptr"], g["col_idx"], g["values"]
    A = eng.Matrix(rp, ci, a, info["m"], info["n"], "csr_vector")
    A.set_always_copy(False)
    x = np.ones(info["n"])
    y = np.ones(info["m"] + 64)
    A.spmv_raw(x, y)
    np.testing.assert_allclose(y[:info["m"]], g["y_csr_d_ones"], rtol=1e-12, atol=1e-13)
    assert np.all(y[info["m"]:] == 1.0)            # slack untouched
    y[:] = 7.0
    A.spmv_raw(x, y)                               # same pointers: no download on later calls
    assert np.all(y == 7.0)
    np.testing.assert_allclose(A.download_y(), g["y_csr_d_ones"], rtol=1e-12, atol=1e-13)
    # a NEW x (another host buffer) must give a new y in the caller's buffer, without always_copy
    x2 = np.full(info["n"], 2.0)
    A.spmv_raw(x2, y)
    np.testing.assert_allclose(y[:info["m"]], 2 * g["y_csr_d_ones"], rtol=1e-12, atol=1e-13)
    A.close()


@pytest.mark.parametrize("fmt", ["coo", "csr_merge"])
def test_blocked_layout_with_several_passes_and_split_rows(eng, oracle, fmt, monkeypatch):
    """The column-blocked layout's rarely taken paths on small inputs: a workgroup's LDS holds only 64 + 64 rows (so a few
    hundred rows already need several passes of 8 x 32 workgroups) and every row of 8 or more entries is split over the 32
    workgroups of its range and recombined by the carry fix-up."""
    monkeypatch.setenv("SPMV_MI355X_COOB_ROWS", "64")
    monkeypatch.setenv("SPMV_MI355X_COOB_LONG_MIN", "8")
    rng = np.random.default_rng(11)
    cases = [load_case(c)[1] for c in ("huge_row", "empty_rows_formats", "rectangular", "pattern_general")]
    cases = [(g["row_ptr"], g["col_idx"], g["values"], len(g["row_ptr"]) - 1, len(g["x_rand"])) for g in cases]
    rp, ci, a = synth(rng, 70000, 70000, "powerlaw")
    cases.append((rp, ci, a, 70000, 70000))
    for rp, ci, a, m, n in cases:
        x = rng.uniform(-1, 1, n)
        absrow = oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x))
        for dtype in (np.float64, np.float32):
            y_ref = oracle.csr_spmv(rp, ci, a, x, dtype)
            for cb in (-1, 7):
                A = eng.Matrix(rp, ci, a, m, n, fmt, dtype, col_blocks=cb)
                assert ("COOB" if fmt == "coo" else "MERGEB") in A.format_name
                if m >= 70000:
                    assert "_split" in A.format_name and int(A.format_name.split("_r")[1].split("_")[0]) > 8
                check(A.spmv(x), y_ref, absrow, dtype, False, f"{A.format_name} m={m}")
                A.close()


@pytest.mark.parametrize("name", ["powerlaw", "short", "regular_ragged", "one_row", "empty"])
def test_device_plain_sell_builder_equals_host_builder(eng, oracle, name):
    """The plain column-major SELL-C-sigma layout (C = 16 / 32 / 64 / 256: the layout the parity entry points a6' / a7 compare with the
    reference's) built on the GPU (convert_sell.hip: sell_plain_convert_device) holds the same bytes as the host builder's."""
    rng = np.random.default_rng(MANIFEST["seed"] + 7)
    if name == "regular_ragged":
        m, n = 4099, 257
        rp, ci, a = synth(rng, m, n, "regular")
    elif name == "one_row":
        m, n = 1000, 70000
    elif name == "empty":
        m, n = 777, 555
    else:
        m = n = 30000
    if name != "regular_ragged":
        rp, ci, a = synth(rng, m, n, name)
    x = rng.uniform(-1, 1, n)
    for dtype in (np.float64, np.float32):
        for C_rows, sigma in ((16, 16), (32, 1024), (64, 64), (64, 16384), (256, 16384)):
            opts = dict(sell_c=C_rows, sell_sigma=sigma, sell_delta=2, sell_window=2)
            H = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, convert_on=2, **opts)
            D = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, convert_on=1, **opts)
            assert D.format_name == H.format_name and D.mem_footprint == H.mem_footprint and "SELLD" not in H.format_name
            for k in ("slice_ptr", "row_of_sorted", "col", "val"):
                np.testing.assert_array_equal(D.stored_array(k), H.stored_array(k), err_msg=f"{name} C={C_rows} sigma={sigma} {k}")
            np.testing.assert_array_equal(D.spmv(x), H.spmv(x))
            H.close()
            D.close()


BLOCKED_ARRAYS = ("entries", "val", "batch_base", "batch_ptr", "chunk_ptr", "chunk_row", "wg_rows", "range_row", "range_long", "long_row")


@pytest.mark.parametrize("small_lds", [False, True])
def test_device_blocked_builder_equals_host_builder(eng, oracle, monkeypatch, small_lds):
    """The entry arrays of the column-blocked layout (kernels_coo.hip) built on the GPU (convert_coo.hip: keys, one stable radix sort, a
    wave per workgroup cutting the sorted entries into groups) hold the same bytes as the host builder's (build_coo.hip), in every
    stored array: power-law, split rows, several passes, pattern (unit) values, a column-block width, empty rows, one huge row."""
    if small_lds:
        monkeypatch.setenv("SPMV_MI355X_COOB_ROWS", "64")
        monkeypatch.setenv("SPMV_MI355X_COOB_LONG_MIN", "8")
    rng = np.random.default_rng(12)
    cases = [load_case(c)[1] for c in ("huge_row", "empty_rows_formats", "rectangular", "pattern_general")]
    cases = [(g["row_ptr"], g["col_idx"], g["values"], len(g["row_ptr"]) - 1, len(g["x_rand"])) for g in cases]
    rp, ci, a = synth(rng, 70000, 70000, "powerlaw")
    cases.append((rp, ci, a, 70000, 70000))
    cases.append((rp, ci, np.ones_like(a), 70000, 70000))                      # unit values: no value array, 8 entries per lane and batch
    rp, ci, a = synth(rng, 3000, 200000, "one_row")
    cases.append((rp, ci, a, 3000, 200000))
    for rp, ci, a, m, n in cases:
        x = rng.uniform(-1, 1, n)
        for fmt, dtype, cb in (("coo", np.float64, -1), ("csr_merge", np.float32, -1), ("coo", np.float32, 7)):
            H = eng.Matrix(rp, ci, a, m, n, fmt, dtype, col_blocks=cb, convert_on=2)
            D = eng.Matrix(rp, ci, a, m, n, fmt, dtype, col_blocks=cb, convert_on=1)
            assert D.format_name == H.format_name and D.mem_footprint == H.mem_footprint and ("COOB" in H.format_name or "MERGEB" in H.format_name)
            for k in BLOCKED_ARRAYS:
                np.testing.assert_array_equal(D.stored_array(k), H.stored_array(k), err_msg=f"{H.format_name} m={m} {k}")
            # same bytes, same kernel; the kernel's LDS atomics add in no fixed order, so two launches agree to rounding only
            absrow = oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x))
            check(D.spmv(x), oracle.csr_spmv(rp, ci, a, x, dtype), absrow, dtype, False, f"{D.format_name} m={m} (GPU-built)")
            H.close()
            D.close()


def test_beta_accumulate_and_row_blocks(eng, oracle):
    """Row-partitioned use: row blocks reproduce the global y in index order; the local/remote column
    split y = A_loc x + A_rem x (beta = 1) matches the unsplit product."""
    import torch
    rng = np.random.default_rng(3)
    rp, ci, a = synth(rng, 30000, 30000, "regular")
    m = n = 30000
    x = rng.uniform(-1, 1, n)
    y_ref = oracle.csr_spmv(rp, ci, a, x)
    absrow = oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x))
    xd = torch.from_numpy(x).cuda()
    for fmt in ("csr_scalar", "csr_vector", "csr_stream", "csr_merge", "sell_c_sigma", "coo"):
        parts = 3
        y_all = []
        for p in range(parts):
            r0, r1 = oracle.partition_prefix_sums(parts, p, rp, m, int(rp[m]))
            c0, c1 = r0, r1
            loc = eng.Matrix(rp, ci, a, m, n, fmt, row_begin=r0, row_end=r1, col_begin=c0, col_end=c1, col_filter_mode=1)
            rem = eng.Matrix(rp, ci, a, m, n, fmt, row_begin=r0, row_end=r1, col_begin=c0, col_end=c1, col_filter_mode=2)
            assert loc.m == r1 - r0 and loc.nnz + rem.nnz == int(rp[r1] - rp[r0])
            yd = torch.full((r1 - r0 + 64,), 1.0, dtype=torch.float64, device="cuda")
            s = torch.cuda.current_stream().cuda_stream
            loc.spmv_device(xd.data_ptr(), yd.data_ptr(), 0, s)
            rem.spmv_device(xd.data_ptr(), yd.data_ptr(), 1, s)
            torch.cuda.synchronize()
            y_all.append(yd[:r1 - r0].cpu().numpy())
            loc.close()
            rem.close()
        y = np.concatenate(y_all)
        check(y, y_ref, absrow, np.float64, False, f"rowblocks/{fmt}")


def test_formats_report_footprint(eng):
    info, g = load_case("banded_symmetric")
    rp, ci, a = g["row_ptr"], g["col_idx"], g["values"]
    m, n, nnz = info["m"], info["n"], info["nnz"]
    csr = nnz * 12 + (m + 1) * 4
    A = eng.Matrix(rp, ci, a, m, n, "csr_vector")
    assert A.mem_footprint == csr == A.csr_mem_footprint == info["csr_mem_footprint_d"]
    A.close()
    A = eng.Matrix(rp, ci, a, m, n, "coo")
    assert A.mem_footprint == nnz * 16          # mkl_coo.cpp:65
    A.close()
    A = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", np.float32, sell_c=64, sell_delta=2)
    lay = A.sell_layout()
    assert A.mem_footprint == (lay["num_slices"] + 1) * 8 + lay["nnz_ext"] * 8 + m * 4
    assert lay["nnz_ext"] >= nnz and sorted(lay["row_of_sorted"].tolist()) == list(range(m))
    plain_fp = A.mem_footprint
    A.close()
    # delta-compressed indices: same rows / values / (decoded) columns, smaller footprint on a banded matrix
    B = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", np.float32, sell_c=64, sell_delta=1)
    layd = B.sell_layout()
    assert B.mem_footprint < plain_fp
    np.testing.assert_array_equal(layd["row_of_sorted"], lay["row_of_sorted"])
    # widths are padded to 4 steps in the compressed layout: compare entry by entry through the slice pointers
    for s in range(lay["num_slices"]):
        w0 = (lay["slice_ptr"][s + 1] - lay["slice_ptr"][s]) // 64
        w1 = (layd["slice_ptr"][s + 1] - layd["slice_ptr"][s]) // 64
        assert w1 >= w0 and w1 - w0 < 4
        c0 = lay["col"][lay["slice_ptr"][s]:lay["slice_ptr"][s + 1]].reshape(w0, 64)
        v0 = lay["val"][lay["slice_ptr"][s]:lay["slice_ptr"][s + 1]].reshape(w0, 64)
        c1 = layd["col"][layd["slice_ptr"][s]:layd["slice_ptr"][s + 1]].reshape(w1, 64)
        v1 = layd["val"][layd["slice_ptr"][s]:layd["slice_ptr"][s + 1]].reshape(w1, 64)
        np.testing.assert_array_equal(v1[:w0], v0)
        real = v0 != 0
        np.testing.assert_array_equal(c1[:w0][real], c0[real])      # padding columns may differ, real entries may not
        assert np.all(v1[w0:] == 0)
    B.close()


def test_create_rejects_bad_input(eng):
    rp = np.array([0, 1, 2], np.int32)
    ci = np.array([0, 5], np.int32)
    a = np.ones(2)
    with pytest.raises(eng.SpmvError):
        eng.Matrix(rp, ci, a, 2, 2, "csr_vector")               # column out of range
    with pytest.raises(eng.SpmvError):
        eng.Matrix(rp, np.array([0, 1], np.int32), a, 2, 2, "csr_vector", lanes_per_row=3)
    with pytest.raises(eng.SpmvError):
        eng.Matrix(rp, np.array([0, 1], np.int32), a, 2, 2, "sell_c_sigma", sell_c=48)


# ---- GPU-side CSR -> SELL-64-sigma-delta conversion (csrc/convert_sell.hip, §8 row f2) against the host builder -------------

def _banded(rng, m, bw, per_row):
    """columns within +-bw of the diagonal: 8-bit deltas when bw is small, 16-bit when it is a few thousand"""
    rp = np.arange(m + 1, dtype=np.int64) * per_row
    ci = np.empty(m * per_row, np.int32)
    for i in range(m):
        lo, hi = max(0, i - bw), min(m, i + bw + 1)
        ci[i * per_row:(i + 1) * per_row] = np.sort(rng.choice(np.arange(lo, hi), per_row, replace=False))
    return rp.astype(np.int32), ci, rng.uniform(-1, 1, m * per_row)


CONVERT_CASES = ["banded8", "banded16", "powerlaw", "short", "regular_ragged", "one_row", "empty"]


@pytest.mark.parametrize("name", CONVERT_CASES)
def test_device_conversion_equals_host_conversion(eng, oracle, name):
    rng = np.random.default_rng(MANIFEST["seed"] + 5)
    if name == "banded8":
        m = n = 5000
        rp, ci, a = _banded(rng, m, 40, 9)
    elif name == "banded16":
        m = n = 6001
        rp, ci, a = _banded(rng, m, 3000, 7)
    elif name == "regular_ragged":
        m, n = 4099, 257
        rp, ci, a = synth(rng, m, n, "regular")
    elif name == "one_row":
        m, n = 1000, 70000
        rp, ci, a = synth(rng, m, n, "one_row")
    elif name == "empty":
        m, n = 777, 555
        rp, ci, a = synth(rng, m, n, "empty")
    else:
        m = n = 30000
        rp, ci, a = synth(rng, m, n, name)
    x = rng.uniform(-1, 1, n)
    for dtype in (np.float64, np.float32):
        for sigma in (64, 1024, 16384):
            H = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_c=64, sell_delta=1, sell_sigma=sigma, sell_split=1, convert_on=2)
            D = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_c=64, sell_delta=1, sell_sigma=sigma, sell_split=1, convert_on=1)
            lh, ld = H.sell_layout(), D.sell_layout()
            assert D.mem_footprint == H.mem_footprint and D.format_name == H.format_name
            for k in ("num_slices", "nnz_ext", "C", "sigma"):
                assert lh[k] == ld[k], k
            for k in ("row_of_sorted", "slice_ptr", "col", "val"):
                np.testing.assert_array_equal(ld[k], lh[k], err_msg=f"{name} sigma={sigma} {k}")
            np.testing.assert_array_equal(D.spmv(x), H.spmv(x))
            y_ref = oracle.csr_spmv(rp, ci, a, x, dtype, num_threads=1)
            np.testing.assert_array_equal(D.spmv(x), y_ref)           # left-to-right FMA per row: bit-exact
            H.close()
            D.close()


@pytest.mark.parametrize("kind", ["affine", "lane_offsets", "exceptions", "delta8", "delta16", "int32"])
def test_every_index_mode_at_every_width(eng, oracle, kind):
    """The delta layout keeps a lane's values in PAIRS of steps (kernels_sell.hip: sell_group_values; the last step of an odd width stands
    alone) and walks 16 / 12 / 8 / 4 steps per trip: every index mode of the layout at every slice width 1 .. 19 (all residues mod 4, trips
    of 4 + 3 + 2 + 1 groups and the 1 - 3 step tail), one and several waves per slice, fp64 and fp32 — the mode is really the one meant
    (index bytes per group of 4 steps), host and GPU builder give the same bytes, and y is bit-identical to the sequential CPU kernel
    with one wave per slice."""
    rng = np.random.default_rng(23)
    m, n = 320, 2_000_000                                 # five slices
    want = dict(affine=16, lane_offsets=16, delta8=272, delta16=528, int32=1024)
    for w in range(1, 20):
        if kind == "affine":
            first = np.arange(m, dtype=np.int64) + 7
            steps = np.sort(rng.choice(np.arange(0, 4000), w, replace=False))
            cols = first[:, None] + steps[None, :]
        elif kind in ("lane_offsets", "exceptions"):
            first = np.repeat(np.arange(m // 64) * 20000, 64) + np.tile(rng.permutation(64) * 3, m // 64) + 11
            steps = np.sort(rng.choice(np.arange(0, 4000), w, replace=False)) * 200
            cols = first[:, None] + steps[None, :]
            if kind == "exceptions" and w > 1:
                for r in (5, 70, 71, 200):                 # a few rows out of line by a column or two on the later steps
                    cols[r, 1:] += rng.integers(1, 3, w - 1).cumsum() % 3 + 1 if w > 1 else 0
        else:
            span = dict(delta8=250, delta16=60000, int32=n - 10)[kind]
            base = np.repeat(rng.integers(0, n - span - 1, m // 64), 64)
            cols = np.sort(np.stack([rng.choice(span, w, replace=False) for _ in range(m)]), axis=1) + base[:, None]
        cols = np.sort(cols, axis=1)
        assert np.all(np.diff(cols, axis=1) > 0) if w > 1 else True
        rp = (np.arange(m + 1) * w).astype(np.int32)
        ci = cols.reshape(-1).astype(np.int32)
        a = rng.uniform(-1, 1, m * w)
        x = rng.uniform(-1, 1, n)
        for dtype in (np.float64, np.float32):
            vb = np.dtype(dtype).itemsize
            y_ref = oracle.csr_spmv(rp, ci, a, x, dtype, num_threads=1)
            H = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_c=64, sell_delta=1, sell_sigma=64, sell_split=1, sell_window=2, convert_on=2)
            D = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_c=64, sell_delta=1, sell_sigma=64, sell_split=1, sell_window=2, convert_on=1)
            assert D.mem_footprint == H.mem_footprint
            groups = (w + 3) // 4
            idx_bytes = H.mem_footprint - (m // 64 + 1) * 16 - m * w * vb - m * 4
            per_group = (idx_bytes / (m // 64) - (256 if kind == "lane_offsets" else 0)) / groups
            if kind == "exceptions":
                if groups >= 2:                            # three slices hold rows out of line: mode 5 (272 B + 32 B per group) beats 8-bit deltas (272 B per group)
                    assert idx_bytes == 3 * (272 + 32 * groups) + 2 * (256 + 16 * groups), (w, idx_bytes)
            elif w > 1:                                    # (a one-step slice is a lane-offset slice whatever its columns: 256 + 16 bytes)
                assert abs(per_group - want[kind]) < 1e-9 or (kind in ("delta8", "delta16") and per_group <= want[kind]), (kind, w, per_group)
            lh, ld = H.sell_layout(), D.sell_layout()
            for k in ("slice_ptr", "col", "val", "row_of_sorted"):
                np.testing.assert_array_equal(ld[k], lh[k], err_msg=f"{kind} w={w} {k}")
            np.testing.assert_array_equal(ld["col"].reshape(-1, 64)[:w * (m // 64)].reshape(m // 64, w, 64).transpose(0, 2, 1).reshape(m, w), cols)
            np.testing.assert_array_equal(D.spmv(x), y_ref, err_msg=f"{kind} w={w} {dtype.__name__}")
            np.testing.assert_array_equal(H.spmv(x), y_ref)
            H.close()
            D.close()
            absrow = oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x))
            for split in (2, 4):
                S = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_c=64, sell_delta=1, sell_sigma=64, sell_split=split, sell_window=2)
                check(S.spmv(x), y_ref, absrow, dtype, False, f"{S.format_name} {kind} w={w}")
                S.close()


@pytest.mark.parametrize("name", ["banded8", "banded16", "short_band", "empty_rows"])
def test_device_window_builder_equals_host_builder(eng, oracle, name):
    """The LDS-window SELL layout (16-bit window-relative indices; kernels_sell_window.hip) built on the GPU (convert_sell.hip:
    sell_window_convert_device) holds the same bytes in every stored array as the host builder (build_sell.hip: build_sell_window), for
    the general and the symmetric-storage variant, and multiplies to the same y."""
    rng = np.random.default_rng(MANIFEST["seed"] + 6)
    if name == "banded8":
        m = 5000
        rp, ci, a = _banded(rng, m, 40, 9)
    elif name == "banded16":
        m = 6001
        rp, ci, a = _banded(rng, m, 3000, 7)
    elif name == "short_band":
        m = 131                                             # two slices and a bit: ragged last slice and last group
        rp, ci, a = _banded(rng, m, 5, 3)
    else:
        m = 4000                                            # a third of the rows empty: the zero slot behind the window
        rp0, ci0, a0 = _banded(rng, m, 60, 11)
        keep = rng.uniform(size=m) > 0.33
        lens = np.where(keep, np.diff(rp0), 0)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        sel = np.repeat(keep, np.diff(rp0))
        ci, a = ci0[sel], a0[sel]
    x = rng.uniform(-1, 1, m)
    cases = [dict(sell_window=1, sell_split=1), dict(sell_window=1, sell_split=2, sell_group=4), dict(sell_window=1, sell_split=1, sell_group=1)]
    for dtype in (np.float64, np.float32):
        for opts in cases:
            H = eng.Matrix(rp, ci, a, m, m, "sell_c_sigma", dtype, convert_on=2, **opts)
            D = eng.Matrix(rp, ci, a, m, m, "sell_c_sigma", dtype, convert_on=1, **opts)
            assert "SELLW" in H.format_name and D.format_name == H.format_name and D.mem_footprint == H.mem_footprint
            for k in ("groups", "row_of_sorted", "desc", "idx", "val"):
                np.testing.assert_array_equal(D.stored_array(k), H.stored_array(k), err_msg=f"{name} {opts} {k}")
            assert H.stored_array("val").size > 0
            np.testing.assert_array_equal(D.spmv(x), H.spmv(x))
            check(D.spmv(x), oracle.csr_spmv(rp, ci, a, x, dtype), oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x)), dtype, False, f"{D.format_name} {name}")
            H.close()
            D.close()
    # symmetric storage: the lower triangle in, the window covers the group's own rows (taken from 65 536 stored entries on)
    if name == "banded8":
        m = 20000
        rp, ci, a = _banded(rng, m, 40, 9)
        rows = np.repeat(np.arange(m), np.diff(rp))
        low = ci <= rows
        rpl = np.concatenate([[0], np.cumsum(np.bincount(rows[low], minlength=m))]).astype(np.int32)
        cil, al = ci[low], a[low]
        assert rpl[-1] >= 65536
        x = rng.uniform(-1, 1, m)
        for dtype in (np.float64, np.float32):
            H = eng.Matrix(rpl, cil, al, m, m, "sell_c_sigma", dtype, symmetric_input=1, sell_window=1, convert_on=2)
            D = eng.Matrix(rpl, cil, al, m, m, "sell_c_sigma", dtype, symmetric_input=1, sell_window=1, convert_on=1)
            assert "SELLWS" in H.format_name and D.format_name == H.format_name and D.mem_footprint == H.mem_footprint
            for k in ("groups", "row_of_sorted", "desc", "idx", "val"):
                np.testing.assert_array_equal(D.stored_array(k), H.stored_array(k), err_msg=f"{name} sym {k}")
            y_ref = oracle.csr_sym_spmv(rpl, cil, al, x, dtype) if hasattr(oracle, "csr_sym_spmv") else None
            if y_ref is not None:
                tol = 1e-12 if dtype == np.float64 else 2e-5
                np.testing.assert_allclose(D.spmv(x), y_ref, rtol=0, atol=tol * 40)
            H.close()
            D.close()


@pytest.mark.parametrize("frac,span", [(0.05, 3), (0.25, 64), (0.6, 3)])
def test_lane_offsets_with_exceptions(eng, oracle, monkeypatch, frac, span):
    """Mode 5 of the delta layout (kernels_sell.hip): a slice whose rows follow one stencil pattern except for a few keeps its
    index-free lane offsets, the rows out of line carry explicit columns. A KKT twin with a fraction of its rows perturbed: host and
    GPU builders give the same bytes, the layout decodes back to the CSR columns, y is bit-identical to the sequential CPU kernel
    (one lane per row, same FMAs), with one and with several waves per slice — and the mode is really taken (fewer index bytes
    than with it switched off) as long as at most 16 of a slice's 64 rows are out of line."""
    import spmv_host as H
    A = H.jitter_columns(H.gen_kkt(20), frac, span)
    rp, ci, a, m, n = A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"]
    x = np.random.default_rng(17).uniform(-1, 1, n)
    for dtype in (np.float64, np.float32):
        y_ref = oracle.csr_spmv(rp, ci, a, x, dtype, num_threads=1)
        Hm = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_c=64, sell_delta=1, sell_sigma=16384, sell_split=1, convert_on=2)
        Dm = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_c=64, sell_delta=1, sell_sigma=16384, sell_split=1, convert_on=1)
        lh, ld = Hm.sell_layout(), Dm.sell_layout()
        assert Dm.mem_footprint == Hm.mem_footprint
        for k in ("row_of_sorted", "slice_ptr", "col", "val"):
            np.testing.assert_array_equal(ld[k], lh[k], err_msg=k)
        # the decoded layout holds the CSR's columns: slice s, step k, lane r = entry k of sorted row 64 s + r
        ros, sp, col = ld["row_of_sorted"], ld["slice_ptr"], ld["col"]
        for sl in (0, len(sp) // 2, len(sp) - 2):
            width = (sp[sl + 1] - sp[sl]) // 64
            for r in (0, 1, 17, 63):
                if sl * 64 + r < m:
                    row = ros[sl * 64 + r]
                    ln = rp[row + 1] - rp[row]
                    np.testing.assert_array_equal(col[sp[sl] + np.arange(min(ln, width)) * 64 + r], ci[rp[row]:rp[row] + min(ln, width)])
        np.testing.assert_array_equal(Dm.spmv(x), y_ref)
        np.testing.assert_array_equal(Hm.spmv(x), y_ref)
        for S in (2, 4):
            W = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_c=64, sell_delta=1, sell_sigma=16384, sell_split=S)
            err = np.abs(W.spmv(x).astype(np.float64) - y_ref.astype(np.float64))
            assert np.all(err <= (1e-12 if dtype == np.float64 else 1e-5) * oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x)))
            W.close()
        with_mode = Dm.mem_footprint
        Hm.close()
        Dm.close()
        monkeypatch.setenv("SPMV_MI355X_SELL_MODES_OFF", "4")
        Off = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_c=64, sell_delta=1, sell_sigma=16384, sell_split=1)
        np.testing.assert_array_equal(Off.spmv(x), y_ref)
        if frac <= 0.25:
            assert with_mode < Off.mem_footprint, (with_mode, Off.mem_footprint)
        Off.close()
        monkeypatch.delenv("SPMV_MI355X_SELL_MODES_OFF")


@pytest.mark.parametrize("C_rows,sigma", [(256, 16384), (64, 16384), (256, 512), (64, 64)])
def test_sell_layout_equals_the_bsc_library_layout(eng, oracle, C_rows, sigma):
    """Row a7: the engine's plain SELL-C-sigma layout against the oracle's restatement of the BSC library's
    (sellcs_format.c:137-200, radix_sort.c:103-122 — itself pinned bit for bit to the reference build, tests/test_oracle_sellcs.py):
    same sigma-window order, same slice widths and pointers, same column-major placement of every real entry (padding columns
    differ by design: the library leaves column 0, the engine repeats the row's last column to keep the gather in cache)."""
    rng = np.random.default_rng(7)
    cases = [(load_case(c)[1], load_case(c)[0]) for c in CASES]
    mats = [(g["row_ptr"], g["col_idx"], g["values"], i["m"], i["n"]) for g, i in cases]
    mats.append(synth(rng, 40000, 40000, "powerlaw") + (40000, 40000))
    mats.append(synth(rng, 30011, 4000, "regular") + (30011, 4000))
    for rp, ci, a, m, n in mats:
        want = oracle.sellcs_layout(rp, ci, a, C_rows, sigma)
        A = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", np.float64, sell_c=C_rows, sell_sigma=sigma, sell_delta=2, sell_window=2)
        got = A.sell_layout()
        A.close()
        assert got["C"] == C_rows and got["sigma"] == sigma
        np.testing.assert_array_equal(got["row_of_sorted"], want["row_order"])
        np.testing.assert_array_equal(got["slice_ptr"], want["slice_ptr"])
        np.testing.assert_array_equal(got["val"], want["val"])
        real = np.zeros(len(want["val"]), bool)                        # positions holding a stored entry (a stored 0.0 included)
        lens = np.diff(rp)[want["row_order"]] if m else np.zeros(0, np.int64)
        for s_ in range(len(want["widths"])):
            rows = lens[s_ * C_rows:(s_ + 1) * C_rows]
            w = int(want["widths"][s_])
            blk = np.zeros((w, C_rows), bool)
            blk[:, :len(rows)] = np.arange(w)[:, None] < rows[None, :]
            real[want["slice_ptr"][s_]:want["slice_ptr"][s_ + 1]] = blk.reshape(-1)
        np.testing.assert_array_equal(got["col"][real], want["col"][real])


def test_device_conversion_on_golden_cases(eng):
    for case in CASES:
        info, z = load_case(case)
        rp, ci, a = z["row_ptr"], z["col_idx"], z["values"]
        m, n = info["m"], info["n"]
        H = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", np.float64, sell_c=64, sell_delta=1, convert_on=2)
        D = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", np.float64, sell_c=64, sell_delta=1, convert_on=1)
        lh, ld = H.sell_layout(), D.sell_layout()
        assert D.mem_footprint == H.mem_footprint, case
        for k in ("row_of_sorted", "slice_ptr", "col", "val"):
            np.testing.assert_array_equal(ld[k], lh[k], err_msg=f"{case} {k}")
        H.close()
        D.close()


# ---- symmetric storage in (row f4: KEEP_SYMMETRY builds, csr_sym.cpp) -------------------------------------------------------

SYM_CASES = [c for c in CASES if MANIFEST["cases"][c].get("sym_nnz") is not None]


@pytest.mark.parametrize("case", SYM_CASES)
def test_symmetric_input_matches_reference_csr_sym(eng, oracle, case):
    """One stored triangle in, y = (T + T^t - diag T) x out. Against the reference csr_sym build's y (fixtures, T = 1) to
    the reordering tolerance, and BIT-identical to the same engine fed the expansion — the expansion is the whole change."""
    info, z = load_case(case)
    m, n = info["m"], info["n"]
    rp, ci, a = z["sym_row_ptr"], z["sym_col_idx"], z["sym_values"]
    # the expansion csr_sym implies: every off-diagonal (i, j, a) also as (j, i, +a) — for skew / Hermitian FILES this is NOT
    # the general-path matrix (which negates / conjugates the mirror image); the symmetric-storage kernel of the reference
    # ignores that too (csr_sym.cpp:204-232, bench_spmv.cpp:135-148)
    import scipy.sparse as sp
    rows = np.repeat(np.arange(m), np.diff(rp))
    off = rows != ci
    E = sp.coo_matrix((np.concatenate([a, a[off]]), (np.concatenate([rows, ci[off]]), np.concatenate([ci, rows[off]]))), shape=(m, n))
    E = E.tocsr()           # duplicates are summed by scipy: none of the symmetric fixtures has any
    assert E.nnz == 2 * len(ci) - int((~off).sum())
    E.sort_indices()
    for dtype, prec in ((np.float64, "d"), (np.float32, "f")):
        for xname, x in (("ones", np.ones(n)), ("rand", z["x_rand"])):
            y_ref = z[f"y_csr_sym_{prec}_{xname}"]
            absrow = abs(E) @ np.abs(x)
            for fmt, opts, _ in VARIANTS[::4]:
                S = eng.Matrix(rp, ci, a, m, n, fmt, dtype, symmetric_input=1, **opts)
                G = eng.Matrix(E.indptr, E.indices, E.data, m, n, fmt, dtype, **opts)
                assert S.nnz == E.nnz and S.m == m
                y = S.spmv(x)
                check(y, y_ref, absrow, dtype, False, f"{case}/{fmt}{opts}/sym/{prec}/{xname}")
                if "COOB" in S.format_name or "MERGEB" in S.format_name:      # LDS atomics: the order of a row's additions is run-dependent
                    check(y, G.spmv(x), absrow, dtype, False, f"{case}/{fmt}{opts}/sym vs expansion")
                else:
                    np.testing.assert_array_equal(y, G.spmv(x))
                S.close()
                G.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_symmetric_storage_kernel_on_banded_matrices(eng, oracle, dtype):
    """Row f4, the traffic-halving half: one stored triangle of a BANDED symmetric matrix in the LDS-window layout, multiplied without
    expanding it (sell_window_sym_kernel: mirrored additions as LDS atomics, the group's y window added to y with coalesced atomics).
    Against the oracle's restatement of the reference csr_sym kernel (pinned bit for bit to the reference build, tests/golden) and
    against the expansion through the general path, to the reordering tolerance; lower and upper triangle; y += A x; the footprint
    is about half the expanded layout's. A matrix whose mirror scatters (no band) is expanded as before."""
    import scipy.sparse as sp
    import spmv_host as H
    tol = 1e-12 if dtype == np.float64 else 1e-5
    A = H.gen_named("cant", 0.5)
    m = A["m"]
    M = sp.csr_matrix((A["values"], A["col_idx"], A["row_ptr"]), shape=(m, m))
    x = np.random.default_rng(23).uniform(-1, 1, m)
    for lower, tri, other in ((True, sp.tril(M).tocsr(), sp.tril(M, -1).T), (False, sp.triu(M).tocsr(), sp.triu(M, 1).T)):
        tri.sort_indices()
        Ex = (tri + other).tocsr()
        Ex.sort_indices()
        rp, ci, a = tri.indptr.astype(np.int32), tri.indices.astype(np.int32), tri.data.astype(np.float64)
        erp, eci, ea = Ex.indptr.astype(np.int32), Ex.indices.astype(np.int32), Ex.data.astype(np.float64)
        absrow = abs(Ex) @ np.abs(x)
        y_sym = oracle.csr_sym_spmv(rp, ci, a, x, dtype)                      # csr_sym.cpp:191-267 restated
        y_exp = oracle.csr_spmv(erp, eci, ea, x, dtype)
        S = eng.Matrix(rp, ci, a, m, m, "sell_c_sigma", dtype, symmetric_input=1, sell_window=1)
        G = eng.Matrix(erp, eci, ea, m, m, "sell_c_sigma", dtype)
        assert "SELLWS" in S.format_name and S.kernel_info()["name"] == "sell_window_sym_kernel"
        assert S.nnz == Ex.nnz and S.m == m and S.mem_footprint < 0.6 * G.mem_footprint
        y = S.spmv(x)
        if lower:          # the reference kernel ASSIGNS y[i] after row i (csr_sym.cpp:233): right for the lower triangle only, what .mtx files hold
            check(y, y_sym, absrow, dtype, False, "symmetric storage vs csr_sym oracle")
        check(y, y_exp, absrow, dtype, False, "symmetric storage vs expansion")
        # y += A x through the device entry point (beta = 1: no clear)
        S.upload_x(x)
        S.upload_y(np.full(m, 2.0))
        S.spmv_device(S.x_device(), S.y_device(), 1, 0)
        check(S.download_y() - 2.0, y_exp, absrow + 2.0, dtype, False, "symmetric storage, beta = 1")
        S.close()
        G.close()
    # default options on a cache-resident matrix: expanded (faster there); a matrix without a band: expanded as well
    S = eng.Matrix(rp, ci, a, m, m, "sell_c_sigma", dtype, symmetric_input=1)
    assert "SELLWS" not in S.format_name
    S.close()
    rng = np.random.default_rng(5)
    n2 = 60000
    R = sp.random(n2, n2, density=12.0 / n2, random_state=rng, format="csr")
    T2 = sp.tril(R + sp.eye(n2)).tocsr()
    T2.sort_indices()
    S = eng.Matrix(T2.indptr.astype(np.int32), T2.indices.astype(np.int32), T2.data.astype(np.float64), n2, n2, "sell_c_sigma", dtype,
                   symmetric_input=1, sell_window=0, sell_sigma=0)
    assert "SELLWS" not in S.format_name
    E2 = (T2 + sp.tril(T2, -1).T).tocsr()
    x2 = rng.uniform(-1, 1, n2)
    check(S.spmv(x2), oracle.csr_spmv(E2.indptr.astype(np.int32), E2.indices.astype(np.int32), E2.data, x2, dtype), abs(E2) @ np.abs(x2), dtype, False,
          "symmetric input without a band")
    S.close()


def test_create_rejects_non_monotone_row_ptr_before_sizing_anything(eng):
    """A row_ptr whose lengths go +10 then -10 used to size the filtered copy from the final prefix sum (0) and overflow it while
    writing at the per-row offsets, before the monotonicity check ran (advisor finding, round 1): now refused up front, on every
    path (plain, row block, column filter)."""
    rp = np.array([0, 10, 0], np.int32)
    ci = np.zeros(10, np.int32)
    a = np.ones(10)
    for opts in ({}, {"row_begin": 0, "row_end": 2, "col_begin": 0, "col_end": 1, "col_filter_mode": 1},
                 {"col_begin": 0, "col_end": 1, "col_filter_mode": 2}):
        with pytest.raises(eng.SpmvError, match="monotone|does not match"):
            eng.Matrix(rp, ci, a, 2, 4, "csr_vector", **opts)
    rp2 = np.array([5, 15, 10, 15], np.int32)                       # a row block of a global CSR, not monotone inside
    with pytest.raises(eng.SpmvError, match="monotone"):
        eng.Matrix(rp2, np.zeros(10, np.int32), a, 3, 4, "csr_vector")


def test_symmetric_input_rejects_bad_input(eng):
    rp = np.array([0, 1, 3], np.int32)
    ci = np.array([0, 0, 1], np.int32)
    with pytest.raises(eng.SpmvError, match="square"):
        eng.Matrix(rp, ci, np.ones(3), 2, 3, "csr_vector", symmetric_input=1)
    with pytest.raises(eng.SpmvError, match="out of range"):
        eng.Matrix(rp, np.array([0, 0, 7], np.int32), np.ones(3), 2, 2, "csr_vector", symmetric_input=1)


def test_driver_keep_symmetry_mode(eng):
    """KEEP_SYMMETRY=1 ./spmv_mi355x_bench file.mtx = the reference's -DKEEP_SYMMETRY build: un-expanded arrays in, the
    symmetric branch of check_accuracy (bench_spmv.cpp:135-148) as the judge."""
    import os
    import subprocess
    from conftest import GOLDEN, ROOT
    exe = os.path.join(ROOT, "spmv-research_amd", "bin", "spmv_mi355x_bench")
    for case in ("symmetric_real", "banded_symmetric"):
        info = MANIFEST["cases"][case]
        env = dict(os.environ, KEEP_SYMMETRY="1", GPU_KERNEL="0", SPMV_MI355X_FORMAT="sell_c_sigma")
        r = subprocess.run([exe, os.path.join(GOLDEN, case + ".mtx")], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        assert "Test failed" not in r.stdout
        row = r.stderr.strip().splitlines()[-1].split(",")
        assert int(row[4]) == info["sym_nnz"] and int(row[5]) == 1          # csr_nnz = stored triangle, symmetry flag
        assert float(row[22]) < 1e-12                                       # spmv_max_ae against the symmetric quad gold


# ---- directly against the GENUINE reference build (oracle/_ref travels with the repo), not through the restatement ----------

def test_gpu_kernels_against_the_reference_build_on_random_matrices(eng):
    """Same inputs into the reference CPU CSR backend (spmv_kernels/csr.cpp compiled in place) and into the GPU kernels:
    bit-identical for the row-sequential kernels, reordering tolerance for the others."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import refdrv
    flavour = "native" if refdrv.available("csr", "d", "native") else "v3"
    if not refdrv.available("csr", "d", flavour):
        pytest.skip("oracle/_ref not present")
    refs = {np.float64: refdrv.RefBackend("csr", "d", flavour, threads=3), np.float32: refdrv.RefBackend("csr", "f", flavour, threads=3)}
    rng = np.random.default_rng(99)
    for trial in range(12):
        kind = ("powerlaw", "short", "regular")[trial % 3]
        m = int(rng.integers(100, 20000))
        n = int(rng.integers(100, 20000))
        rp, ci, a = synth(rng, m, n, kind)
        x = rng.uniform(-1, 1, n)
        absrow = None
        for dtype, ref in refs.items():
            ref.csr_to_format(rp, ci, a, m, n)
            y_ref = ref.spmv(x)
            if absrow is None:
                refs[np.float64].csr_to_format(rp, ci, np.abs(a), m, n)
                absrow = refs[np.float64].spmv(np.abs(x)).astype(np.float64)
                ref.csr_to_format(rp, ci, a, m, n)
                y_ref = ref.spmv(x)
            for fmt, opts, exact in (("csr_scalar", {}, True), ("sell_c_sigma", {"sell_split": 1}, True), ("csr_vector", {}, False),
                                     ("csr_stream", {}, False), ("csr_merge", {}, False), ("coo", {}, False)):
                A = eng.Matrix(rp, ci, a, m, n, fmt, dtype, **opts)
                check(A.spmv(x), y_ref, absrow, dtype, exact, f"vs reference build: trial {trial} {kind} {fmt} {np.dtype(dtype).name}")
                A.close()


def test_merge_drops_the_value_stream_of_pattern_matrices(eng, oracle):
    """Matrix-Market `pattern` matrices (soc-LiveJournal1) hold the dummy value 1.0 everywhere (matrix_market.c:308-317): the
    merge kernel keeps the constant and not the array — same bits as with the array, smaller footprint."""
    rng = np.random.default_rng(5)
    rp, ci, a = synth(rng, 30000, 30000, "powerlaw")
    x = rng.uniform(-1, 1, 30000)
    for const in (1.0, -2.5):
        ones = np.full(len(ci), const)
        for dtype in (np.float64, np.float32):
            U = eng.Matrix(rp, ci, ones, 30000, 30000, "csr_merge", dtype)
            assert "_unit_" in U.format_name and U.mem_footprint < U.csr_mem_footprint
            # the same matrix with ONE different value keeps its array: identical arithmetic elsewhere
            pert = ones.copy()
            pert[-1] = const * 2
            G = eng.Matrix(rp, ci, pert, 30000, 30000, "csr_merge", dtype)
            assert "_unit_" not in G.format_name
            yu, yg = U.spmv(x), G.spmv(x)
            last_row = int(np.searchsorted(rp, len(ci) - 1, side="right") - 1)
            keep = np.ones(30000, bool)
            keep[last_row] = False
            np.testing.assert_array_equal(yu[keep], yg[keep])
            y_ref = oracle.csr_spmv(rp, ci, ones, x, dtype, num_threads=2)
            absrow = oracle.csr_spmv(rp, ci, np.abs(ones), np.abs(x))
            check(yu, y_ref, absrow, dtype, False, f"unit merge {const} {np.dtype(dtype).name}")
            U.close()
            G.close()


def test_engine_placed_vectors(eng, oracle):
    """The handle's own x / y pair and an output vector from output_alloc (csrc/placement.hip): results through them equal the
    host-buffer path; y += A x through upload_y. Small matrix: no placement (below 8 MiB, and not asked for), only the plumbing."""
    info, g = load_case("general_real")
    rp, ci, a = g["row_ptr"], g["col_idx"], g["values"]
    m, n = info["m"], info["n"]
    A = eng.Matrix(rp, ci, a, m, n, "csr_vector")
    x = np.random.default_rng(5).uniform(-1, 1, n)
    y_ref = A.spmv(x)
    A.upload_x(x)
    xp, yp = A.x_device(), A.y_device()
    assert xp and yp and xp != yp
    A.upload_y(np.full(m, 3.0))
    A.spmv_device(xp, yp, 1, 0)
    np.testing.assert_allclose(A.download_y(), y_ref + 3.0, rtol=1e-13, atol=1e-13)
    v = A.output_vector()
    assert v.count == m + 64
    t = v.torch()
    assert float(t.abs().sum().item()) == 0.0          # zero-filled
    A.spmv_device(xp, v.ptr, 0, 0)
    import torch
    torch.cuda.synchronize()
    assert np.array_equal(t[:m].cpu().numpy(), y_ref)
    v.free()
    A.close()


def test_placement_search_keeps_results(eng):
    """A matrix whose y is above the 8 MiB threshold, placement asked for (opts.placement = 1): the first handle walks the device's
    free memory for the two vector pools, its vectors and a caller's output / input vectors are pool slices; results must not change,
    the pools outlive the handle and go when asked."""
    m = 4_600_000
    rp = np.arange(0, 3 * m + 1, 3, dtype=np.int32)
    rows = np.arange(m, dtype=np.int64)
    ci = np.stack([np.maximum(rows - 1, 0), rows, np.minimum(rows + 1, m - 1)], axis=1)
    ci[0] = [0, 1, 2]
    ci[-1] = [m - 3, m - 2, m - 1]
    ci = ci.reshape(-1).astype(np.int32)
    a = np.tile(np.array([0.5, 2.0, -0.25]), m)
    A = eng.Matrix(rp, ci, a, m, m, "sell_c_sigma", placement=1, placement_budget_gib=64)
    x = np.random.default_rng(7).uniform(-1, 1, m)
    v = A.output_vector()                                # before the handle's own pair exists (only its zeroed x): makes the walk
    y = A.spmv(x)                                        # places the handle's y and x, uploads x
    ref = 0.5 * x[ci[0::3]] + 2.0 * x[ci[1::3]] - 0.25 * x[ci[2::3]]
    np.testing.assert_allclose(y, ref, rtol=1e-13, atol=1e-13)
    A.spmv_device(A.x_device(), v.ptr, 0, 0)
    import torch
    torch.cuda.synchronize()
    np.testing.assert_allclose(v.torch()[:m].cpu().numpy(), ref, rtol=1e-13, atol=1e-13)
    xin = A.input_vector()                               # a vector the kernel READS, e.g. the x a collective fills
    xin.torch().copy_(torch.from_numpy(x).cuda())
    A.spmv_device(xin.ptr, v.ptr, 0, 0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(v.torch()[:m].cpu().numpy(), ref, rtol=1e-13, atol=1e-13)
    B = eng.Matrix(rp, ci, a, m, m, "csr_vector", placement=1)      # a second handle: no walk, slices of the same pools
    np.testing.assert_allclose(B.spmv(x), ref, rtol=1e-12, atol=1e-12)
    B.close()
    v.free()
    xin.free()
    A.close()
    eng.placement_release()                              # nothing of the pools is live any more
    # default: placement off — plain allocations, nothing held behind the handle's back
    free0 = torch.cuda.mem_get_info()[0]
    A = eng.Matrix(rp, ci, a, m, m, "sell_c_sigma")
    np.testing.assert_allclose(A.spmv(x), ref, rtol=1e-13, atol=1e-13)
    A.close()
    assert abs(torch.cuda.mem_get_info()[0] - free0) < (64 << 20)


def test_placement_moves_index_arrays_safely(eng, oracle):
    """A matrix with scattered columns (4-byte SELL indices: an index array of tens of MiB) and a y above the placement threshold:
    the pass copies the index array to other sites and times the kernel there — a trial launched before its copy had finished would
    gather x through garbage columns (that race faulted the GPU once; the copy now runs on the trial's stream)."""
    m, k = 1_300_000, 12
    rng = np.random.default_rng(11)
    ci = np.sort(rng.integers(0, m, (m, k), dtype=np.int64), axis=1)
    ci += np.arange(k)                                   # strictly increasing inside a row
    ci = np.minimum(ci, m - 1 - (k - 1 - np.arange(k))).astype(np.int32).reshape(-1)
    rp = np.arange(0, m * k + 1, k, dtype=np.int32)
    a = rng.uniform(-1, 1, m * k)
    A = eng.Matrix(rp, ci, a, m, m, "sell_c_sigma", placement=3, placement_budget_gib=64)      # 3 = pools + the search over the matrix arrays
    x = rng.uniform(-1, 1, m)
    y = A.spmv(x)                                        # first use of the handle's vectors: the placement pass
    sample = rng.integers(0, m, 3000)
    ref = np.array([np.dot(a[i * k:(i + 1) * k], x[ci[i * k:(i + 1) * k]]) for i in sample])
    den = np.array([np.dot(np.abs(a[i * k:(i + 1) * k]), np.abs(x[ci[i * k:(i + 1) * k]])) for i in sample])
    assert np.all(np.abs(y[sample] - ref) <= 1e-12 * den)
    info = eng.placement_info(0)                         # what the one walk of the device found (spmv_mi355x_placement_info)
    assert info["state"] in ("pools kept", "no contrast: plain allocations") and info["candidates"] >= 1 and info["walked_gib"] <= 64
    assert (info["pools"] >= 2 and len(info["us_per_pool"]) == info["pools"] and min(info["us_per_pool"]) > 0) or info["pools"] == 0
    # the same search for a CALLER's vector pair (spmv_mi355x_place_arrays: what bench_multi.py runs for its local part)
    import torch
    xo, yo = A.input_vector(), A.output_vector()
    xo.torch().copy_(torch.from_numpy(x).cuda())
    A.place_arrays(xo.ptr, yo.ptr)                       # moves arrays, overwrites y
    A.spmv_device(xo.ptr, yo.ptr, 0, 0)
    torch.cuda.synchronize()
    y2 = yo.torch()[:m].cpu().numpy()
    assert np.all(np.abs(y2[sample] - ref) <= 1e-12 * den) and np.array_equal(y2, y)      # one lane per row: the same FMAs wherever the arrays lie
    xo.free()
    yo.free()
    A.close()
    eng.placement_release()
    assert eng.placement_info(0)["state"] == "no walk"


def test_handle_from_a_csr_that_arrives_in_pieces(eng, oracle):
    """spmv_mi355x_create_from_stream: rows appended piece by piece to a device-resident CSR give the SAME handle as create() on the
    whole matrix (same format name, same footprint, bit-identical y), fp64 and fp32; the pieces are validated as they pass."""
    import spmv_host as H
    A = H.gen_kkt(22)                                    # 24 k rows: one lane per row, several sigma windows
    rp, ci, a, m, n = A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"]
    x = np.random.default_rng(3).uniform(-1, 1, n)
    cuts = [0, 1, 700, 701, 9000, m - 5, m]               # ragged pieces, one of a single row
    for dtype in (np.float64, np.float32):
        whole = eng.Matrix(rp, ci, a, m, n, "sell_c_sigma", dtype, sell_window=2)
        st = eng.CsrStream(m, n, int(rp[m]) + 1000)       # capacity is an upper bound
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            s, e = int(rp[r0]), int(rp[r1])
            st.append(rp[r0:r1 + 1] - s, ci[s:e], a[s:e])
        pieces = st.finish("sell_c_sigma", dtype)
        assert pieces.format_name == whole.format_name and pieces.mem_footprint == whole.mem_footprint
        assert (pieces.m, pieces.n, pieces.nnz) == (m, n, int(rp[m]))
        assert np.array_equal(pieces.spmv(x.astype(dtype)), whole.spmv(x.astype(dtype)))
        pieces.close()
        whole.close()
    # validation: wrong order of things
    st = eng.CsrStream(4, 4, 10)
    with pytest.raises(eng.SpmvError, match="start at 0"):
        st.append(np.array([1, 2], np.int32), np.array([0], np.int32), np.ones(1))
    with pytest.raises(eng.SpmvError, match="out of range"):
        st.append(np.array([0, 1], np.int32), np.array([9], np.int32), np.ones(1))
    st.append(np.array([0, 1, 2], np.int32), np.array([0, 3], np.int32), np.ones(2))
    with pytest.raises(eng.SpmvError, match="2 of 4 rows"):
        st.finish("sell_c_sigma")
    st = eng.CsrStream(2, 2, 4)
    st.append(np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32), np.ones(2))
    with pytest.raises(eng.SpmvError, match="SELL-C-sigma only"):
        st.finish("csr_vector")
    st = eng.CsrStream(2, 2, 1)
    with pytest.raises(eng.SpmvError, match="capacity"):
        st.append(np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32), np.ones(2))
    st.discard()
