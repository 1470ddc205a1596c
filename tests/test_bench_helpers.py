"""Host-only pieces of bench.py (no GPU): the algorithmic-bytes formula of SURVEY §8(d), the per-SpMV traffic lookup (a
format that needs several dispatches per SpMV records the per-dispatch counters and how many there are), the defaults."""
import importlib.util
import json
import os

import numpy as np

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_formula():
    b = _bench()
    # nnz*(V+4) + (m+1)*4 + (n+m)*V
    assert b.algorithmic_bytes(10, 20, 100, 8) == 100 * 12 + 11 * 4 + 30 * 8
    assert b.algorithmic_bytes(10, 20, 100, 4) == 100 * 8 + 11 * 4 + 30 * 4
    assert b.kkt_edge(1.0) == 240 and b.kkt_edge(1.0 / 8) == 120 and b.kkt_edge(1e-9) == 4


def test_traffic_lookup_is_per_spmv_and_matches_the_kernel_that_ran(tmp_path):
    """A PMC record is attached only to the kernel build, kernel name and converted format it was collected on."""
    b = _bench()
    sha = b.kernel_source_sha()
    assert len(sha) == 16 and sha == b.kernel_source_sha()
    recs = [
        dict(workload="nlpkkt240", dtype="f64", kernel="sell_delta_kernel<double, true>", format_name="MI355X_SELLD_64_16384_d",
             kernel_src_sha=sha, hbm_bytes_per_launch=8_000_000_000),
        dict(workload="soc-LiveJournal1", dtype="f64", kernel="coo_kernel<double, 4, true>", format_name="MI355X_COO_k4_d",
             kernel_src_sha=sha, hbm_bytes_per_launch=1_000, dispatches_per_spmv=3),
        dict(workload="cant", dtype="f64", kernel="csr_vector_kernel<double, 16, false>", format_name="MI355X_CSR_VECTOR_g16_d",
             kernel_src_sha="0123456789abcdef", hbm_bytes_per_launch=5),
    ]
    with open(tmp_path / "traffic_test.json", "w") as f:
        json.dump(dict(records=recs), f)
    d = str(tmp_path)
    assert b.load_traffic("nlpkkt240", "MI355X_SELLD_64_16384_d", "f64", "sell_delta_kernel", pdir=d) == 8_000_000_000
    # another kernel, another converted format (different options / block counts) or another precision must not borrow it
    assert b.load_traffic("nlpkkt240", "MI355X_SELLD_64_16384_d", "f64", "sell_kernel", pdir=d) is None
    assert b.load_traffic("nlpkkt240", "MI355X_SELLD_64_4096_d", "f64", "sell_delta_kernel", pdir=d) is None
    assert b.load_traffic("nlpkkt240", "MI355X_SELLD_64_16384_d", "f32", "sell_delta_kernel", pdir=d) is None
    # several dispatches per SpMV: the record holds the per-dispatch figure
    assert b.load_traffic("soc-LiveJournal1", "MI355X_COO_k4_d", "f64", "coo_kernel", pdir=d) == 3_000
    # counters collected on other kernel sources are stale
    assert b.load_traffic("cant", "MI355X_CSR_VECTOR_g16_d", "f64", "csr_vector_kernel", pdir=d) is None
    # the committed records all carry what the lookup keys on (older rounds' files without it simply never match)
    for f in os.listdir(os.path.join(ROOT, "profiles")):
        if f.startswith("traffic_r02") and f.endswith(".json"):
            for r in json.load(open(os.path.join(ROOT, "profiles", f)))["records"]:
                assert {"workload", "dtype", "kernel", "format_name", "kernel_src_sha", "hbm_bytes_per_launch"} <= set(r)


def test_strided_cpu_sample_covers_the_whole_matrix():
    import numpy as np
    b = _bench()
    rng = np.random.default_rng(0)
    m = 5000
    lens = rng.integers(0, 40, m)
    rp = np.zeros(m + 1, np.int32)
    np.cumsum(lens, out=rp[1:])
    ci = rng.integers(0, m, rp[-1]).astype(np.int32)
    va = rng.uniform(-1, 1, rp[-1])
    srp, sci, sva, rows, what = b.strided_sample(rp, ci, va, m, max_nnz=20000, chunks=16)
    assert srp[0] == 0 and len(srp) == len(rows) + 1 and srp[-1] == len(sci) == len(sva) <= 20000 + 16 * 40
    assert np.all(np.diff(rows) > 0) and rows[0] < m // 16 and rows[-1] > m - m // 8        # spread over the matrix, ascending
    for k in (0, len(rows) // 2, len(rows) - 1):                                             # rows are copied whole
        r = rows[k]
        assert np.array_equal(sci[srp[k]:srp[k + 1]], ci[rp[r]:rp[r + 1]])
    whole = b.strided_sample(rp, ci, va, m, max_nnz=10 ** 9)
    assert len(whole[3]) == m and whole[1] is not None


def test_default_kernel_per_workload():
    b = _bench()
    assert b.DEFAULT_FORMAT["nlpkkt240"] == "sell_c_sigma" and b.DEFAULT_DTYPE.get("nlpkkt240", "f64") == "f64"
    assert b.DEFAULT_FORMAT["soc-LiveJournal1"] == "coo" and b.DEFAULT_OPTS["soc-LiveJournal1"] == {"col_blocks": -1}
    assert b.DEFAULT_DTYPE["pwtk"] == "f32"                      # config 3 of BASELINE.json is the fp32 one
    # the kernels BASELINE.json names: one wavefront per row on scircuit, SELL-C-sigma on pwtk, merge path on soc-LiveJournal1
    assert b.NAMED_KERNEL["scircuit"][0] == "csr_vector" and b.NAMED_KERNEL["scircuit"][1]["lanes_per_row"] == 64
    assert b.NAMED_KERNEL["pwtk"][0] == "sell_c_sigma" and b.NAMED_KERNEL["soc-LiveJournal1"][0] == "csr_merge"
    assert set(b.SMALL_CONFIGS) | {"nlpkkt240"} == set(b.WORKLOADS) == set(b.NAMED_KERNEL)


def test_multi_rank_piece_helpers():
    """bench_multi builds a rank's handles piece by piece: the pieces cover the rows once, in order, within the nnz budget."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))
    import bench_multi as M
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 40, 10_000)
    for budget in (0, 1, 500, 5_000, 10**9):
        pieces = M._chunks(lens, budget)
        assert pieces[0][0] == 0 and pieces[-1][1] == len(lens)
        assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:])) and all(b > a for a, b in pieces)
        if budget >= 500:
            assert max(int(lens[a:b].sum()) for a, b in pieces) <= 1.5 * max(budget, int(lens.sum()) / len(pieces)) + 40
    assert M._chunks(np.zeros(0, np.int64), 10) == [] and M._chunks(np.array([7]), 3) == [(0, 1)]
    sa = [(3, np.array([1, 2]), np.array([1.0, 2.0])), (9, np.array([4]), np.array([4.0]))]
    sb = [(3, np.array([7]), np.array([7.0])), (9, np.zeros(0, np.int64), np.zeros(0))]
    merged = M._merge_samples(sa, sb)
    assert [m[0] for m in merged] == [3, 9] and merged[0][1].tolist() == [1, 2, 7] and merged[1][2].tolist() == [4.0]
    p = M.PeakRSS()
    assert p.gib() > 0.01


def test_filtered_kkt_rows_equal_the_filtered_block():
    """The generator-side column filter (local / remote halves of a rank's rows) against filtering the generated block."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))
    import spmv_host as H
    N = 10
    A = H.gen_kkt(N)
    rp, ci, va, m = A["row_ptr"], A["col_idx"], A["values"], A["m"]
    for (r0, r1, lo, hi) in ((0, m, 0, m // 3), (200, 1400, 180, 1500), (m - 300, m, 0, 50)):
        s, e = int(rp[r0]), int(rp[r1])
        c, v = ci[s:e], va[s:e]
        rows = np.repeat(np.arange(r1 - r0), np.diff(rp[r0:r1 + 1]))
        for keep in (1, 0):
            B = H.gen_kkt_rows_filtered(N, lo, hi, keep, r0=r0, count=r1 - r0)
            sel = ((c >= lo) & (c < hi)) == bool(keep)
            assert np.array_equal(np.diff(B["row_ptr"]), np.bincount(rows[sel], minlength=r1 - r0))
            assert np.array_equal(B["col_idx"], c[sel]) and np.array_equal(B["values"], v[sel])
    some = np.array([5, 17, 900, 901, m - 1], np.int32)
    B = H.gen_kkt_rows_filtered(N, 0, 100, 0, rows=some, values=False)
    assert B["values"] is None and B["m"] == 5 and int(B["row_ptr"][-1]) == B["nnz"] == len(B["col_idx"])
