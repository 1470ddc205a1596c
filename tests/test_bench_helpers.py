"""Host-only pieces of bench.py (no GPU): the algorithmic-bytes formula of SURVEY §8(d), the per-SpMV traffic lookup (a
format that needs several dispatches per SpMV records the per-dispatch counters and how many there are), the defaults."""
import importlib.util
import json
import os

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


def test_traffic_lookup_is_per_spmv_and_matches_the_kernel_that_ran():
    b = _bench()
    recs = json.load(open(os.path.join(ROOT, "profiles", "traffic_r01.json")))["records"]
    by_tag = {r["tag"]: r for r in recs}
    head = by_tag["nlpkkt240_sell_c_sigma_f64"]
    assert b.load_traffic("nlpkkt240", "sell_c_sigma", "f64", "sell_delta_kernel") == head["hbm_bytes_per_launch"]
    # a different kernel of the same format (plain SELL instead of the delta layout) must not borrow the record
    assert b.load_traffic("nlpkkt240", "sell_c_sigma", "f64", "sell_kernel") == by_tag["nlpkkt240_sell_c_sigma_f64_sell_delta_2"]["hbm_bytes_per_launch"]
    coob = by_tag["soc-LiveJournal1_coo_f64_col_blocks_-1"]
    assert coob["dispatches_per_spmv"] == 2
    assert b.load_traffic("soc-LiveJournal1", "coo", "f64", "coo_blocked_kernel") == 2 * coob["hbm_bytes_per_launch"]
    assert b.load_traffic("cant", "csr_scalar", "f64", "csr_scalar_kernel") is None


def test_default_kernel_per_workload():
    b = _bench()
    assert b.DEFAULT_FORMAT["nlpkkt240"] == "sell_c_sigma" and b.DEFAULT_DTYPE.get("nlpkkt240", "f64") == "f64"
    assert b.DEFAULT_FORMAT["soc-LiveJournal1"] == "coo" and b.DEFAULT_OPTS["soc-LiveJournal1"] == {"col_blocks": -1}
    assert b.DEFAULT_DTYPE["pwtk"] == "f32"                      # config 3 of BASELINE.json is the fp32 one
