"""bench.py keeps the driver's contract: one JSON line on stdout with the agreed keys, N = 1 default path, through the
C ABI on the GPU (scaled-down workload so it runs in seconds; a scaled run is never a reported result)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.parametrize("extra", [[], ["--format", "csr_stream"], ["--workload", "pwtk", "--scale", "0.2"]])
def test_bench_json_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--cpu-baseline-seconds", "0.5"]
    if "--scale" not in extra:
        cmd += ["--scale", "0.01"]
    r = subprocess.run(cmd + extra, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert KEYS <= set(j)
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["unit"] == "GFLOP/s" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert j["dtype"] in ("f64", "f32") and "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # one clock for every fraction: the wall time per step (ms_per_step); the event time is kept beside it; a short sample is five windows
    assert abs(rf["ms"] - j["ms_per_step"]) < 1e-9 and rf["kernel_ms"] > 0
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (j["ms_per_step"] * 1e-3) / 1e9) <= 0.01 * rf["achieved"]
    assert abs(j["hbm_pct_of_peak"] - 100 * rf["frac"]) < 0.02
    assert j["windows"] == 5 and j["ms_per_step_min"] <= j["ms_per_step"] <= j["ms_per_step_max"]
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert j["check_max_err_over_abs_row"] <= (1e-12 if j["dtype"] == "f64" else 1e-5)


def test_bench_configs_carry_cold_legs_cpu_baselines_and_the_symmetric_leg():
    """The small configs as the default run reports them (scaled down here): named kernel = the literal one BASELINE.json words
    (config 4: the CSR-order merge path), best kernel, cold figures for the cache-resident ones, a CPU baseline each, and the
    symmetric-storage object for the two FEM twins."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--cpu-baseline-seconds", "0.3",
           "--scale", "0.05", "--configs", "on", "--configs-steps", "20"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    by = {c["workload"].split()[0]: c for c in j["configs"]}
    assert set(by) == {"cant", "scircuit", "pwtk", "soc-LiveJournal1"}
    assert by["soc-LiveJournal1"]["named_kernel"] == "merge_kernel" and by["soc-LiveJournal1"]["best_kernel"] == "coo_blocked_kernel"
    assert by["pwtk"]["dtype"] == "f32"
    for w, c in by.items():
        assert c["named_frac"] > 0 and c["best_frac"] > 0 and c["cpu_baseline"]["value"] > 0
        if c["cache_resident"]:
            assert c["cold_frac"] > 0 and c["named_cold_frac"] > 0 and c["best_cold_ms"] >= 0.5 * c["best_kernel_ms"]
    for w in ("cant", "pwtk"):
        sym = by[w]["symmetric_storage"]
        assert sym["kernel"] == "sell_window_sym_kernel" and sym["stored_nnz"] < 0.55 * sym["expanded_nnz"]
        assert sym["mem_footprint"] < 0.7 * sym["expanded"]["mem_footprint"]      # 0.54 / 0.58 at full size; padding weighs more on a 1/20 twin
