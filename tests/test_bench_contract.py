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
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert j["check_max_err_over_abs_row"] <= (1e-12 if j["dtype"] == "f64" else 1e-5)
