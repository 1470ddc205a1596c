"""bench.py's N > 1 path as the driver launches it (python -m torch.distributed.run, one process per rank), rehearsed with
gloo ranks sharing the one GPU of a gpurun box (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's).
Scaled-down workloads: a scaled run is never a reported result — under test are the partition / layout / exchange choices,
that every rank takes the same branches (a rank that did not would dead-lock the collectives and time out here), the
sampled-row check inside bench.py, and the JSON contract for N > 1."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(world, extra, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--steps", "4",
           "--warmup", "2"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 alone prints
    return json.loads(lines[0])


CASES = [
    # (world, extra args, expected partition kind, expected layout, expected exchange prefix)
    (2, ["--scale", "0.01"], "graph", "original", "packed halo alltoall"),
    (3, ["--scale", "0.01", "--halo", "p2p"], "graph", "original", "packed halo p2p"),
    (2, ["--scale", "0.01", "--layout", "padded"], "graph", "padded", None),
    (2, ["--scale", "0.01", "--layout", "padded", "--exchange", "allgather", "--overlap", "0"], "graph", "padded", "allgather"),
    (3, ["--scale", "0.01", "--partition", "rows"], "rows", None, None),
    (2, ["--scale", "0.01", "--overlap", "0"], "graph", "original", "packed halo alltoall"),
    (2, ["--workload", "cant", "--scale", "0.3"], None, None, None),
    (2, ["--workload", "soc-LiveJournal1", "--scale", "0.02"], None, None, None),
]


@pytest.mark.parametrize("world,extra,kind,layout,exchange", CASES, ids=[f"w{c[0]}-" + "_".join(a.strip("-") for a in c[1]) for c in CASES])
def test_multirank_bench_line(world, extra, kind, layout, exchange):
    j = _run(world, extra, 29700 + (abs(hash(tuple(extra))) % 200))
    assert j["n_gpus"] == world and j["steps"] == 4 and j["scaling"] == "strong" and j["value"] > 0
    assert j["check_max_err_over_abs_row"] <= (1e-12 if j["dtype"] == "f64" else 1e-5)
    assert j["roofline"]["traffic"] is None and "cpu_baseline" not in j          # N = 1 only
    p = j["partition"]
    if kind is not None:
        assert p["kind"] == kind
    if layout is not None:
        assert p["layout"] == layout
    if p.get("kind") == "graph":
        c = p["considered_max_remote_x_entries"]
        assert max(p["remote_x_entries_per_rank"]) == c["graph"] and ("rows" not in c or c["graph"] <= c["rows"] or "--partition" in extra)
    if p.get("layout") == "original":
        assert p["interior_rows"] + p["boundary_rows"] > 0
        assert j["exchange"]["recv_x_entries"] == p["remote_x_entries_per_rank"][0]
    if exchange is not None:
        assert j["exchange"]["chosen"].startswith(exchange)
    b = j["breakdown_ms"]
    assert b["exchange_alone"] > 0 and b["kernels_alone"] > 0
