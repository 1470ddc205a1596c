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


def _run(world, extra, port, backend="gloo"):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", backend, "--steps", "4",
           "--warmup", "2"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 alone prints
    return json.loads(lines[0])


CASES = [
    # (world, extra args, expected winner's partition kind or None, variants that must have been timed)
    (2, ["--scale", "0.01"], None, ["rows+allgather", "graph+halo"]),
    (3, ["--scale", "0.01", "--halo", "p2p"], None, ["rows+allgather", "graph+halo"]),
    (3, ["--scale", "0.01", "--partition", "rows"], "rows", None),
    (2, ["--scale", "0.01", "--partition", "graph", "--overlap", "0"], "graph", ["graph+halo"]),
    (2, ["--scale", "0.01", "--exchange", "allgather", "--partition", "rows"], "rows", ["rows+allgather"]),
    (2, ["--workload", "cant", "--scale", "0.3"], "rows", None),
    (2, ["--workload", "soc-LiveJournal1", "--scale", "0.02"], None, None),
    # a rank's block generated and converted in several pieces (one handle per piece), both variants, with and without overlap
    (3, ["--scale", "0.01", "--host-chunk-nnz", "400000"], None, ["rows+allgather", "graph+halo"]),
    (2, ["--scale", "0.01", "--host-chunk-nnz", "300000", "--overlap", "0"], None, ["rows+allgather", "graph+halo"]),
    (2, ["--workload", "cant", "--scale", "0.3", "--host-chunk-nnz", "200000"], "rows", None),
    # ... and the same with one handle PER PIECE (what formats without a device-side conversion fall back to)
    (2, ["--scale", "0.01", "--host-chunk-nnz", "300000", "--piece-handles"], None, ["rows+allgather", "graph+halo"]),
]


@pytest.mark.parametrize("world,extra,kind,variants", CASES, ids=[f"w{c[0]}-" + "_".join(a.strip("-") for a in c[1]) for c in CASES])
def test_multirank_bench_line(world, extra, kind, variants):
    j = _run(world, extra, 29700 + (abs(hash(tuple(extra))) % 200))
    assert j["n_gpus"] == world and j["steps"] == 4 and j["scaling"] == "strong" and j["value"] > 0
    assert j["check_max_err_over_abs_row"] <= (1e-12 if j["dtype"] == "f64" else 1e-5)
    assert j["roofline"]["traffic"] is None and "cpu_baseline" not in j          # N = 1 only
    p = j["partition"]
    if kind is not None:
        assert p["kind"] == kind
    V = j["variants"]
    assert j["config"]["variant"] in V and abs(V[j["config"]["variant"]]["ms_per_step"] - j["ms_per_step"]) < 1e-9
    assert j["ms_per_step"] == min(v["ms_per_step"] for v in V.values())         # value is the faster variant's
    if variants is not None:
        assert sorted(V) == sorted(variants)
    for name, v in V.items():
        assert v["check_max_err_over_abs_row"] <= (1e-12 if j["dtype"] == "f64" else 1e-5) and v["value"] > 0
        b = v["breakdown_ms"]
        assert b["exchange_alone"] > 0 and b["kernels_alone"] > 0
        q = v["partition"]
        if q["kind"] == "graph":
            c = q["considered_max_remote_x_entries"]
            assert max(q["remote_x_entries_per_rank"]) == c["graph"] and q["layout"] == "original"
            assert q["interior_rows"] + q["boundary_rows"] > 0
            assert v["exchange"]["recv_x_entries"] == q["remote_x_entries_per_rank"][0]
            assert v["exchange"]["chosen"].startswith("packed halo " + ("p2p" if "p2p" in extra else "alltoall"))
        else:
            assert v["exchange"]["chosen"] in ("allgather", "p2p")
    # the KKT twin's breadth-first slabs read far fewer remote entries than its row blocks: "auto" must have looked at both
    if "--workload" not in extra and "--partition" not in extra:
        c = V["graph+halo"]["partition"]["considered_max_remote_x_entries"]
        assert c["graph"] < c["rows"]
    assert j["setup_s"]["max_host_rss_gib_over_ranks"] > 0


RCCL_CASES = [
    (["--scale", "0.02"], "rows+allgather"),                                       # in-place all_gather_into_tensor (+ the p2p exchange it is compared with)
    (["--scale", "0.02", "--partition", "graph"], "graph+halo"),                   # all_to_all_single of the halo lists and of the packed halo
    (["--scale", "0.02", "--partition", "graph", "--halo", "p2p"], "graph+halo"),  # batched isend / irecv
]


@pytest.mark.parametrize("extra,must_have", RCCL_CASES, ids=["rows", "graph-alltoall", "graph-p2p"])
def test_rccl_calls_with_one_rank(extra, must_have):
    """The N > 1 code path on the real backend (nccl = RCCL) with WORLD_SIZE = 1 — all a one-GPU box allows: process-group
    initialisation, the in-place all_gather_into_tensor of the row-block scheme, all_to_all_single / batched isend-irecv of the
    packed halo (empty with one rank), the timing all_reduce. Exchange volumes are trivially zero; what is under test is that
    every RCCL call bench.py makes is accepted by the library."""
    j = _run(1, ["--force-multi"] + extra, 29950, backend="nccl")
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["config"]["variant"] in j["variants"]
    assert must_have in j["variants"]
    for v in j["variants"].values():
        assert v["check_max_err_over_abs_row"] <= 1e-12


def test_multi_rank_path_is_as_fast_per_byte_as_the_single_gpu_path():
    """The N > 1 code path with ONE rank (all a one-GPU box allows under RCCL) against the N = 1 path on the same matrix at 1/8 of
    the headline's size: the row-block handles come from the device-resident CSR stream, their vectors from the engine's pools
    (csrc/placement.hip), the empty remote-column half is not launched — the step must cost what the single handle's SpMV costs
    (round 2: 1.466 against 1.278 ms at full size, every rank of an 8-GPU run would have carried those 13 %)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--scale", "0.125", "--steps", "300", "--warmup", "20", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--configs", "off"] + common, capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    single = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    multi = _run(1, ["--force-multi", "--partition", "rows"] + common[:2] + ["--no-cpu-baseline"], 29951, backend="nccl")
    # _run times 4 steps; time the same path again over 300 for a figure worth comparing
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29952",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-multi", "--partition", "rows"] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    multi = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert multi["config"]["format"] == single["config"]["format"]
    ratio = multi["ms_per_step"] / single["ms_per_step"]
    assert ratio <= 1.03, f"N>1 path {multi['ms_per_step']:.4f} ms against {single['ms_per_step']:.4f} ms for the N=1 path (x{ratio:.3f})"
