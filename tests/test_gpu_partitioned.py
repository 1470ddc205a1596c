"""Several GPUs behind ONE C-ABI handle (spmv_mi355x_create_partitioned): the row blocks are the reference's
nnz-balanced ranges (lib/parallel_util.h:156-184 with one worker per GPU), every row is summed on one device as
local-column part + remote-column part, y comes back in global row order. A gpurun box has one GPU, so the parts share device
0 and the exchange runs through the copy back end (RCCL refuses a device twice); the RCCL back end is the same code path
behind exchange_x() and is the driver's to run on a real node."""
import os
import subprocess

import numpy as np
import pytest

from conftest import CASES, ROOT, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import spmv_mi355x as E
    assert E.device_count() >= 1
    return E


def _emulate(eng, rp, ci, a, m, n, offsets, fmt, dtype, x, **opts):
    """What the partitioned handle computes, rebuilt from single-GPU handles: for every row block, the product with the block's
    own columns, then += the product with the others (the same two kernels, the same order) — bit for bit."""
    y = np.zeros(m, dtype)
    for p in range(len(offsets) - 1):
        r0, r1 = int(offsets[p]), int(offsets[p + 1])
        if r1 == r0:
            continue
        s, e = int(rp[r0]), int(rp[r1])
        lrp = (rp[r0:r1 + 1] - s).astype(np.int32)
        loc = eng.Matrix(lrp, ci[s:e], a[s:e], r1 - r0, n, fmt, dtype, col_begin=r0, col_end=r1, col_filter_mode=1, **opts)
        rem = eng.Matrix(lrp, ci[s:e], a[s:e], r1 - r0, n, fmt, dtype, col_begin=r0, col_end=r1, col_filter_mode=2, **opts)
        y[r0:r1] = (loc.spmv(x).astype(dtype) + rem.spmv(x).astype(dtype)).astype(dtype)
        loc.close()
        rem.close()
    return y


@pytest.mark.parametrize("nparts", [1, 2, 3, 5])
@pytest.mark.parametrize("fmt,opts", [("sell_c_sigma", {"sell_split": 1}), ("csr_vector", {}), ("csr_merge", {})],
                         ids=["sell", "csr_vector", "csr_merge"])
def test_partitioned_handle_on_one_device(eng, oracle, nparts, fmt, opts):
    import spmv_host as H
    A = H.gen_kkt(14)                                   # square, symmetric pattern, far couplings: every part reads every other
    rp, ci, a, m, n = A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"]
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, n)
    absrow = oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x))
    for dtype, tol in ((np.float64, 1e-12), (np.float32, 1e-5)):
        P = eng.PartitionedMatrix(rp, ci, a, m, n, nparts, fmt, dtype, devices=[0] * nparts, **opts)
        assert P.nparts == nparts and P.format_name.startswith(f"MI355X_PART{nparts}_")
        assert ("none" in P.exchange) if nparts == 1 else ("copies" in P.exchange)
        # the blocks are the reference partitioner's (loop_partitioner_balance_prefix_sums with W = nparts)
        want = [0] + [H.partition_prefix_sums(nparts, p, rp, m, int(rp[m]))[1] for p in range(nparts)]
        np.testing.assert_array_equal(P.offsets, want)
        y = P.spmv(x)
        y_ref = oracle.csr_spmv(rp, ci, a, x, dtype)
        err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
        assert np.all(err <= tol * absrow + 1e-300), f"{fmt} x{nparts}: max {np.max(err / np.maximum(absrow, 1e-300))}"
        if fmt == "sell_c_sigma":                          # one lane per row in both kernels: the split sum is reproducible bit for bit
            np.testing.assert_array_equal(y, _emulate(eng, rp, ci, a, m, n, P.offsets, fmt, dtype, x.astype(dtype), **opts))
        # a new x gives a new y; the device-resident loop (exchange forced every iteration) leaves the same y behind
        y2 = P.spmv(2 * x)
        assert np.all(np.abs(y2.astype(np.float64) - 2 * y_ref.astype(np.float64)) <= 2 * tol * absrow + 1e-300)
        assert P.time(5) > 0
        np.testing.assert_array_equal(P.spmv(2 * x, always_copy=True), y2)
        P.close()


@pytest.mark.parametrize("case", ["banded_symmetric", "general_real", "huge_row", "empty_rows_formats", "tiny"])
def test_partitioned_golden_cases(eng, case):
    info, g = load_case(case)
    if info["m"] != info["n"]:
        pytest.skip("square matrices only")
    rp, ci, a, m, n = g["row_ptr"], g["col_idx"], g["values"], info["m"], info["n"]
    for nparts in (2, 3):
        P = eng.PartitionedMatrix(rp, ci, a, m, n, nparts, "sell_c_sigma", np.float64, devices=[0] * nparts)
        for xn, x in (("ones", np.ones(n)), ("rand", g["x_rand"])):
            np.testing.assert_allclose(P.spmv(x), g[f"y_csr_d_{xn}"], rtol=0, atol=1e-12 * max(1.0, float(np.abs(g[f"y_csr_d_{xn}"]).max()) * 8))
        P.close()


def test_partitioned_rejects_bad_input(eng):
    rp = np.array([0, 1, 2], np.int32)
    ci = np.array([0, 1], np.int32)
    with pytest.raises(eng.SpmvError, match="square"):
        eng.PartitionedMatrix(rp, ci, np.ones(2), 2, 3, 2, "csr_vector", devices=[0, 0])
    with pytest.raises(eng.SpmvError, match="out of range"):
        eng.PartitionedMatrix(rp, np.array([0, 9], np.int32), np.ones(2), 2, 2, 2, "csr_vector", devices=[0, 0])
    with pytest.raises(eng.SpmvError, match="RCCL"):
        eng.PartitionedMatrix(rp, ci, np.ones(2), 2, 2, 2, "csr_vector", devices=[0, 0], exchange=1)       # RCCL needs distinct devices
    with pytest.raises(eng.SpmvError, match="device"):
        eng.PartitionedMatrix(rp, ci, np.ones(2), 2, 2, 2, "csr_vector", devices=[0, 99])


def test_adapter_honours_ngpus(tmp_path):
    """The Matrix_Format adapter (host/spmv_kernel_mi355x.cpp) run by the stand-alone driver with SPMV_MI355X_NGPUS=3: same CSV
    row shape, a partitioned format name, errors within the driver's own bar."""
    import spmv_host as H
    A = H.gen_named("cant", 0.1)
    path = str(tmp_path / "cant_small.mtx")
    H.mtx_write_csr(path, A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"])
    exe = os.path.join(ROOT, "spmv-research_amd", "bin", "spmv_mi355x_bench")
    env = dict(os.environ, SPMV_MI355X_NGPUS="3", SPMV_MI355X_FORMAT="sell_c_sigma", GPU_KERNEL="0", OMP_NUM_THREADS="4")
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    row = [l for l in r.stderr.splitlines() if "MI355X_PART3_" in l]
    assert len(row) == 1, r.stderr[-2000:]
    assert "Test failed" not in r.stdout + r.stderr


def test_driver_artificial_matrix_mode(tmp_path):
    """USE_ARTIFICIAL_MATRICES=1: the generator's feature vector on the command line (bench.cpp:569-579), the synthetic-dataset CSV row
    (bench_spmv.cpp:532-559) on stderr."""
    exe = os.path.join(ROOT, "spmv-research_amd", "bin", "spmv_mi355x_bench")
    env = dict(os.environ, USE_ARTIFICIAL_MATRICES="1", SPMV_MI355X_FORMAT="csr_vector", GPU_KERNEL="0", OMP_NUM_THREADS="4")
    args = "20000 20000 12.5 3.0 normal random 0.05 2.0 1.2 0.6 14 mytest".split()
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    row = [l for l in r.stderr.splitlines() if l.startswith("synthetic,")]
    assert len(row) == 1, r.stderr[-2000:]
    f = row[0].split(",")
    assert len(f) == 28 and f[1:4] == ["normal", "random", "14"] and f[4:6] == ["20000", "20000"] and f[23].startswith("MI355X_CSR_VECTOR")
    assert abs(float(f[10]) - 12.5) < 0.2 and float(f[25]) > 0                      # avg_nnz_per_row as asked, gflops measured
    assert "Test failed" not in r.stdout + r.stderr and "time generate artificial matrix" in r.stdout


def test_rccl_back_end_with_one_part(eng, oracle):
    """exchange = 1 forced on ONE part: librccl is dlopen'ed, ncclCommInitAll(1 device) and the grouped in-place ncclAllGather
    of the one slice run on the real library — all a one-GPU box allows of the RCCL back end (csrc/partitioned.hip)."""
    import spmv_host as H
    A = H.gen_kkt(12)
    rp, ci, a, m, n = A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"]
    x = np.random.default_rng(9).uniform(-1, 1, n)
    P = eng.PartitionedMatrix(rp, ci, a, m, n, 1, "sell_c_sigma", np.float64, devices=[0], exchange=1)
    assert "RCCL" in P.exchange
    y = P.spmv(x)
    absrow = oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x))
    assert np.all(np.abs(y - oracle.csr_spmv(rp, ci, a, x)) <= 1e-12 * absrow + 1e-300)
    assert P.time(5) > 0
    y2 = P.spmv(2 * x)
    assert np.all(np.abs(y2 - 2 * y) <= 4e-12 * absrow + 1e-300)
    P.close()


@pytest.mark.parametrize("exchange", [0, 1, 2], ids=["auto", "rccl", "peer-copies"])
def test_parts_on_distinct_devices(eng, oracle, exchange):
    """The first box with more than one GPU runs what a one-GPU box cannot: parts on DISTINCT devices, ncclCommInitAll over them with
    the grouped in-place ncclAllGather (exchange 0 / 1) and hipMemcpyPeerAsync (exchange 2); the caller's current device is left as
    it was by every entry point. Skipped on the one-GPU boxes of the pool."""
    import torch
    ndev = eng.device_count()
    if ndev < 2:
        pytest.skip("needs at least two GPUs")
    import spmv_host as H
    A = H.gen_kkt(16)
    rp, ci, a, m, n = A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"]
    x = np.random.default_rng(21).uniform(-1, 1, n)
    absrow = oracle.csr_spmv(rp, ci, np.abs(a), np.abs(x))
    y_ref = oracle.csr_spmv(rp, ci, a, x)
    nparts = min(ndev, 4)
    torch.cuda.set_device(ndev - 1)                          # not device 0: a restore to "0" would show
    P = eng.PartitionedMatrix(rp, ci, a, m, n, nparts, "sell_c_sigma", np.float64, devices=list(range(nparts)), exchange=exchange, sell_split=1)
    assert torch.cuda.current_device() == ndev - 1
    assert ("RCCL" in P.exchange) if exchange in (0, 1) else ("cop" in P.exchange)
    y = P.spmv(x)
    assert torch.cuda.current_device() == ndev - 1
    assert np.all(np.abs(y - y_ref) <= 1e-12 * absrow + 1e-300)
    np.testing.assert_array_equal(y, _emulate(eng, rp, ci, a, m, n, P.offsets, "sell_c_sigma", np.float64, x, sell_split=1))
    assert P.time(5) > 0 and torch.cuda.current_device() == ndev - 1
    P.close()
    assert torch.cuda.current_device() == ndev - 1
    torch.cuda.set_device(0)
