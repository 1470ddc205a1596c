"""Row-partitioned solvers (SURVEY §8 rows e + f3) on the GPU: `world` gloo ranks share the one MI355X of a gpurun box (an
8-GPU node is the driver's to run; RCCL refuses two ranks on one device, gloo moves the slices through the host). Under
test: spmv_mi355x_pcg_dist / _pbicgstab_dist with spmv_dist.DistributedSolver's callbacks — every rank must return its
slice of the single-GPU solver's x, the same history / iteration count on all ranks (the `err < eps` break is taken at the
same iteration: a rank that stopped early would dead-lock the collectives), and the oracle's numbers to tolerance."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _system(kind):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_solvers import laplace2d, nonsymmetric_dd, rhs
    A = laplace2d(48) if kind == "laplace" else nonsymmetric_dd(3000, 0.004, 9)
    return A, rhs(A)


def _worker(rank, world, port, kind, method, iters, q, partition="rows"):
    for p in (os.path.join(ROOT, "spmv-research_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import spmv_dist as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        A, b = _system(kind)
        if partition == "graph":
            # breadth-first slabs, x in original numbering, packed halo exchange overlapped with the interior rows
            rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
            part = D.graph_partition(rp, ci, A.shape[0], A.shape[1], world, "graph")
            S = D.GraphDistributedSolver(dist, torch, rp, ci, A.data, part, rank, world, fmt="csr_vector")
            r = (S.pcg if method == "pcg" else S.pbicgstab)(b[S.rows], iters)
            rows = np.asarray(S.rows)
        else:
            off = D.row_partition(A.indptr.astype(np.int32), world)
            blk = D.local_block(A.indptr.astype(np.int32), A.indices, A.data, off, rank)
            S = D.DistributedSolver(dist, torch, blk, off, rank, world, fmt="csr_vector")
            r = (S.pcg if method == "pcg" else S.pbicgstab)(b[off[rank]:off[rank + 1]], iters)
            rows = np.arange(off[rank], off[rank + 1])
        q.put((rank, r["x"], r["iterations"], r["history"], r["error"], r["error_best"], r["restarts"], dict(S.calls), rows))
        dist.barrier()
    except Exception as e:                                        # surface the failure instead of a hung join
        import traceback
        q.put((rank, "ERROR", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partition", ["rows", "graph"])
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("kind,method,iters", [("laplace", "pcg", 1000), ("laplace", "pbicgstab", 130), ("nonsym", "pbicgstab", 130)])
def test_distributed_solver_matches_single_gpu(oracle, world, kind, method, iters, partition):
    import spmv_mi355x as eng
    A, b = _system(kind)
    M = eng.Matrix(A.indptr, A.indices, A.data, A.shape[0], A.shape[1], "csr_vector")
    single = (M.pcg if method == "pcg" else M.pbicgstab)(A.indptr, A.indices, A.data, b, iters)
    want = (oracle.pcg if method == "pcg" else oracle.pbicgstab)(A.indptr, A.indices, A.data, b, iters)
    M.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + world * 7 + (hash((kind, method, partition)) % 50)
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, method, iters, q, partition)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=300)
        assert not (isinstance(item[1], str) and item[1] == "ERROR"), item[2]
        res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = np.zeros(A.shape[0])
    for r in range(world):
        x[res[r][7]] = res[r][0]                                   # rank r's slice sits at the rows it owns
    its = {res[r][1] for r in range(world)}
    assert len(its) == 1, f"ranks disagree on the iteration count: {its}"
    it = its.pop()
    # the last decades before eps = 1e-15*|b| are rounding-dominated: the sequential oracle may need ~10 % more or fewer
    # iterations than the tree-summed GPU dots; distributed vs single GPU differ only in the summation tree of the dots
    assert abs(it - single["iterations"]) <= 2 and abs(it - want["iterations"]) <= max(2, 0.1 * want["iterations"])
    for r in range(1, world):                                      # identical scalars on every rank
        np.testing.assert_array_equal(res[r][2], res[0][2])
        assert res[r][3:6] == res[0][3:6]
    k = min(20, it, single["iterations"], want["iterations"])
    tol = 1e-9 if method == "pcg" else 1e-7
    np.testing.assert_allclose(res[0][2][:k], single["history"][:k], rtol=tol)
    np.testing.assert_allclose(res[0][2][:k], want["history"][:k], rtol=tol)
    assert np.linalg.norm(x - single["x"]) <= 1e-9 * np.linalg.norm(single["x"])
    assert np.linalg.norm(b - A @ x) == pytest.approx(res[0][3], rel=1e-3, abs=1e-13 * np.linalg.norm(b))
    calls = res[0][6]
    per_it = (1, 2) if method == "pcg" else (2, 3)
    assert calls["spmv"] >= per_it[0] * it + 2 and calls["allreduce"] >= per_it[1] * it + 2
