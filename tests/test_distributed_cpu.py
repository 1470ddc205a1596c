"""world_size-2 gloo rehearsal (CPU) of the multi-GPU path of bench.py: nnz-balanced row blocks, padded x slices filled
by ONE in-place all_gather_into_tensor, local-column / remote-column split with y = A_loc x + A_rem x, y blocks
concatenated in rank order = global row order. The per-block arithmetic here is the ORACLE (there is no CPU product
path); what is under test is the host logic the GPU run shares: spmv_dist, spmv_host.remap_columns, the collective."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q, mode="rows"):
    for p in (os.path.join(ROOT, "spmv-research_amd", "python"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import oracle as orc
    import spmv_dist as D
    import spmv_host as H
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = H.gen_kkt(9)                                           # same matrix on every rank (deterministic)
        m = n = A["m"]
        x_orig = np.random.default_rng(14).uniform(-1, 1, n)
        part = D.graph_partition(A["row_ptr"], A["col_idx"], m, n, world, mode)
        assert part.kind == mode
        # the graph partition works on P A P^T: x and y live in the new numbering
        x = x_orig if part.perm is None else x_orig[part.perm]
        off = part.offsets
        if mode == "rows":
            np.testing.assert_array_equal(off, D.row_partition(A["row_ptr"], world))
        r0, r1 = int(off[rank]), int(off[rank + 1])
        blk = D.partition_block(A["row_ptr"], A["col_idx"], A["values"], part, rank)
        padded = D.padded_len(off)
        D.to_padded_columns(blk["col_idx"], off, padded)
        x_full = torch.zeros(world * padded, dtype=torch.float64)
        x_loc = x_full[rank * padded:(rank + 1) * padded]
        x_loc[:r1 - r0] = torch.from_numpy(x[r0:r1])
        dist.all_gather_into_tensor(x_full, x_loc)                 # in place: own slice already sits inside x_full
        xp = x_full.numpy()
        np.testing.assert_array_equal(xp, D.scatter_x_padded(x, off, padded))
        # local / remote column split of the block (what the two GPU handles hold)
        c0, c1 = rank * padded, rank * padded + (r1 - r0)
        inside = (blk["col_idx"] >= c0) & (blk["col_idx"] < c1)
        rows = np.repeat(np.arange(blk["m"]), np.diff(blk["row_ptr"]))

        def sub(mask):
            rp = np.zeros(blk["m"] + 1, np.int32)
            np.add.at(rp, rows[mask] + 1, 1)
            return np.cumsum(rp).astype(np.int32), blk["col_idx"][mask], blk["values"][mask]
        y = orc.csr_spmv(*sub(inside), xp) + orc.csr_spmv(*sub(~inside), xp)
        # trimmed exchange (grouped send/recv of only the referenced sub-ranges) must deliver every x entry the block reads
        lo, hi = D.needed_ranges(blk["col_idx"], padded, world)
        x_trim = torch.zeros(world * padded, dtype=torch.float64)
        x_trim[rank * padded:rank * padded + (r1 - r0)] = torch.from_numpy(x[r0:r1])
        ex = D.TrimmedExchange(dist, x_trim, padded, rank, world, lo, hi)
        for req in ex.start():
            req.wait()
        assert ex.recv_elems <= (world - 1) * padded
        y_trim = orc.csr_spmv(blk["row_ptr"], blk["col_idx"], blk["values"], x_trim.numpy())
        y_full = orc.csr_spmv(blk["row_ptr"], blk["col_idx"], blk["values"], xp)
        np.testing.assert_array_equal(y_trim, y_full)
        # several sub-ranges per peer (hull cut at its largest gaps): never more data than the hull, same result
        sub = D.needed_subranges(blk["col_idx"], padded, world, min_gap=8)
        x_sub = torch.zeros(world * padded, dtype=torch.float64)
        x_sub[rank * padded:rank * padded + (r1 - r0)] = torch.from_numpy(x[r0:r1])
        ex2 = D.TrimmedExchange(dist, x_sub, padded, rank, world, ranges=sub)
        for req in ex2.start():
            req.wait()
        assert ex2.recv_elems <= ex.recv_elems
        for a, b in ex2.delivered():
            np.testing.assert_array_equal(x_sub[a:b].numpy(), xp[a:b])
        np.testing.assert_array_equal(orc.csr_spmv(blk["row_ptr"], blk["col_idx"], blk["values"], x_sub.numpy()), y_full)
        ypad = torch.zeros(padded, dtype=torch.float64)             # validation only, not on the data path
        ypad[:r1 - r0] = torch.from_numpy(y)
        ys = [torch.zeros(padded, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(ys, ypad)
        if rank == 0:
            y_all = torch.cat([ys[p][:int(off[p + 1] - off[p])] for p in range(world)]).numpy()
            y_ref = orc.csr_spmv(A["row_ptr"], A["col_idx"], A["values"], x_orig)
            absrow = orc.csr_spmv(A["row_ptr"], A["col_idx"], np.abs(A["values"]), np.abs(x_orig))
            if part.perm is not None:
                y_ref, absrow = y_ref[part.perm], absrow[part.perm]
            q.put(float(np.max(np.abs(y_all - y_ref) / absrow)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rows", "graph"])
@pytest.mark.parametrize("world", [2, 3])
def test_row_partition_allgather_gloo(world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) <= 1e-12


def _worker_packed(rank, world, port, q, mode="p2p"):
    """Original-numbering layout: full-length x on every rank, packed halo exchange (spmv_dist.PackedExchange)."""
    for p in (os.path.join(ROOT, "spmv-research_amd", "python"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import oracle as orc
    import spmv_dist as D
    import spmv_host as H
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = H.gen_kkt(9)
        m = A["m"]
        x = np.random.default_rng(14).uniform(-1, 1, m)
        part = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, world, "graph")
        owner = part.owner()
        send, recv = H.halo_lists(A["row_ptr"], A["col_idx"], owner, world, rank)
        blk = D.interior_boundary_blocks(A["row_ptr"], A["col_idx"], A["values"], owner, rank)
        rows, inner, outer = blk["rows"], blk["interior"], blk["boundary"]
        x_full = torch.full((m,), float("nan"), dtype=torch.float64)         # NaN: an entry that never arrives poisons y
        x_full[torch.from_numpy(rows)] = torch.from_numpy(x[rows])
        ex = D.PackedExchange(dist, torch, x_full, send, recv, rank, world, mode)
        assert ex.recv_elems == int(part.volume[rank])
        for it in range(2):                                                  # twice: the cached op list is reusable
            reqs = ex.start()
            # interior rows while the halo is in flight: they must not need anything that has not arrived yet
            y_in = orc.csr_spmv(inner["row_ptr"], inner["col_idx"], inner["values"], x_full.numpy().copy())
            assert not np.isnan(y_in).any()
            ex.finish(reqs)
            xs = x_full.numpy()
            assert not np.isnan(xs[blk["col_idx"]]).any()
            y = np.concatenate([y_in, orc.csr_spmv(outer["row_ptr"], outer["col_idx"], outer["values"], np.nan_to_num(xs))])
        y_ref = orc.csr_spmv(A["row_ptr"], A["col_idx"], A["values"], x)[rows]
        np.testing.assert_array_equal(y, y_ref)                              # whole rows in the matrix's entry order: bit-exact
        absrow = orc.csr_spmv(A["row_ptr"], A["col_idx"], np.abs(A["values"]), np.abs(x))[rows]
        err = torch.tensor([float(np.max(np.abs(y - y_ref) / absrow))], dtype=torch.float64)
        dist.all_reduce(err, op=dist.ReduceOp.MAX)
        if rank == 0:
            q.put(float(err.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["p2p", "alltoall"])
@pytest.mark.parametrize("world", [2, 3])
def test_packed_halo_exchange_gloo(world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_packed, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) <= 1e-12


def _halo_worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "spmv-research_amd", "python"),):
        sys.path.insert(0, p)
    import spmv_dist as D
    import spmv_host as H
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N = 9
        A = H.gen_kkt(N)                                           # the whole matrix: only to CHECK what the lean path computes
        m = A["m"]
        owner = H.kkt_bfs_owner(N, world)                          # matrix-free partition == the partition of the stored matrix
        order = H.bfs_order(A["row_ptr"], A["col_idx"], m, m)
        np.testing.assert_array_equal(owner, H.owners_from_order(A["row_ptr"], order, world))
        np.testing.assert_array_equal(H.kkt_partition_volume(N, owner, world), H.partition_volume(A["row_ptr"], A["col_idx"], owner, world))
        # the rank's rows, generated on their own
        mine = np.flatnonzero(owner == rank).astype(np.int32)
        blk = H.gen_kkt_rows(N, mine)
        for k in (0, len(mine) // 2, len(mine) - 1):
            r = mine[k]
            np.testing.assert_array_equal(blk["col_idx"][blk["row_ptr"][k]:blk["row_ptr"][k + 1]], A["col_idx"][A["row_ptr"][r]:A["row_ptr"][r + 1]])
            np.testing.assert_array_equal(blk["values"][blk["row_ptr"][k]:blk["row_ptr"][k + 1]], A["values"][A["row_ptr"][r]:A["row_ptr"][r + 1]])
        # interior / boundary split from the rank's own block == the split computed from the whole matrix
        order_r, split, interior, boundary = D.split_interior_boundary(blk, owner, rank)
        ref = D.interior_boundary_blocks(A["row_ptr"], A["col_idx"], A["values"], owner, rank)
        np.testing.assert_array_equal(mine[order_r], ref["rows"])
        assert split == ref["split"]
        for got, want in ((interior, ref["interior"]), (boundary, ref["boundary"])):
            for key in ("row_ptr", "col_idx", "values"):
                np.testing.assert_array_equal(got[key], want[key])
        # halo lists: receive side from the own block, send side from ONE all_to_all of the receive lists == the lists from the whole matrix
        recv = D.recv_lists_from_block(blk, owner, rank, world)
        send = D.exchange_send_lists(dist, torch, recv, rank, world, torch.device("cpu"))
        send_ref, recv_ref = H.halo_lists(A["row_ptr"], A["col_idx"], owner, world, rank)
        for qq in range(world):
            np.testing.assert_array_equal(recv[qq], recv_ref[qq])
            np.testing.assert_array_equal(send[qq], send_ref[qq])
        q.put((rank, "ok"))
    except Exception as e:                                         # pragma: no cover
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_matrix_free_partition_and_halo_lists(world):
    """What bench.py's N > 1 setup computes from a rank's OWN rows (matrix-free partition on the analytic KKT matrix, interior /
    boundary split, receive lists, send lists through an all_to_all) equals what the whole-matrix routines give."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + world
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_subranges_from_marks_equal_subranges_from_columns():
    """A rank that builds its block piece by piece accumulates the 'touched' marks per piece (bench_multi.RowsVariant): the ranges cut
    from the marks must be the ranges cut from the whole column list."""
    import spmv_dist as D
    rng = np.random.default_rng(8)
    world, padded = 4, 100_000
    cols = np.concatenate([rng.integers(0, 3_000, 500), padded + rng.integers(40_000, 41_000, 300), padded + rng.integers(90_000, 99_000, 50),
                           3 * padded + rng.integers(0, padded, 2_000)]).astype(np.int64)
    whole = D.needed_subranges(cols, padded, world, min_gap=1 << 12)
    touched = np.zeros(world * padded, bool)
    for piece in np.array_split(cols, 7):
        touched[piece] = True
    np.testing.assert_array_equal(D.subranges_of_touched(touched, padded, world, min_gap=1 << 12), whole)
    assert whole[2].sum() == 0 and whole[1, :, 1].max() > 90_000        # peer 2 unused, peer 1 cut into its two regions
