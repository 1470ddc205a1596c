"""Row-block partition of y = A*x over the GPUs of one node — host logic shared by bench.py and the
world_size-2 gloo tests.

Partition: the reference's nnz-balanced contiguous row ranges (loop_partitioner_balance_prefix_sums,
lib/parallel_util.h:156-184, as csr.cpp:140 uses it per thread) applied with W = number of GPUs. Rank p owns rows
[offsets[p], offsets[p+1]) of A, the same slice of y and (square matrices) the same slice of x.

x layout: `world` slices padded to one common length, so that ONE equal-sized all_gather_into_tensor (RCCL allgather
over xGMI) fills it in place; column c owned by part p is renumbered p*padded + (c - offsets[p]) once at conversion
time (spmv_host.remap_columns), which costs nothing per SpMV.
"""
import numpy as np

import spmv_host as H


def row_partition(row_ptr, world):
    """offsets[world+1]: nnz-balanced contiguous row blocks."""
    m = len(row_ptr) - 1
    total = int(row_ptr[m]) - int(row_ptr[0])
    offsets = np.zeros(world + 1, np.int64)
    for p in range(world):
        s, e = H.partition_prefix_sums(world, p, row_ptr, m, total)
        offsets[p], offsets[p + 1] = s, e
    offsets[0], offsets[world] = 0, m
    assert np.all(np.diff(offsets) >= 0)
    return offsets


class Partition:
    """Which rank owns which rows/x entries. kind "rows": the reference's contiguous nnz-balanced row blocks of A as it is
    (perm None). kind "graph": slabs of a breadth-first order of the matrix graph — the engine then works on P A P^T, whose
    row blocks are contiguous again: perm[new] = old, inv[old] = new; x and y live in the new numbering
    (x_new = x_old[perm], y_old[perm] = y_new). volume[p] = x entries rank p reads from its peers."""

    def __init__(self, kind, offsets, volume, perm=None):
        self.kind, self.offsets, self.volume, self.perm = kind, np.asarray(offsets, np.int64), np.asarray(volume, np.int64), perm
        self.inv = None
        self.considered = {kind: int(self.volume.max()) if len(self.volume) else 0}      # max remote x entries of each candidate looked at
        if perm is not None:
            self.inv = np.empty(len(perm), np.int32)
            self.inv[perm] = np.arange(len(perm), dtype=np.int32)


    def owner(self):
        """owner[v] of every ORIGINAL vertex v."""
        by_new = np.repeat(np.arange(len(self.offsets) - 1, dtype=np.int32), np.diff(self.offsets))
        if self.perm is None:
            return by_new
        out = np.empty(len(self.perm), np.int32)
        out[self.perm] = by_new
        return out


def graph_partition(row_ptr, col_idx, m, n, world, mode="auto"):
    """Choose the row partition of a square matrix for `world` GPUs (host/graph_partition.cpp).

    "rows"  : row_partition() — what the reference does for its threads (parallel_util.h:156-184)
    "graph" : the same nnz balance cut out of a breadth-first order; inside a rank the x entries its peers read come first
    "auto"  : whichever makes the busiest rank read fewer remote x entries (every rank computes the same answer)"""
    assert mode in ("auto", "rows", "graph")
    rows = None
    if mode in ("auto", "rows") or m != n:
        offs = row_partition(row_ptr, world)
        owner = np.repeat(np.arange(world, dtype=np.int32), np.diff(offs))
        vol = H.partition_volume(row_ptr, col_idx, owner, world) if m == n else np.zeros(world, np.int64)
        rows = Partition("rows", offs, vol)
        if mode == "rows" or m != n or world == 1:
            return rows
    order = H.bfs_order(row_ptr, col_idx, m, n)
    owner = H.owners_from_order(row_ptr, order, world)
    del order
    vol = H.partition_volume(row_ptr, col_idx, owner, world)
    if mode == "auto" and int(vol.max()) >= int(rows.volume.max()):
        rows.considered["graph"] = int(vol.max())
        return rows
    perm, offs = H.partition_layout(row_ptr, col_idx, owner, world)
    part = Partition("graph", offs, vol, perm)
    if rows is not None:
        part.considered["rows"] = int(rows.volume.max())
    return part


def partition_block(row_ptr, col_idx, values, part, rank):
    """Rows of `rank` under `part` as a local CSR (row_ptr from 0, column indices GLOBAL in the partition's numbering)."""
    if part.perm is None:
        return local_block(row_ptr, col_idx, values, part.offsets, rank)
    blk = H.permuted_block(row_ptr, col_idx, values, part.perm, part.inv, int(part.offsets[rank]), int(part.offsets[rank + 1]))
    return dict(m=blk["m"], nnz=blk["nnz"], row_ptr=blk["row_ptr"], col_idx=blk["col_idx"], values=blk["values"])


def original_block(row_ptr, col_idx, values, owner, rank):
    """Rows owned by `rank` in ascending ORIGINAL order as a local CSR whose column indices stay the ORIGINAL ones (the
    "original numbering" layout: x is a full-length buffer on every rank, see PackedExchange). Returns (block, rows)."""
    m = len(owner)
    order = np.argsort(owner, kind="stable").astype(np.int32)            # vertices grouped by owner, original order inside
    counts = np.bincount(owner, minlength=int(owner.max()) + 1 if m else 1)
    r0 = int(counts[:rank].sum())
    r1 = r0 + int(counts[rank]) if rank < len(counts) else r0
    ident = np.arange(m, dtype=np.int32)
    blk = H.permuted_block(row_ptr, col_idx, values, order, ident, r0, r1)
    return blk, order[r0:r1].copy()


def interior_boundary_blocks(row_ptr, col_idx, values, owner, rank):
    """The rank's rows for the original-numbering layout, INTERIOR rows first (every column owned by the rank: they can be
    computed while the halo is in flight), then the BOUNDARY rows (at least one column owned by a peer: computed once the
    halo has arrived) — both groups in ascending original order with original column indices, each row whole, so a row is
    summed in the matrix's own entry order whatever the number of GPUs. Returns dict(rows = original row numbers in the
    order of the rank's y, split = number of interior rows, interior = CSR, boundary = CSR)."""
    blk, rows = original_block(row_ptr, col_idx, values, owner, rank)
    lm = blk["m"]
    remote = (owner[blk["col_idx"]] != rank).astype(np.int64)
    per_row = np.zeros(lm, np.int64)
    nz = np.flatnonzero(np.diff(blk["row_ptr"]) > 0)
    if len(nz):
        per_row[nz] = np.add.reduceat(remote, blk["row_ptr"][:-1][nz])
    del remote
    order = np.argsort(per_row > 0, kind="stable")                        # interior (False) first, ascending inside a group
    split = int((per_row == 0).sum())
    if split < lm and not np.array_equal(order, np.arange(lm)):
        rows = rows[order]
        blk = H.permuted_block(row_ptr, col_idx, values, np.ascontiguousarray(rows, np.int32), np.arange(len(owner), dtype=np.int32), 0, lm)
    rp = blk["row_ptr"]
    cut = int(rp[split])
    interior = dict(m=split, nnz=cut, row_ptr=rp[:split + 1].copy(), col_idx=blk["col_idx"][:cut], values=blk["values"][:cut])
    boundary = dict(m=lm - split, nnz=int(rp[lm]) - cut, row_ptr=(rp[split:] - cut).astype(np.int32), col_idx=blk["col_idx"][cut:],
                    values=blk["values"][cut:])
    return dict(rows=rows, split=split, interior=interior, boundary=boundary, m=lm, nnz=int(rp[lm]), row_ptr=rp,
                col_idx=blk["col_idx"], values=blk["values"])


def split_by_owner(blk, owner, rank):
    """(local, remote): the block's entries whose column is owned by `rank` / by a peer, as two CSRs over the same rows
    (entry order inside a row kept) — y = A_loc x can start before the halo has arrived, y += A_rem x follows it."""
    lm = blk["m"]
    rows = np.repeat(np.arange(lm, dtype=np.int32), np.diff(blk["row_ptr"]))
    loc = owner[blk["col_idx"]] == rank
    out = []
    for mask in (loc, ~loc):
        rp = np.zeros(lm + 1, np.int64)
        np.cumsum(np.bincount(rows[mask], minlength=lm), out=rp[1:])
        out.append(dict(m=lm, nnz=int(rp[lm]), row_ptr=rp.astype(np.int32), col_idx=np.ascontiguousarray(blk["col_idx"][mask]),
                        values=np.ascontiguousarray(blk["values"][mask])))
    return out[0], out[1]


def split_interior_boundary(blk, owner, rank):
    """The rank's rows (a CSR in the rank's row order, ORIGINAL column numbers) reordered INTERIOR rows first (every column
    owned by the rank: they can be computed while the halo is in flight), then BOUNDARY rows (at least one column owned by a
    peer) — both groups keep their order and every row stays whole. Returns (order, split, interior CSR, boundary CSR):
    order[k] = index in blk of the k-th row of the new order. Pure numpy on the rank's own block: no rank ever needs more of
    the matrix than its rows."""
    lm = int(blk["m"])
    rp = np.asarray(blk["row_ptr"], np.int64)
    ci, va = blk["col_idx"], blk["values"]
    lens = np.diff(rp)
    remote = (owner[ci] != rank)
    per_row = np.zeros(lm, np.int64)
    nz = np.flatnonzero(lens > 0)
    if len(nz):
        per_row[nz] = np.add.reduceat(remote.astype(np.int64), rp[:-1][nz])
    is_b = per_row > 0
    order = np.concatenate([np.flatnonzero(~is_b), np.flatnonzero(is_b)])
    split = int((~is_b).sum())
    ent_b = np.repeat(is_b, lens)                              # entry belongs to a boundary row
    out = []
    for rows_sel, ent_sel in ((~is_b, ~ent_b), (is_b, ent_b)):
        l = lens[rows_sel]
        r = np.zeros(len(l) + 1, np.int64)
        np.cumsum(l, out=r[1:])
        out.append(dict(m=int(len(l)), nnz=int(r[-1]), row_ptr=r.astype(np.int32), col_idx=np.ascontiguousarray(ci[ent_sel]),
                        values=np.ascontiguousarray(va[ent_sel])))
    return order, split, out[0], out[1]


def recv_lists_from_block(blk, owner, rank, world):
    """recv[q] = ascending original numbers of the x entries owned by q that this rank's rows read (from its own block only)."""
    ci = blk["col_idx"]
    n = len(owner)
    touched = np.zeros(n, bool)
    touched[ci] = True
    cols = np.flatnonzero(touched)
    own = owner[cols]
    return [cols[own == q].astype(np.int32) if q != rank else np.zeros(0, np.int32) for q in range(world)]


def exchange_send_lists(dist, torch, recv, rank, world, device):
    """send[q] = what rank q asked for (its recv[rank]): one all_to_all of the list lengths, one of the lists themselves. The
    matrix-free way to halo lists: spmv_host.halo_lists needs the whole matrix on every rank, this needs only the rank's rows."""
    counts = torch.tensor([len(recv[q]) for q in range(world)], dtype=torch.int64, device=device)
    got = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(got, counts)
    got = [int(v) for v in got.cpu().numpy()]
    flat = np.concatenate([np.asarray(r, np.int64) for r in recv]) if sum(len(r) for r in recv) else np.zeros(0, np.int64)
    sendbuf = torch.from_numpy(flat).to(device)
    recvbuf = torch.zeros(sum(got), dtype=torch.int64, device=device)
    dist.all_to_all_single(recvbuf, sendbuf, got, [len(recv[q]) for q in range(world)])
    arr = recvbuf.cpu().numpy().astype(np.int32)
    off = np.concatenate([[0], np.cumsum(got)])
    return [arr[off[q]:off[q + 1]] for q in range(world)]


class PackedExchange:
    """Halo exchange for the original-numbering layout: every rank keeps a full-length x in the matrix's ORIGINAL numbering
    (so its rows keep the column patterns the single-GPU format compresses), owns the entries of its vertices and receives
    only the entries its rows read from peers. Per step: gather the entries peers need into one packed buffer (one
    index_select), grouped RCCL send/recv of the per-peer segments (batch_isend_irecv), scatter what arrived to its original
    positions (one index_copy_). The index lists come from spmv_host.halo_lists — computed by every rank on its own from the
    shared matrix and owner map, ascending on both sides, so the packed segments need no header."""

    def __init__(self, dist, torch, x_full, send, recv, rank, world, mode="p2p"):
        """mode "p2p": batch_isend_irecv of the per-peer segments; "alltoall": ONE all_to_all_single with the segment lengths
        as split sizes — the same transfers (RCCL runs it as one grouped send/recv) for a fraction of the host time per
        step, which matters when a step is < 200 us of GPU time."""
        assert mode in ("p2p", "alltoall")
        self.mode = mode
        self.dist, self.torch, self.x_full, self.rank, self.world = dist, torch, x_full, rank, world
        dev = x_full.device
        cat = lambda lists: np.concatenate([np.asarray(l, np.int64) for l in lists]) if len(lists) else np.zeros(0, np.int64)
        self.send_idx = torch.from_numpy(cat(send)).to(dev)
        self.recv_idx = torch.from_numpy(cat(recv)).to(dev)
        self.sendbuf = torch.zeros(len(self.send_idx), dtype=x_full.dtype, device=dev)
        self.recvbuf = torch.zeros(len(self.recv_idx), dtype=x_full.dtype, device=dev)
        so = np.concatenate([[0], np.cumsum([len(l) for l in send])]).astype(np.int64)
        ro = np.concatenate([[0], np.cumsum([len(l) for l in recv])]).astype(np.int64)
        self.send_elems, self.recv_elems = int(so[-1]), int(ro[-1])
        self.send_splits = [int(so[q + 1] - so[q]) for q in range(world)]
        self.recv_splits = [int(ro[q + 1] - ro[q]) for q in range(world)]
        self.recv_max_from_one_peer = int(max([len(l) for l in recv] + [0]))
        # RCCL orders its transfers after the work already queued on the current stream; the gloo rehearsal backend reads a
        # device send buffer without looking at the stream, so there the pack has to be finished first
        self.sync_after_pack = x_full.is_cuda and dist.get_backend() != "nccl"
        P = dist.P2POp
        self._ops = []
        for q in range(world):
            if q == rank:
                continue
            if ro[q + 1] > ro[q]:
                self._ops.append(P(dist.irecv, self.recvbuf[ro[q]:ro[q + 1]], q))
        for q in range(world):
            if q == rank:
                continue
            if so[q + 1] > so[q]:
                self._ops.append(P(dist.isend, self.sendbuf[so[q]:so[q + 1]], q))

    def start(self):
        """Pack on the current stream and post the sends/receives; returns the requests for finish()."""
        if self.send_elems:
            self.torch.index_select(self.x_full, 0, self.send_idx, out=self.sendbuf)
            if self.sync_after_pack:
                self.torch.cuda.synchronize()
        if self.mode == "alltoall":
            return [self.dist.all_to_all_single(self.recvbuf, self.sendbuf, self.recv_splits, self.send_splits, async_op=True)]
        return self.dist.batch_isend_irecv(self._ops) if self._ops else []

    def finish(self, reqs):
        for r in reqs:
            r.wait()
        if self.recv_elems:
            self.x_full.index_copy_(0, self.recv_idx, self.recvbuf)


def padded_len(offsets, align=64):
    return int((int(np.diff(offsets).max()) + align - 1) // align * align) if len(offsets) > 1 else 0


def local_block(row_ptr, col_idx, values, offsets, rank):
    """Rows of `rank` as a local CSR (row_ptr from 0, GLOBAL column indices)."""
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    s, e = int(row_ptr[r0]), int(row_ptr[r1])
    return dict(m=r1 - r0, nnz=e - s, row_ptr=(row_ptr[r0:r1 + 1] - s).astype(np.int32),
                col_idx=np.ascontiguousarray(col_idx[s:e]).copy(), values=np.ascontiguousarray(values[s:e]).copy())


def to_padded_columns(col_idx, offsets, padded):
    """In place: global column -> position in the padded slice layout."""
    return H.remap_columns(col_idx, offsets, padded)


def padded_to_global(cols, offsets, padded):
    cols = np.asarray(cols, np.int64)
    p = cols // padded
    return offsets[p] + (cols - p * padded)


def scatter_x_padded(x_global, offsets, padded):
    """The padded x every rank holds after the allgather (host reference of the layout)."""
    world = len(offsets) - 1
    out = np.zeros(world * padded, x_global.dtype)
    for p in range(world):
        out[p * padded:p * padded + int(offsets[p + 1] - offsets[p])] = x_global[offsets[p]:offsets[p + 1]]
    return out


def needed_ranges(col_idx_padded, padded, world):
    """(lo, hi) per peer: the part of each peer's x slice this block references (hi == lo == 0: nothing)."""
    return H.column_ranges(col_idx_padded, padded, world)


MAX_RANGES = 4


def needed_subranges(col_idx_padded, padded, world, max_ranges=MAX_RANGES, min_gap=1 << 15):
    """ranges[q, r] = (lo, hi): up to `max_ranges` disjoint sub-ranges of peer q's x slice that cover every column this
    block references there (hi == lo: unused). One hull per peer is wasteful when a block touches two regions of a peer —
    the nlpkkt240 twin's first row block needs 5.0 M entries of the second half of x but their hull spans 14.2 M (a few
    constraint rows at the very end of the vector): the hull is cut at its largest gaps (>= min_gap entries)."""
    touched = np.zeros(world * padded, bool)
    touched[np.asarray(col_idx_padded, np.int64)] = True
    return subranges_of_touched(touched, padded, world, max_ranges, min_gap)


def subranges_of_touched(touched, padded, world, max_ranges=MAX_RANGES, min_gap=1 << 15):
    """needed_subranges from the marks themselves (touched[c] = the block reads padded column c): lets a caller that builds its
    block in pieces accumulate the marks piece by piece."""
    out = np.zeros((world, max_ranges, 2), np.int64)
    for q in range(world):
        pos = np.flatnonzero(touched[q * padded:(q + 1) * padded])
        if len(pos) == 0:
            continue
        gaps = np.diff(pos)
        cut = np.sort(np.argsort(gaps)[::-1][:max_ranges - 1]) if len(gaps) else np.zeros(0, np.int64)
        cut = cut[gaps[cut] >= min_gap]
        starts = np.concatenate([[pos[0]], pos[cut + 1]])
        ends = np.concatenate([pos[cut] + 1, [pos[-1] + 1]])
        out[q, :len(starts), 0] = starts
        out[q, :len(starts), 1] = ends
    return out


class TrimmedExchange:
    """x exchange that moves only what each row block reads: rank p receives from rank q up to MAX_RANGES contiguous
    sub-ranges of q's slice that cover the columns p touches — grouped RCCL send/recv (batch_isend_irecv) straight into the
    padded x buffer, no packing. Degenerates to the full allgather when every rank reads everything. The range table is
    agreed once (one small all_gather), so sends and receives always match. `lo`/`hi` (one hull per peer) or `ranges`
    (needed_subranges) describe what THIS rank needs."""

    def __init__(self, dist, x_full, padded, rank, world, lo=None, hi=None, ranges=None):
        import torch
        self.dist, self.x_full, self.padded, self.rank, self.world = dist, x_full, padded, rank, world
        if ranges is None:
            ranges = np.zeros((world, 1, 2), np.int64)
            ranges[:, 0, 0], ranges[:, 0, 1] = lo, hi
        ranges = np.ascontiguousarray(ranges, np.int64)
        self.R = ranges.shape[1]
        mine = torch.tensor(ranges.reshape(-1))
        table = [torch.zeros_like(mine) for _ in range(world)]
        dev = x_full.device if dist.get_backend() == "nccl" else torch.device("cpu")
        mine_d = mine.to(dev)
        table = [t.to(dev) for t in table]
        dist.all_gather(table, mine_d)
        self.need = np.stack([t.cpu().numpy().reshape(world, self.R, 2) for t in table])      # need[p, q, r] = (lo, hi)
        self.recv_elems = int(sum((self.need[rank, q, :, 1] - self.need[rank, q, :, 0]).sum() for q in range(world) if q != rank))
        self._ops = None

    def ops(self):
        d, P, r, pad = self.dist, self.dist.P2POp, self.rank, self.padded
        out = []
        for q in range(self.world):
            if q == r:
                continue
            for k in range(self.R):
                lo, hi = int(self.need[r, q, k, 0]), int(self.need[r, q, k, 1])
                if hi > lo:
                    out.append(P(d.irecv, self.x_full[q * pad + lo:q * pad + hi], q))
            for k in range(self.R):
                lo, hi = int(self.need[q, r, k, 0]), int(self.need[q, r, k, 1])
                if hi > lo:
                    out.append(P(d.isend, self.x_full[r * pad + lo:r * pad + hi], q))
        return out

    def start(self):
        # the op list (tensor views + peers) never changes: build it once — at 8 GPUs a step is a few hundred microseconds
        # and re-slicing the views per step is host time on the critical path
        if self._ops is None:
            self._ops = self.ops()
        return self.dist.batch_isend_irecv(self._ops) if self._ops else []

    def delivered(self):
        """[(a, b)] absolute index ranges of x_full this rank receives (for validation)."""
        return [(q * self.padded + int(self.need[self.rank, q, k, 0]), q * self.padded + int(self.need[self.rank, q, k, 1]))
                for q in range(self.world) if q != self.rank for k in range(self.R)
                if self.need[self.rank, q, k, 1] > self.need[self.rank, q, k, 0]]


class DistributedSolver:
    """Row-partitioned Jacobi-PCG / BiCGSTAB (the multi-GPU form of bench_cg.cpp / bench_bicg.cpp, SURVEY §8 rows e + f3).

    The solver itself is the device-resident C-ABI one (csrc/solvers.hip); this class supplies what needs the communicator:
    the SpMV callback = copy of the rank's vector slice into the padded exchange buffer + in-place all_gather_into_tensor
    (RCCL over xGMI) + the local SpMV launch, and the all-reduce of the dot-product partials. `dist` is torch.distributed
    with an initialised process group (nccl on the GPUs of a node; gloo for the single-GPU rehearsal and the tests)."""

    def __init__(self, dist, torch, block, offsets, rank, world, fmt="sell_c_sigma", dtype=np.float64, **opts):
        import ctypes as C
        import spmv_mi355x as E
        self.dist, self.torch, self.E, self.C = dist, torch, E, C
        self.rank, self.world = rank, world
        self.offsets = np.asarray(offsets, np.int64)
        self.dtype = np.dtype(dtype)
        self.m = int(block["m"])
        self.block = block                                   # local CSR with GLOBAL columns: the Jacobi diagonal is read from it
        self.padded = padded_len(self.offsets)
        cols = np.ascontiguousarray(block["col_idx"]).copy()
        to_padded_columns(cols, self.offsets, self.padded)
        self.M = E.Matrix(block["row_ptr"], cols, block["values"], self.m, world * self.padded, fmt, dtype, **opts)
        td = torch.float64 if self.dtype == np.float64 else torch.float32
        self.x_full = torch.zeros(world * self.padded, dtype=td, device="cuda")
        self.x_own = self.x_full[rank * self.padded:(rank + 1) * self.padded]
        self.red = torch.zeros(8, dtype=torch.float64, device="cuda")
        self.calls = dict(spmv=0, allreduce=0)
        vb = self.dtype.itemsize

        def spmv_cb(_ctx, in_ptr, out_ptr):
            try:
                E._check(E.lib().spmv_mi355x_copy_device_async(C.c_void_p(self.x_own.data_ptr()), C.c_void_p(in_ptr),
                                                              C.c_long(self.m * vb), None))
                if world > 1:
                    dist.all_gather_into_tensor(self.x_full, self.x_own)
                self.M.spmv_device(self.x_full.data_ptr(), out_ptr, 0, 0)
                self.calls["spmv"] += 1
                return 0
            except Exception as e:                           # never let an exception cross the C frame
                self.error = e
                return 1

        def allreduce_cb(_ctx, _buf, count):
            try:
                if world > 1:
                    dist.all_reduce(self.red[:count])
                self.calls["allreduce"] += 1
                return 0
            except Exception as e:
                self.error = e
                return 1

        self.error = None
        self._cbs = (E.SPMV_CB(spmv_cb), E.ALLREDUCE_CB(allreduce_cb))      # keep the thunks alive
        self.ops = E.DistOps()
        self.ops.struct_size = C.sizeof(E.DistOps)
        self.ops.row_offset = int(self.offsets[rank])
        self.ops.spmv = self._cbs[0]
        self.ops.allreduce_sum = self._cbs[1]
        self.ops.reduce_buf_dev = self.red.data_ptr()
        self.ops.ctx = None

    def _solve(self, method, b_local, max_iterations, history):
        b = self.block
        try:
            return self.E.solve_distributed(method, self.ops, self.dtype, self.m, b["row_ptr"], b["col_idx"], b["values"], b_local,
                                            max_iterations, history)
        except Exception:
            if self.error is not None:
                raise self.error
            raise

    def pcg(self, b_local, max_iterations, history=True):
        return self._solve("pcg", b_local, max_iterations, history)

    def pbicgstab(self, b_local, max_iterations, history=True):
        return self._solve("pbicgstab", b_local, max_iterations, history)


def diagonal_only(block, rows):
    """CSR with just the FIRST stored diagonal entry of every row of `block` (rows[i] = the original number of local row i,
    columns in original numbering), renumbered so that the diagonal of local row i is column i — what the solvers' Jacobi
    extraction (solvers.hip jacobi_diagonal: first stored diagonal entry, row_offset + i) wants to see. A row without a stored
    diagonal stays empty and is reported by the solver exactly as on one GPU."""
    rp = np.asarray(block["row_ptr"], np.int64)
    lm = len(rp) - 1
    entry_row = np.repeat(np.arange(lm, dtype=np.int64), np.diff(rp))
    hits = np.flatnonzero(np.asarray(block["col_idx"], np.int64) == np.asarray(rows, np.int64)[entry_row])
    with_diag, first = np.unique(entry_row[hits], return_index=True)
    drp = np.zeros(lm + 1, np.int64)
    drp[with_diag + 1] = 1
    np.cumsum(drp, out=drp)
    return dict(m=lm, nnz=len(with_diag), row_ptr=drp.astype(np.int32), col_idx=with_diag.astype(np.int32),
                values=np.ascontiguousarray(np.asarray(block["values"])[hits[first]]))


class GraphDistributedSolver(DistributedSolver):
    """The same solvers on the communication-aware partition (graph_partition) in the original-numbering layout: rank p owns
    the vertices the partition gives it, keeps a full-length x in the matrix's original numbering, and its SpMV callback is
    scatter(own vector -> x) + packed halo exchange (PackedExchange) overlapped with the interior rows + boundary rows. The
    rank's vectors (b, x, the solver's work vectors) live in the order of `self.rows` (interior rows first, then boundary
    rows): b_local = b[self.rows], and the returned x is x_global[self.rows]."""

    def __init__(self, dist, torch, row_ptr, col_idx, values, part, rank, world, fmt="sell_c_sigma", dtype=np.float64,
                 halo="alltoall", **opts):
        import ctypes as C
        import spmv_mi355x as E
        self.dist, self.torch, self.E, self.C = dist, torch, E, C
        self.rank, self.world = rank, world
        self.dtype = np.dtype(dtype)
        owner = part.owner()
        n = len(owner)
        blocks = interior_boundary_blocks(row_ptr, col_idx, values, owner, rank)
        self.rows, self.split, self.m = blocks["rows"], blocks["split"], blocks["m"]
        self.offsets = part.offsets
        send, recv = H.halo_lists(row_ptr, col_idx, owner, world, rank)
        td = torch.float64 if self.dtype == np.float64 else torch.float32
        self.x_full = torch.zeros(n, dtype=td, device="cuda")
        self.stage = torch.zeros(max(self.m, 1), dtype=td, device="cuda")
        self.rows_dev = torch.from_numpy(np.asarray(self.rows, np.int64)).cuda()
        self.exchange = PackedExchange(dist, torch, self.x_full, send, recv, rank, world, halo)
        self.handles = [(E.Matrix(b["row_ptr"], b["col_idx"], b["values"], b["m"], n, fmt, dtype, **opts), first)
                        for b, first in ((blocks["interior"], 0), (blocks["boundary"], self.split)) if b["m"] > 0]
        self.M = self.handles[0][0] if self.handles else None
        # the C side only reads the Jacobi diagonal from the CSR it is handed: give it exactly that, numbered by local row
        self.block = diagonal_only(blocks, self.rows)
        self.red = torch.zeros(8, dtype=torch.float64, device="cuda")
        self.calls = dict(spmv=0, allreduce=0)
        vb = self.dtype.itemsize

        def spmv_cb(_ctx, in_ptr, out_ptr):
            try:
                E._check(E.lib().spmv_mi355x_copy_device_async(C.c_void_p(self.stage.data_ptr()), C.c_void_p(in_ptr),
                                                              C.c_long(self.m * vb), None))
                self.x_full.index_copy_(0, self.rows_dev, self.stage[:self.m])       # own entries to their original positions
                reqs = self.exchange.start() if world > 1 else []
                for M, first in self.handles:
                    if first == 0 and self.split > 0:                                # interior rows while the halo is in flight
                        M.spmv_device(self.x_full.data_ptr(), out_ptr, 0, 0)
                if world > 1:
                    self.exchange.finish(reqs)
                for M, first in self.handles:
                    if not (first == 0 and self.split > 0):                          # boundary rows
                        M.spmv_device(self.x_full.data_ptr(), out_ptr + first * vb, 0, 0)
                self.calls["spmv"] += 1
                return 0
            except Exception as e:                           # never let an exception cross the C frame
                self.error = e
                return 1

        def allreduce_cb(_ctx, _buf, count):
            try:
                if world > 1:
                    dist.all_reduce(self.red[:count])
                self.calls["allreduce"] += 1
                return 0
            except Exception as e:
                self.error = e
                return 1

        self.error = None
        self._cbs = (E.SPMV_CB(spmv_cb), E.ALLREDUCE_CB(allreduce_cb))
        self.ops = E.DistOps()
        self.ops.struct_size = C.sizeof(E.DistOps)
        self.ops.row_offset = 0                              # the diagonal-only CSR above is numbered by local row
        self.ops.spmv = self._cbs[0]
        self.ops.allreduce_sum = self._cbs[1]
        self.ops.reduce_buf_dev = self.red.data_ptr()
        self.ops.ctx = None
