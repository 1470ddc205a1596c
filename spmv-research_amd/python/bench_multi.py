"""bench.py for N > 1: row-partitioned y = A*x over the GPUs of one node, one process per GPU (torch.distributed; backend nccl =
RCCL over xGMI). A step = the x exchange, forced every step (in a solver x changes every iteration), overlapped with the part
of the rank's rows / columns that needs no remote x, then the rest. Total work is fixed as N grows ("scaling": "strong").

Two VARIANTS are timed in one invocation (K steps each, own warm-up, barrier + synchronize on both sides, MAX over ranks) and
reported under "variants"; `value` is the faster one's:
  * "rows+allgather" — BASELINE.json's scheme: the reference's nnz-balanced contiguous row blocks (lib/parallel_util.h:156-184)
    of A as it is, x as `world` equal padded slices filled by ONE in-place all_gather_into_tensor (RCCL allgather); the
    local-column part of the block runs while the allgather is in flight, the remote-column part is accumulated after it;
  * "graph+halo" — the same nnz balance cut out of a breadth-first order of the matrix graph (slabs), every rank keeps a
    full-length x in the matrix's ORIGINAL numbering and receives only the entries its rows read (packed halo: index_select ->
    one all_to_all_single -> index_copy_); interior rows run while the halo is in flight, boundary rows after it. Chosen by
    "auto" only when its busiest rank reads fewer remote x entries than under row blocks.

Setup scales with the rank's share: no rank ever builds the whole matrix. For the analytic KKT twin every rank generates only
its own rows (spmv_host.gen_kkt_block / gen_kkt_rows) and rank 0 partitions the graph matrix-free (spmv_host.kkt_bfs_owner,
~10 s, 0.35 GB) and broadcasts the owner map; halo send lists come from an all_to_all of the receive lists, not from the matrix.
"""
import os
import sys
import time

import numpy as np


def _rss(tag, rank=None):
    """Peak host memory so far (SPMV_BENCH_VERBOSE): where a rank's high-water mark comes from."""
    if os.environ.get("SPMV_BENCH_VERBOSE"):
        import resource
        print(f"[bench] rank {os.environ.get('RANK', '0')} peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / (1 << 20):.2f} GiB after {tag}",
              file=sys.stderr, flush=True)


class PeakRSS:
    """Peak resident set of THIS process over a section: the kernel's high-water mark (VmHWM) can be reset by the process itself
    (/proc/self/clear_refs <- 5), so a section's own peak can be read apart from what came before it (the gloo rehearsals stage
    every collective through host buffers, which would otherwise hide what building the matrix costs)."""

    def __init__(self):
        self.ok = True
        try:
            with open("/proc/self/clear_refs", "w") as f:
                f.write("5")
        except Exception:
            self.ok = False

    def gib(self):
        try:
            with open("/proc/self/status") as f:
                for line in f:
                    if line.startswith("VmHWM:"):
                        return int(line.split()[1]) / (1 << 20)
        except Exception:
            pass
        import resource
        return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / (1 << 20)


class VariantUnavailable(Exception):
    """Raised on EVERY rank together (the decision went through an all-reduce): this variant's exchange did not validate on this
    backend; the run goes on with the other variant."""


class Source:
    """The global matrix as far as a rank needs it: sizes, the global row_ptr, and generators for row blocks / row lists."""

    def __init__(self, H, B, workload, scale):
        self.H, self.workload = H, workload
        self.kkt = workload == "nlpkkt240" and not (os.environ.get("SPMV_MTX_DIR") and scale == 1.0 and
                                                    os.path.exists(os.path.join(os.environ["SPMV_MTX_DIR"], "nlpkkt240.mtx")))
        if self.kkt:
            self.N = B.kkt_edge(scale)
            self.row_ptr = H.gen_kkt_row_ptr(self.N)
            self.m = self.n = len(self.row_ptr) - 1
            self.nnz = int(self.row_ptr[self.m])
            self.A = None
            self.data = "synthetic"
        else:
            self.A, self.data = B.load_workload(H, workload, scale)
            self.m, self.n, self.nnz, self.row_ptr = self.A["m"], self.A["n"], self.A["nnz"], self.A["row_ptr"]

    def block(self, r0, r1):
        if self.kkt:
            return self.H.gen_kkt_rows_into(self.N, int(self.row_ptr[r1]) - int(self.row_ptr[r0]), r0=r0, count=r1 - r0)
        A = self.A
        s, e = int(A["row_ptr"][r0]), int(A["row_ptr"][r1])
        return dict(m=r1 - r0, n=self.n, nnz=e - s, row_ptr=(A["row_ptr"][r0:r1 + 1] - s).astype(np.int32),
                    col_idx=np.ascontiguousarray(A["col_idx"][s:e]).copy(), values=np.ascontiguousarray(A["values"][s:e]).copy())

    def block_filtered(self, r0, r1, col_lo, col_hi, keep_inside):
        """Rows [r0, r1) with only the columns inside / outside [col_lo, col_hi) (original numbering)."""
        if self.kkt:
            return self.H.gen_kkt_rows_filtered(self.N, col_lo, col_hi, keep_inside, r0=r0, count=r1 - r0)
        A = self.A
        s, e = int(A["row_ptr"][r0]), int(A["row_ptr"][r1])
        ci, va = A["col_idx"][s:e], A["values"][s:e]
        keep = ((ci >= col_lo) & (ci < col_hi)) == bool(keep_inside)
        row_of = np.repeat(np.arange(r1 - r0), np.diff(A["row_ptr"][r0:r1 + 1].astype(np.int64)))
        rp = np.zeros(r1 - r0 + 1, np.int64)
        np.cumsum(np.bincount(row_of[keep], minlength=r1 - r0), out=rp[1:])
        return dict(m=r1 - r0, n=self.n, nnz=int(rp[-1]), row_ptr=rp.astype(np.int32), col_idx=np.ascontiguousarray(ci[keep]),
                    values=np.ascontiguousarray(va[keep]))

    def rows(self, rows, values=True):
        """An ascending list of rows as a CSR with original column numbers; values=False: the structure alone."""
        if self.kkt:
            nnz = int((self.row_ptr[np.asarray(rows, np.int64) + 1].astype(np.int64) - self.row_ptr[np.asarray(rows, np.int64)]).sum())
            return self.H.gen_kkt_rows_into(self.N, nnz, rows=rows, values=values)
        A = self.A
        ident = np.arange(self.n, dtype=np.int32)
        return self.H.permuted_block(A["row_ptr"], A["col_idx"], A["values"], np.ascontiguousarray(rows, np.int32), ident, 0, len(rows))

    def graph_owner(self, world):
        """(owner, volume per rank) of the breadth-first slab partition — called on rank 0 only."""
        H = self.H
        if self.kkt:
            owner = H.kkt_bfs_owner(self.N, world)
            return owner, H.kkt_partition_volume(self.N, owner, world)
        A = self.A
        order = H.bfs_order(A["row_ptr"], A["col_idx"], self.m, self.n)
        owner = H.owners_from_order(A["row_ptr"], order, world)
        return owner, H.partition_volume(A["row_ptr"], A["col_idx"], owner, world)

    def rows_volume(self, offsets, world):
        owner = np.repeat(np.arange(world, dtype=np.int32), np.diff(offsets))
        if self.kkt:
            return self.H.kkt_partition_volume(self.N, owner, world)
        return self.H.partition_volume(self.A["row_ptr"], self.A["col_idx"], owner, world)


def _chunks(lens, max_nnz):
    """(i0, i1) index ranges over a list of rows with lengths `lens`, each of at most max_nnz non-zeros (at least one row);
    max_nnz <= 0: one range. A rank builds its handles piece by piece so that its host copy of the matrix never exceeds a piece."""
    n = len(lens)
    if n == 0:
        return []
    if max_nnz <= 0:
        return [(0, n)]
    pre = np.concatenate([[0], np.cumsum(np.asarray(lens, np.int64))])
    pieces = max(1, int(-(-int(pre[-1]) // max_nnz)))
    cuts = np.unique(np.concatenate([[0], np.searchsorted(pre, np.arange(1, pieces) * (pre[-1] / pieces)), [n]]))
    return [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def _streamable(c, pieces):
    """Several pieces become ONE handle through the engine's device-resident CSR stream when the format allows it (SELL-64 delta
    layout, what sell_c_sigma picks for matrices of this size); otherwise every piece becomes its own handle."""
    o = c.opts
    return (pieces > 1 and c.fmt == "sell_c_sigma" and o.get("sell_c", 64) in (0, 64) and o.get("sell_delta", 0) != 2
            and o.get("convert_on", 0) != 2 and o.get("sell_window", 0) != 1 and not c.args.piece_handles)


def _merge_samples(sa, sb):
    """Samples of the same rows taken from the local-column and the remote-column half of a block: one sample per row."""
    assert len(sa) == len(sb)
    return [(ya, np.concatenate([ca, cb]), np.concatenate([va, vb])) for (ya, ca, va), (yb, cb, vb) in zip(sa, sb) if ya == yb]


def _bcast_array(dist, torch, arr, src, device):
    t = torch.from_numpy(arr).to(device)
    dist.broadcast(t, src)
    return t.cpu().numpy()


def _max_over_ranks(dist, torch, v):
    t = torch.tensor([v], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _all_agree(dist, torch, ok):
    f = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return int(f.item()) == 1


def _time_steps(dist, torch, fn, reps, warm=3):
    for _ in range(warm):
        fn()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return _max_over_ranks(dist, torch, time.perf_counter() - t0) / reps * 1e3


def _take_samples(blk, y_first, col_map=None, count=1000, seed=1):
    """A few rows of a block kept for the sanity check (y index, global columns, values) so that the block itself can be freed."""
    lm = int(blk["m"])
    if lm == 0:
        return []
    rp, ci, va = blk["row_ptr"], blk["col_idx"], blk["values"]
    out = []
    for i in np.unique(np.random.default_rng(seed).integers(0, lm, count)):
        a, e = int(rp[i]), int(rp[i + 1])
        cols = ci[a:e].astype(np.int64)
        out.append((y_first + int(i), cols if col_map is None else col_map(cols), va[a:e].copy()))
    return out


class RowsVariant:
    """BASELINE.json's scheme: contiguous nnz-balanced row blocks, allgather(x) (or, with --exchange p2p/auto, grouped send/recv of
    only the sub-ranges of each peer's slice that the block reads), local columns overlapped, remote columns accumulated."""

    def __init__(self, ctx, exchange):
        self.ctx, self.exchange = ctx, exchange
        c = ctx
        D, E, torch, dist = c.D, c.E, c.torch, c.dist
        self.offsets = D.row_partition(c.src.row_ptr, c.world)
        self.r0, self.r1 = int(self.offsets[c.rank]), int(self.offsets[c.rank + 1])
        self.padded = D.padded_len(self.offsets)
        self.n_x = self.padded * c.world
        lm = self.r1 - self.r0
        # The block is generated and converted PIECE BY PIECE (--host-chunk-nnz), and with the overlap scheme its local-column and
        # remote-column halves are generated separately (the generator filters): the host never holds more than one half of one
        # piece. Every piece becomes its own handle writing its own rows of y.
        peak = PeakRSS()
        self.launches, self.samples = [], []              # (local-column handle, remote-column handle or None, first row of y)
        touched = np.zeros(self.n_x, bool)                # padded columns the block reads (for the trimmed exchange)
        self.t_gen = self.t_conv = 0.0
        lens = np.diff(np.asarray(c.src.row_ptr[self.r0:self.r1 + 1], np.int64))
        pieces = _chunks(lens, c.args.host_chunk_nnz)
        use_stream = _streamable(c, len(pieces))          # several pieces -> ONE handle converted from a device-resident CSR
        built = []                                        # per half: [(handle, first row)], samples
        for keep in ((1, 0) if c.args.overlap else (None,)):
            mats, smp = [], []
            st = E.CsrStream(lm, self.n_x, int(lens.sum())) if use_stream else None      # capacity: the unfiltered block bounds either half
            for k, (i0, i1) in enumerate(pieces):
                a0, a1 = self.r0 + i0, self.r0 + i1
                t0 = time.time()
                blk = c.src.block(a0, a1) if keep is None else c.src.block_filtered(a0, a1, self.r0, self.r1, keep)
                D.to_padded_columns(blk["col_idx"], self.offsets, self.padded)   # x lives as `world` slices padded to a common length
                if keep != 1:
                    touched[blk["col_idx"]] = True
                self.t_gen += time.time() - t0
                smp += _take_samples(blk, i0, self.col_map, max(2000 // len(pieces), 50), seed=1 + k)
                t0 = time.time()
                if st is not None:
                    st.append(blk["row_ptr"], blk["col_idx"], blk["values"])
                else:
                    mats.append((E.Matrix(blk["row_ptr"], blk["col_idx"], blk["values"], i1 - i0, self.n_x, c.fmt, c.np_dtype, **c.opts), i0))
                self.t_conv += time.time() - t0
                del blk
            if st is not None:
                t0 = time.time()
                mats = [(st.finish(c.fmt, c.np_dtype, **c.opts), 0)]
                self.t_conv += time.time() - t0
            built.append((mats, smp))
        if len(built) == 1:
            self.samples = built[0][1]
            self.launches = [(M, None, first) for M, first in built[0][0]]
        else:
            self.samples = _merge_samples(built[0][1], built[1][1])
            self.launches = [(ml, mr, first) for (ml, first), (mr, _f) in zip(built[0][0], built[1][0])]
        self.mats = [M for l in self.launches for M in l[:2] if M is not None]
        _rss("rows: handles built")
        self.build_peak_rss_gib = peak.gib()
        self.ranges = D.subranges_of_touched(touched, self.padded, c.world)
        del touched
        self.y_vec = self.mats[0].output_vector(lm + 64)        # placed by the engine relative to the handle's arrays (csrc/placement.hip)
        self.y = self.y_vec.torch()
        self.x_vec = self.mats[0].input_vector(self.n_x)        # ... and x the same way (a zero-copy torch view: the collectives write into it)
        self.x_full = self.x_vec.torch()
        if c.opts.get("placement") == 3:
            try:
                self.mats[0].place_arrays(self.x_vec.ptr, self.y_vec.ptr)      # ... and the local part's matrix arrays relative to the two
            except c.E.SpmvError as e:                                           # a tuning step: never the reason a rank drops out of the job
                print(f"[bench] rank {c.rank}: array placement skipped: {e}", file=sys.stderr, flush=True)
        self.y.fill_(1.0)
        self.x_loc = self.x_full[c.rank * self.padded:(c.rank + 1) * self.padded]       # in-place allgather: own slice lives inside x_full
        self._fill_own()
        # RCCL gathers in place (send buffer = own slice of the receive buffer); gloo stages through the host and wants a separate one
        self.x_send = self.x_loc if c.args.backend == "nccl" else self.x_loc.clone()
        self.use_p2p, self.exch = False, None
        self.info = {"kind": "rows"}
        self.exchange_info = {"chosen": "allgather"}
        self.lm = lm
        self._validate()

    def _fill_own(self):
        self.x_loc[:self.r1 - self.r0].copy_(self.ctx.torch.from_numpy(self.ctx.x_host[self.r0:self.r1]))

    def _validate(self):
        """One untimed exchange, checked: every rank must end up with the same padded x; then, unless the allgather was asked for
        by name, the trimmed send/recv exchange is validated and (auto) timed against it."""
        c = self.ctx
        D, torch, dist = c.D, c.torch, c.dist
        x_expect = torch.from_numpy(D.scatter_x_padded(c.x_host, self.offsets, self.padded)).cuda()
        inplace_ok = True
        try:
            dist.all_gather_into_tensor(self.x_full, self.x_send)
            torch.cuda.synchronize()
        except Exception as e:                                  # a backend that refuses the aliased buffers outright
            print(f"[bench] in-place allgather refused ({repr(e)[:120]}); using a separate send buffer", file=sys.stderr)
            inplace_ok = False
        if not _all_agree(dist, torch, inplace_ok and torch.equal(self.x_full, x_expect)):
            self.x_full.zero_()
            self._fill_own()
            self.x_send = self.x_loc.clone()
            dist.all_gather_into_tensor(self.x_full, self.x_send)
            torch.cuda.synchronize()
            if not torch.equal(self.x_full, x_expect):
                ok = False
            else:
                ok = True
            if not _all_agree(dist, torch, ok):
                raise VariantUnavailable("allgather(x) did not produce the expected padded vector")
        if self.exchange != "allgather":
            ok = True
            try:
                self.exch = D.TrimmedExchange(dist, self.x_full, self.padded, c.rank, c.world,
                                              ranges=self.ranges)
                self.x_full.zero_()
                self._fill_own()
                for r in self.exch.start():
                    r.wait()
                torch.cuda.synchronize()
                ok = all(torch.equal(self.x_full[a:b], x_expect[a:b]) for a, b in self.exch.delivered())
            except Exception as e:                      # e.g. a rehearsal backend without device send/recv
                ok = False
                self.exchange_info["p2p_error"] = repr(e)[:200]
            if _all_agree(dist, torch, ok):
                self.exchange_info["p2p_recv_fraction_of_allgather"] = round(self.exch.recv_elems / float(max(c.world - 1, 1) * self.padded), 4)
                if self.exchange == "p2p":
                    self.use_p2p = True
                else:
                    t_each = {}
                    for name, flag in (("allgather", False), ("p2p", True)):
                        self.use_p2p = flag
                        t_each[name] = _time_steps(dist, torch, self.step, 10)
                    self.use_p2p = t_each["p2p"] < t_each["allgather"]
                    self.exchange_info.update({"allgather_ms": round(t_each["allgather"], 4), "p2p_ms": round(t_each["p2p"], 4)})
                self.exchange_info["chosen"] = "p2p" if self.use_p2p else "allgather"
            if not self.use_p2p:                         # leave x_full complete for the allgather path
                self.x_full.zero_()
                self._fill_own()
                dist.all_gather_into_tensor(self.x_full, self.x_send)
                torch.cuda.synchronize()

    def _exchange(self):
        if self.use_p2p:
            return self.exch.start()
        return [self.ctx.dist.all_gather_into_tensor(self.x_full, self.x_send, async_op=True)]

    def step(self):
        c = self.ctx
        reqs = self._exchange()
        xp, yp, vb = self.x_full.data_ptr(), self.y.data_ptr(), c.vbytes
        if c.args.overlap:
            for loc, _rem, first in self.launches:
                loc.spmv_device(xp, yp + first * vb, 0, c.sp)         # local columns: only the own slice of x
            for r in reqs:
                r.wait()
            for _loc, rem, first in self.launches:
                rem.spmv_device(xp, yp + first * vb, 1, c.sp)         # remote columns, y += ...
        else:
            for r in reqs:
                r.wait()
            for M, _none, first in self.launches:
                M.spmv_device(xp, yp + first * vb, 0, c.sp)

    def comm_only(self):
        for r in self._exchange():
            r.wait()

    def kernels_only(self):
        xp, yp, vb = self.x_full.data_ptr(), self.y.data_ptr(), self.ctx.vbytes
        for loc, rem, first in self.launches:
            loc.spmv_device(xp, yp + first * vb, 0, self.ctx.sp)
            if rem is not None:
                rem.spmv_device(xp, yp + first * vb, 1, self.ctx.sp)

    def col_map(self, cols):
        p = cols // self.padded
        return self.offsets[p] + (cols - p * self.padded)

    def describe(self):
        how = "send/recv of the needed x ranges" if self.use_p2p else "allgather(x)"
        return f"row-partitioned x{self.ctx.world} (row blocks of A), RCCL {how} " + \
               ("overlapped with local columns" if self.ctx.args.overlap else "then SpMV")

    def close(self):
        for M in self.mats:
            M.close()
        self.mats, self.launches = [], []
        self.x_full = self.x_loc = self.x_send = self.y = self.exch = None
        if getattr(self, "y_vec", None) is not None:
            self.y_vec.free()
            self.y_vec = None


class GraphVariant:
    """Breadth-first slabs, x in the matrix's original numbering, packed halo exchange, interior rows overlapped."""

    def __init__(self, ctx, owner, volume, halo):
        self.ctx = ctx
        c = ctx
        D, E, H, torch, dist = c.D, c.E, c.H, c.torch, c.dist
        n = c.src.n
        t0 = time.time()
        mine = np.flatnonzero(owner == c.rank).astype(np.int32)           # the rank's rows, ascending original order
        # pass 1 on the STRUCTURE alone, piece by piece: which rows read a peer's x (boundary), which x entries arrive from whom
        my_lens = (np.asarray(c.src.row_ptr, np.int64)[mine.astype(np.int64) + 1] - np.asarray(c.src.row_ptr, np.int64)[mine])
        per_row = np.zeros(len(mine), np.int64)
        need = np.zeros(n, bool)                                          # x entries of peers that the rank's rows read
        _rss("graph: start")
        peak = PeakRSS()
        for i0, i1 in _chunks(my_lens, min(c.args.host_chunk_nnz, 16_000_000) if c.args.host_chunk_nnz > 0 else 0):
            st = c.src.rows(mine[i0:i1], values=False)
            _rss("graph: structure piece generated")
            remote = owner[st["col_idx"]] != c.rank
            _rss("graph: structure piece owner lookup")
            nz = np.flatnonzero(my_lens[i0:i1] > 0)
            if len(nz):
                per_row[i0 + nz] = np.add.reduceat(remote, st["row_ptr"][:-1][nz].astype(np.int64), dtype=np.int64)
            _rss("graph: structure piece reduceat")
            need[st["col_idx"][remote]] = True
            del st, remote
        _rss("graph: structure pass")
        self.build_peak_rss_gib = peak.gib()
        recv = [np.flatnonzero(need & (owner == q)).astype(np.int32) if q != c.rank else np.zeros(0, np.int32) for q in range(c.world)]
        del need
        is_b = per_row > 0
        rows_int, rows_bnd = mine[~is_b], mine[is_b]                      # both ascending; every row whole
        split = int(len(rows_int))
        self.rows = np.concatenate([rows_int, rows_bnd])
        self.split, self.lm = split, len(mine)
        self.samples = []
        self.t_gen = time.time() - t0
        dev = torch.device("cuda") if c.args.backend == "nccl" else torch.device("cpu")
        send = D.exchange_send_lists(dist, torch, recv, c.rank, c.world, dev)
        self.x_full = torch.zeros(n, dtype=c.t_dtype, device="cuda")
        mine_dev = torch.from_numpy(mine.astype(np.int64)).cuda()
        want = np.concatenate([c.x_host[l] for l in recv]) if sum(len(l) for l in recv) else np.zeros(0, c.np_dtype)
        self.packed = None
        self.info = {"kind": "graph", "layout": "original", "remote_x_entries_per_rank": [int(v) for v in volume],
                     "interior_rows": int(split), "boundary_rows": int(self.lm - split)}
        # one all_to_all_single per step if the backend delivers it correctly, else grouped isend/irecv — validated entry by entry
        # BEFORE anything is timed on it; every rank takes the same branch
        for xmode in (("alltoall", "p2p") if halo == "auto" else (halo,)):
            ok = True
            try:
                self.x_full.zero_()
                self.x_full[mine_dev] = torch.from_numpy(c.x_host[mine]).cuda()
                self.packed = D.PackedExchange(dist, torch, self.x_full, send, recv, c.rank, c.world, xmode)
                self.packed.finish(self.packed.start())
                torch.cuda.synchronize()
                got = self.x_full[self.packed.recv_idx].cpu().numpy()
                if not np.array_equal(got, want):
                    ok = False
                    print(f"[bench] rank {c.rank}: packed halo exchange ({xmode}) delivered {int((got != want).sum())} wrong entries "
                          f"of {len(want)}", file=sys.stderr)
            except Exception as e:
                ok = False
                self.info["packed_exchange_error_" + xmode] = repr(e)[:200]
                print(f"[bench] rank {c.rank}: packed halo exchange ({xmode}) failed: {repr(e)[:300]}", file=sys.stderr)
            if _all_agree(dist, torch, ok):
                break
        else:
            raise VariantUnavailable("the packed halo exchange did not validate on every rank")
        del mine_dev
        t0 = time.time()
        self.launches = []
        # (handle, first row of y it writes, phase: 0 = while the halo is in flight, 1 = after it has arrived); a block's host copy
        # lives only while its handle is being built
        parts = ((rows_int, 0, 0), (rows_bnd, split, 1)) if c.args.overlap else ((self.rows, 0, 1),)
        peak = PeakRSS()                                     # apart from the exchange validation above
        for k, (rws, first, phase) in enumerate(parts):
            if len(rws) == 0:
                continue
            lens = np.asarray(c.src.row_ptr, np.int64)[rws.astype(np.int64) + 1] - np.asarray(c.src.row_ptr, np.int64)[rws]
            pieces = _chunks(lens, c.args.host_chunk_nnz)
            st = E.CsrStream(len(rws), n, int(lens.sum())) if _streamable(c, len(pieces)) else None
            for j, (i0, i1) in enumerate(pieces):                 # piece by piece: the host holds one piece of the block at a time
                b = c.src.rows(rws[i0:i1])
                self.samples += _take_samples(b, first + i0, None, max(1000 // len(pieces), 50), seed=1 + 16 * k + j)
                if st is not None:
                    st.append(b["row_ptr"], b["col_idx"], b["values"])
                else:
                    self.launches.append((E.Matrix(b["row_ptr"], b["col_idx"], b["values"], b["m"], n, c.fmt, c.np_dtype, **c.opts), first + i0, phase))
                del b
            if st is not None:                                    # ONE handle for the part, converted on the GPU from the resident CSR
                self.launches.append((st.finish(c.fmt, c.np_dtype, **c.opts), first, phase))
        self.mats = [l[0] for l in self.launches]
        _rss("graph: handles built")
        self.build_peak_rss_gib = max(self.build_peak_rss_gib, peak.gib())
        self.t_conv = time.time() - t0
        self.y_vec = self.mats[0].output_vector(self.lm + 64)   # placed by the engine relative to the first block's arrays
        self.y = self.y_vec.torch()
        self.x_vec = self.mats[0].input_vector(n)               # ... and x the same way (zero-copy torch view)
        x_new = self.x_vec.torch()
        x_new.copy_(self.x_full)
        self.x_full = self.packed.x_full = x_new         # the exchange scatters into / packs from the vector it is handed
        self.y.fill_(1.0)
        self.exchange_info = {"chosen": "packed halo " + self.packed.mode, "recv_x_entries": self.packed.recv_elems,
                              "send_x_entries": self.packed.send_elems, "recv_max_from_one_peer": self.packed.recv_max_from_one_peer}

    def step(self):
        c = self.ctx
        vb = c.vbytes
        reqs = self.packed.start()                               # pack + the halo transfers
        for M, first, phase in self.launches:
            if phase == 0:                                       # interior rows: every column is owned by this rank
                M.spmv_device(self.x_full.data_ptr(), self.y.data_ptr() + first * vb, 0, c.sp)
        self.packed.finish(reqs)                                 # wait + scatter to the original positions
        for M, first, phase in self.launches:
            if phase == 1:                                       # boundary rows (or all rows without overlap)
                M.spmv_device(self.x_full.data_ptr(), self.y.data_ptr() + first * vb, 0, c.sp)

    def comm_only(self):
        self.packed.finish(self.packed.start())

    def kernels_only(self):
        for M, first, _phase in self.launches:
            M.spmv_device(self.x_full.data_ptr(), self.y.data_ptr() + first * self.ctx.vbytes, 0, self.ctx.sp)

    def col_map(self, cols):
        return cols

    def describe(self):
        how = "all_to_all" if self.packed.mode == "alltoall" else "send/recv"
        return f"row-partitioned x{self.ctx.world} (breadth-first slabs of the matrix graph, x in original numbering), RCCL packed halo {how} " + \
               ("overlapped with the interior rows" if self.ctx.args.overlap else "then SpMV")

    def close(self):
        for M in self.mats:
            M.close()
        self.mats, self.launches = [], []
        self.x_full = self.y = self.packed = None
        if getattr(self, "y_vec", None) is not None:
            self.y_vec.free()
            self.y_vec = None


class Ctx:
    pass


def _sampled_check(B, v, c):
    """Sampled rows of the rank's y against host dot products with the GLOBAL x (fp64)."""
    yh = v.y[:v.lm].cpu().numpy().astype(np.float64)
    xg = c.x_host.astype(np.float64)
    worst = 0.0
    for yi, cols, vals in v.samples:
        vals = vals.astype(c.np_dtype).astype(np.float64)
        ref = float(np.dot(vals, xg[cols]))
        den = float(np.dot(np.abs(vals), np.abs(xg[cols]))) or 1.0
        worst = max(worst, abs(ref - float(yh[yi])) / den)
    tol = 1e-12 if c.dts == "f64" else 1e-5
    if not (worst <= tol) or not np.all(yh == yh):
        raise SystemExit(f"bench sanity check failed on rank {c.rank}: sampled rows differ from the host dot products (max {worst})")
    return _max_over_ranks(c.dist, c.torch, worst)


def _measure(B, v, c, K, warmup):
    """Warm-up, breakdown (untimed), then EXACTLY K steps bracketed by barrier + synchronize; MAX over ranks."""
    torch, dist = c.torch, c.dist
    comm_ms = _time_steps(dist, torch, v.comm_only, 10)
    kern_ms = _time_steps(dist, torch, v.kernels_only, 10)
    if v.lm * c.vbytes >= (8 << 20) and c.args.idle_after_placement > 0:
        time.sleep(c.args.idle_after_placement)      # the placement search has returned its ballast: let the driver finish clearing it (bench.py)
    for _ in range(warmup):
        v.step()
    # settle, as bench.py does at N = 1 (the reference driver warms GPU kernels with 1000 untimed calls, bench_spmv.cpp:287-294; the first
    # few hundred launches after the uploads run a few % slower): batches of 20 steps until two agree within 1 %, at least 5 and at most
    # 15 of them or ~1.5 s. Every decision is taken on all-reduced times, so every rank runs the same number of steps.
    prev, spent, batches = None, 0.0, 0
    while batches < 15 and spent < 1500.0:
        tb = _time_steps(dist, torch, v.step, 20, warm=0)
        batches, spent = batches + 1, spent + 20 * tb
        if batches >= 5 and prev is not None and abs(tb - prev) <= 0.01 * tb:
            break
        prev = tb
    dist.barrier()
    torch.cuda.synchronize()
    # ONE pair of events around the K steps, as at N = 1: an event recorded between two launches keeps the second kernel's workgroups
    # from starting while the first drains — with a pair per step every step paid a whole ramp-up and drain (3.6 % of a 1.27 ms step,
    # 6.5 % at 1/8 of the size: what made the N > 1 path look slower per byte than the N = 1 path)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    compute = torch.cuda.current_stream()
    t0 = time.perf_counter()
    e0.record(compute)
    for i in range(K):
        v.step()
    e1.record(compute)
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = _max_over_ranks(dist, torch, time.perf_counter() - t0)
    stream_ms = e0.elapsed_time(e1) / K                                      # per step on the compute stream
    ms = elapsed / K * 1e3
    check = _sampled_check(B, v, c)
    lnnz = sum(M.nnz for M in v.mats)
    lm = v.lm
    return dict(ms_per_step=ms, stream_ms=stream_ms, comm_ms=comm_ms, kern_ms=kern_ms, check=check, lnnz=int(lnnz), lm=int(lm),
                format_name=v.mats[0].format_name, kernel=v.mats[0].kernel_info()["name"])


def run(args, B):
    import torch
    import torch.distributed as dist
    import spmv_dist as D
    import spmv_host as H
    import spmv_mi355x as E
    c = Ctx()
    c.args, c.torch, c.dist, c.D, c.H, c.E = args, torch, dist, D, H, E
    c.world = int(os.environ["WORLD_SIZE"])
    c.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()   # a gloo rehearsal may put several ranks on one GPU
    torch.cuda.set_device(local_rank)
    with B.QuietStdout():                      # the gloo transport announces its connections on C stdout; ours carries ONE JSON line
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
            dist.barrier()
    dev = torch.device("cuda") if args.backend == "nccl" else torch.device("cpu")
    workload = args.workload
    c.fmt = args.format or B.DEFAULT_FORMAT.get(workload, "csr_vector")
    c.dts = args.dtype or B.DEFAULT_DTYPE.get(workload, "f64")
    c.np_dtype = np.float64 if c.dts == "f64" else np.float32
    c.t_dtype = torch.float64 if c.dts == "f64" else torch.float32
    c.vbytes = 8 if c.dts == "f64" else 4
    c.opts = B.collect_opts(args, workload)
    # csrc/placement.hip: opt-in, and the bench opts in (bench.py --placement) — unless ranks share a GPU (a gloo rehearsal on one box):
    # the walks of two processes on one device race for its free memory
    c.opts.setdefault("placement", getattr(c.args, "placement", 3) if c.world <= torch.cuda.device_count() else 0)
    c.opts.setdefault("placement_budget_gib", 160)
    c.sp = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    c.src = Source(H, B, workload, args.scale)
    t_src = time.time() - t0
    _rss("imports + process group + source")
    src = c.src
    if src.m != src.n:
        raise SystemExit("row-partitioned SpMV with an x exchange assumes a square matrix (x slices follow the row blocks)")
    c.x_host = np.random.default_rng(14).uniform(-1.0, 1.0, src.n).astype(c.np_dtype)   # global x (same on every rank)

    # ---- which variants: the north-star scheme and the auto choice, unless one was asked for by name
    one = args.variants == "one" or (args.variants == "auto" and (args.partition != "auto" or args.exchange != "auto"))
    t0 = time.time()
    owner = volume = None
    considered = {}
    want_graph = args.partition in ("auto", "graph")
    if want_graph:
        # rank 0 partitions the graph (matrix-free for the KKT twin) and broadcasts the owner map; the volumes decide "auto"
        if c.rank == 0:
            owner, volume = src.graph_owner(c.world)
            rows_vol = src.rows_volume(D.row_partition(src.row_ptr, c.world), c.world)
            head = np.concatenate([volume, rows_vol]).astype(np.int64)
        else:
            owner, head = np.zeros(src.m, np.int32), np.zeros(2 * c.world, np.int64)
        owner = _bcast_array(dist, torch, owner, 0, dev)
        head = _bcast_array(dist, torch, head, 0, dev)
        volume, rows_vol = head[:c.world], head[c.world:]
        considered = {"graph": int(volume.max()), "rows": int(rows_vol.max())}
        if args.partition == "auto" and int(volume.max()) >= int(rows_vol.max()):
            want_graph = False                                       # banded / FEM matrices: contiguous row blocks read no more
    t_part = time.time() - t0
    _rss("partition")
    plan = []
    if one:
        plan.append("graph+halo" if want_graph else "rows")
    else:
        plan.append("rows+allgather")
        plan.append("graph+halo" if want_graph else "rows")
        if plan[1] == "rows" and args.exchange == "allgather":
            plan.pop()

    results, order_run, unavailable = {}, [], {}
    for name in plan:
        t0 = time.time()
        try:
            if name == "graph+halo":
                v = GraphVariant(c, owner, volume, args.halo)
            else:
                v = RowsVariant(c, "allgather" if name == "rows+allgather" else args.exchange)
        except VariantUnavailable as e:
            unavailable[name] = str(e)
            torch.cuda.empty_cache()
            continue
        v.info["considered_max_remote_x_entries"] = considered
        r = _measure(B, v, c, args.steps, args.warmup)
        r.update(parallelism=v.describe(), partition=v.info, exchange=v.exchange_info, setup_s=round(time.time() - t0, 2),
                 generate_s=round(v.t_gen, 2), convert_s=round(v.t_conv, 2), handles=len(v.mats),
                 host_peak_rss_gib_building_the_matrix=round(_max_over_ranks(dist, torch, v.build_peak_rss_gib), 2))
        key = "rows+" + v.exchange_info["chosen"] if name == "rows" else name
        if key in results:
            key += " (auto)"
        results[key] = r
        order_run.append(key)
        v.close()
        del v
        torch.cuda.empty_cache()
    if not order_run:
        raise SystemExit(f"no variant of the multi-GPU step validated its exchange: {unavailable}")
    best = min(order_run, key=lambda k: results[k]["ms_per_step"])
    r = results[best]
    ms = r["ms_per_step"]
    nnz, m, n = src.nnz, src.m, src.n
    B_alg = B.algorithmic_bytes(m, n, nnz, c.vbytes)
    B_loc = B.algorithmic_bytes(r["lm"], n, r["lnnz"], c.vbytes)
    ach = B_loc / (r["stream_ms"] * 1e-3) / 1e9
    try:
        import resource
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / (1 << 20)
    except Exception:
        rss = 0.0
    rss = _max_over_ranks(dist, torch, rss)
    wl = f"{workload} ({'synthetic twin' if src.data == 'synthetic' else src.data})" + ("" if args.scale == 1.0 else f" scale={args.scale}")

    def summary(k):
        q = results[k]
        return {"ms_per_step": round(q["ms_per_step"], 6), "value": round(2.0 * nnz / (q["ms_per_step"] * 1e-3) / 1e9, 3),
                "parallelism": q["parallelism"], "format": q["format_name"], "exchange": q["exchange"], "partition": q["partition"],
                "breakdown_ms": {"exchange_alone": round(q["comm_ms"], 4), "kernels_alone": round(q["kern_ms"], 4),
                                 "overlap_efficiency": round((q["comm_ms"] + q["kern_ms"]) / q["ms_per_step"], 3)},
                "check_max_err_over_abs_row": q["check"], "setup_s": q["setup_s"],
                "host_peak_rss_gib_building_the_matrix": q["host_peak_rss_gib_building_the_matrix"], "handles_per_rank": q["handles"]}

    result = {
        "metric": f"GFLOP/s (2*nnz/t, {'fp64' if c.dts == 'f64' else 'fp32'} SpMV y=A*x); achieved HBM GB/s and % of peak in 'roofline'",
        "value": round(2.0 * nnz / (ms * 1e-3) / 1e9, 3), "unit": "GFLOP/s", "n_gpus": c.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 6), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": c.dts,
        "data": "synthetic" if src.data == "synthetic" else src.data,
        "config": {"workload": wl, "format": r["format_name"], "rows": int(m), "cols": int(n), "nnz": int(nnz), "parallelism": r["parallelism"],
                   "variant": best},
        "hbm_gbps_algorithmic": round(B_alg / (ms * 1e-3) / 1e9, 2),
        "hbm_pct_of_peak": round(100.0 * B_alg / (ms * 1e-3) / 1e9 / (B.HBM_PEAK_GBPS * c.world), 2),
        "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": B.HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / B.HBM_PEAK_GBPS, 4),
                     "frac_algorithmic": round(ach / B.HBM_PEAK_GBPS, 4), "frac_hbm_measured": None, "traffic": None, "kernel": r["kernel"],
                     "kernel_ms": round(r["stream_ms"], 6), "algorithmic_bytes_per_launch": int(B_loc), "cache_resident": False},
        "check_max_err_over_abs_row": r["check"],
        "exchange": r["exchange"], "partition": r["partition"],
        "breakdown_ms": {"exchange_alone": round(r["comm_ms"], 4), "kernels_alone": round(r["kern_ms"], 4),
                         "overlap_efficiency": round((r["comm_ms"] + r["kern_ms"]) / ms, 3)},
        "variants": {k: summary(k) for k in order_run},
        "variants_unavailable": unavailable,
        "setup_s": {"source": round(t_src, 2), "partition": round(t_part, 2), "max_host_rss_gib_over_ranks": round(rss, 2)},
    }
    dist.barrier()
    dist.destroy_process_group()
    return result
