#!/usr/bin/env python3
"""Per-rank kernel times of a W-way partition measured one rank at a time on ONE GPU (no communicator): what each GPU of a
W-GPU run would spend in its local-column and remote-column kernels, and how many x entries it has to receive, under the
reference's row blocks and under the breadth-first graph partition. Output: one JSON line per (partition, world) with
per-rank figures and a step model  max_r( max(local_r, exchange_r) + remote_r )  at a stated link rate.
usage: python tools/partition_probe.py [--scale 1.0] [--worlds 2,8] [--format sell_c_sigma] [--out file.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))

LINK_GBPS = 50.0          # achievable per direction on one xGMI link (MI355X_MICROARCH.md: ~153 GB/s raw bidirectional per link)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="nlpkkt240")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--worlds", default="2,8")
    ap.add_argument("--format", default="sell_c_sigma")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--modes", default="rows,graph,graph-original",
                    help="rows | graph (P A P^T, padded x slices) | graph-original (same owners, x and columns in original numbering)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import spmv_dist as D
    import spmv_host as H
    import spmv_mi355x as E
    t = time.time()
    A = H.gen_named(args.workload, args.scale)
    m = n = A["m"]
    print(f"[probe] generated {m} rows {A['nnz']} nnz in {time.time() - t:.1f}s", flush=True)
    s = torch.cuda.current_stream().cuda_stream
    records = []
    for world in [int(w) for w in args.worlds.split(",")]:
        for mode in args.modes.split(","):
            t = time.time()
            original = mode in ("graph-original", "graph-original-rows")
            whole_rows = mode == "graph-original-rows"            # interior rows / boundary rows instead of local / remote columns
            part = D.graph_partition(A["row_ptr"], A["col_idx"], m, n, world, "graph" if original else mode)
            t_part = time.time() - t
            padded = D.padded_len(part.offsets)
            owner = part.owner() if original else None
            n_x = n if original else world * padded
            x = torch.from_numpy(np.random.default_rng(14).uniform(-1, 1, n_x)).cuda()
            per_rank = []
            for r in range(world):
                if original:
                    _send, _recv = H.halo_lists(A["row_ptr"], A["col_idx"], owner, world, r)
                    recv = int(sum(len(l) for l in _recv))
                    recv_max_peer = int(max(len(l) for l in _recv))
                    if whole_rows:
                        blk = D.interior_boundary_blocks(A["row_ptr"], A["col_idx"], A["values"], owner, r)
                        pair = (blk["interior"], blk["boundary"])
                    else:
                        blk, _rows = D.original_block(A["row_ptr"], A["col_idx"], A["values"], owner, r)
                        pair = D.split_by_owner(blk, owner, r)
                else:
                    blk = D.partition_block(A["row_ptr"], A["col_idx"], A["values"], part, r)
                    D.to_padded_columns(blk["col_idx"], part.offsets, padded)
                    rg = D.needed_subranges(blk["col_idx"], padded, world)
                    recv = int(sum((rg[q, :, 1] - rg[q, :, 0]).sum() for q in range(world) if q != r))
                    recv_max_peer = int(max((rg[q, :, 1] - rg[q, :, 0]).sum() for q in range(world) if q != r))
                c0, c1 = r * padded, r * padded + blk["m"]
                y = torch.zeros(blk["m"] + 64, dtype=torch.float64, device="cuda")
                ms = []
                fp = 0.0
                for fm in (1, 2):
                    if original:
                        b = pair[fm - 1]
                        M = E.Matrix(b["row_ptr"], b["col_idx"], b["values"], b["m"], n_x, args.format, np.float64)
                    else:
                        M = E.Matrix(blk["row_ptr"], blk["col_idx"], blk["values"], blk["m"], n_x, args.format, np.float64,
                                     col_begin=c0, col_end=c1, col_filter_mode=fm)
                    M.time_device(x.data_ptr(), y.data_ptr(), 5, s)
                    ms.append(float(np.median([M.time_device(x.data_ptr(), y.data_ptr(), args.iters, s) for _ in range(3)])))
                    fp += M.mem_footprint if hasattr(M, "mem_footprint") else 0.0
                    name = M.format_name
                    del M
                exch_ms = recv_max_peer * 8 / (LINK_GBPS * 1e6)
                per_rank.append(dict(rank=r, rows=int(blk["m"]), nnz=int(blk["nnz"]), recv_x_entries=recv, recv_max_from_one_peer=recv_max_peer,
                                     local_ms=round(ms[0], 4), remote_ms=round(ms[1], 4), footprint_bytes=int(fp), exchange_model_ms=round(exch_ms, 4), format=name))
                print(f"[probe] world {world} {mode} rank {r}: rows {blk['m']} nnz {blk['nnz']} recv {recv} local {ms[0]*1e3:.1f} us remote {ms[1]*1e3:.1f} us", flush=True)
                del blk, y
            step = max(max(p["local_ms"], p["exchange_model_ms"]) + p["remote_ms"] for p in per_rank)
            rec = dict(workload=args.workload, scale=args.scale, world=world, partition=mode, partition_seconds=round(t_part, 2),
                       padded_slice=padded, model_step_ms=round(step, 4), model_link_gbps=LINK_GBPS,
                       model="max over ranks of max(local kernel, largest single-peer receive / link rate) + remote kernel",
                       ranks=per_rank)
            records.append(rec)
            print(json.dumps({k: v for k, v in rec.items() if k != "ranks"}), flush=True)
            del x
    if args.out:
        json.dump(dict(records=records), open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
