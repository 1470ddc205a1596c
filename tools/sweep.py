#!/usr/bin/env python3
"""Sweep kernels x tunables x workloads on one GPU; prints a table and writes JSON (profiles/sweep_*.json).
Device time per launch from HIP events around back-to-back launches (spmv_mi355x_time_device)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))


def variants(dts):
    v = [("csr_scalar", {}), ("csr_vector", {}), ("csr_stream", {}), ("csr_merge", {}), ("sell_c_sigma", {}), ("coo", {})]   # auto
    for g in (2, 4, 8, 16, 32, 64):
        v.append(("csr_vector", {"lanes_per_row": g}))
    for g in (8, 16, 32):
        for rpg in (2, 4):
            v.append(("csr_vector", {"lanes_per_row": g, "rows_per_group": rpg}))
    for r in (4, 8, 16, 32, 64):
        v.append(("csr_stream", {"lanes_per_row": r}))
    for r in (32,):
        v.append(("csr_stream", {"lanes_per_row": r, "stream_mode": 2}))
    for r in (8, 16):
        v.append(("csr_stream", {"lanes_per_row": r, "stream_mode": 1}))
    for g in (8, 16, 32):
        for k in (1, 2, 4):
            v.append(("csr_stream", {"stream_mode": 4, "lanes_per_row": g, "merge_items": k}))
    for i in (5, 7, 9, 11, 13):
        v.append(("csr_merge", {"merge_items": i}))
    for c in (16, 32, 64, 256):
        v.append(("sell_c_sigma", {"sell_c": c, "sell_window": 2}))
    for sp, ng in ((4, 1), (4, 2), (4, 4), (2, 4), (2, 8), (1, 8), (1, 16)):
        v.append(("sell_c_sigma", {"sell_window": 1, "sell_split": sp, "sell_group": ng}))
    v.append(("sell_c_sigma", {"sell_window": 2}))                                 # delta layout where the window layout would be taken
    v.append(("csr_scalar", {"kahan": 1}))
    v.append(("csr_merge", {"col_blocks": -1}))
    v.append(("csr_merge", {"col_blocks": -2}))
    v.append(("sell_c_sigma", {"sell_c": 64, "sell_delta": 2}))
    for sp in (1, 2, 4):
        v.append(("sell_c_sigma", {"sell_c": 64, "sell_split": sp}))
    for k in (2, 4, 8):
        v.append(("coo", {"merge_items": k}))
    for b in (-1, 8, 16, 32, 64):
        v.append(("coo", {"col_blocks": b}))
    return v


class _Ptr:
    def __init__(self, p):
        self.p = p

    def data_ptr(self):
        return self.p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="cant,scircuit,pwtk,soc-LiveJournal1,nlpkkt240")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--dtypes", default="f64")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--formats", default="")
    ap.add_argument("--nt", default="0", help="comma list of nontemporal settings (0 auto,1 on,2 off)")
    ap.add_argument("--remap", default="0", help="comma list of xcd_remap settings (0 auto/on, 2 off)")
    ap.add_argument("--cold", action="store_true",
                    help="GPU analogue of the reference's CLEAR_CACHES (bench_spmv.cpp:331-348): overwrite a 1 GiB buffer "
                         "(4x the 256 MiB Infinity Cache) before every launch and time single launches")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import spmv_host as H
    import spmv_mi355x as E
    rows = []
    flush = None
    for w in args.workloads.split(","):
        t = time.time()
        A = H.gen_named(w, args.scale)
        m, n, nnz = A["m"], A["n"], A["nnz"]
        print(f"# {w}: m={m} nnz={nnz} gen {time.time() - t:.1f}s", flush=True)
        for dts in args.dtypes.split(","):
            npd = np.float64 if dts == "f64" else np.float32
            td = torch.float64 if dts == "f64" else torch.float32
            vb = 8 if dts == "f64" else 4
            B = nnz * (vb + 4) + (m + 1) * 4 + (n + m) * vb
            x_host = np.random.default_rng(14).uniform(-1, 1, n).astype(npd)
            for fmt, o in variants(dts):
                if args.formats and fmt not in args.formats.split(","):
                    continue
                for nt in args.nt.split(","):
                    for rm in args.remap.split(","):
                        oo = dict(o)
                        if int(nt):
                            oo["nontemporal"] = int(nt)
                        if int(rm):
                            oo["xcd_remap"] = int(rm)
                        try:
                            M = E.Matrix(A["row_ptr"], A["col_idx"], A["values"], m, n, fmt, npd, **oo)
                        except Exception as e:
                            print("skip", fmt, oo, str(e)[:120])
                            continue
                        M.upload_x(x_host)                       # the handle's own, engine-placed x / y (as bench.py)
                        x, y = _Ptr(M.x_device()), _Ptr(M.y_device())
                        s = torch.cuda.current_stream().cuda_stream
                        # warm up for >= 0.25 s (clocks, caches; the reference warms GPU kernels with 1000 calls), then the
                        # median of 7 batches: single short bursts were seen to be bimodal on small matrices
                        t_w = time.time()
                        it = max(args.iters, 20)
                        while time.time() - t_w < 0.25:
                            M.time_device(x.data_ptr(), y.data_ptr(), it, s)
                        ms = float(np.median([M.time_device(x.data_ptr(), y.data_ptr(), it, s) for _ in range(7)]))
                        if args.cold:
                            if flush is None:
                                flush = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
                            cold = []
                            for rep in range(15):
                                flush.fill_(float(rep))
                                cold.append(M.time_device(x.data_ptr(), y.data_ptr(), 1, s))
                            ms = float(np.median(cold))
                        gbps = B / ms / 1e6
                        rec = dict(workload=w, dtype=dts, format=M.format_name, opts=oo, ms=ms, gbps=gbps,
                                   gflops=2 * nnz / ms / 1e6, frac=gbps / 8000, mem_ratio=M.mem_footprint / M.csr_mem_footprint)
                        rows.append(rec)
                        print(f"{w:18s} {dts} {M.format_name:30s} nt={nt} rm={rm} {ms*1e3:10.2f} us {gbps:8.1f} GB/s  {100*gbps/8000:5.1f}%  memx{rec['mem_ratio']:.2f}", flush=True)
                        M.close()
            del x, y
        del A
    if args.out:
        with open(args.out, "w") as f:
            json.dump(dict(records=rows), f, indent=1)


if __name__ == "__main__":
    main()
