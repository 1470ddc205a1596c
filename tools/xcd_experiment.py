#!/usr/bin/env python3
"""Same box, same process: XCD tile-order variants of one (workload, format): balanced contiguous ranges vs chunked
round-robin with several chunk sizes vs identity."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="nlpkkt240")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--format", default="sell_c_sigma")
    ap.add_argument("--chunks", default="16,64,256,1024,4096")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--stencil", type=int, default=0, help="use the uniform 27-point stencil on an N^3 grid instead of a twin")
    args = ap.parse_args()
    import torch
    import spmv_host as H
    import spmv_mi355x as E
    if args.stencil:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from solver_bench import stencil27
        t0 = time.time()
        rp, ci, va, m = stencil27(args.stencil)
        A = dict(row_ptr=rp, col_idx=ci, values=va, m=m, n=m, nnz=len(ci))
        args.workload = f"stencil27_{args.stencil}"
        print(f"generated {args.workload}: m={m} nnz={len(ci)} in {time.time() - t0:.0f}s", flush=True)
    else:
        A = H.gen_named(args.workload, args.scale)
    opts = {k: int(v) for k, v in (o.split("=") for o in args.opt)}
    x = torch.from_numpy(np.random.default_rng(14).uniform(-1, 1, A["n"])).cuda()
    y = torch.zeros(A["m"] + 64, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    B = A["nnz"] * 12 + (A["m"] + 1) * 4 + (A["n"] + A["m"]) * 8
    variants = [("identity", dict(xcd_remap=2), None), ("balanced", dict(xcd_remap=1), None)]
    variants += [(f"chunk{c}", dict(xcd_remap=3), c) for c in args.chunks.split(",")]
    for name, o, chunk in variants:
        if chunk:
            os.environ["SPMV_MI355X_XCD_CHUNK"] = str(chunk)
        M = E.Matrix(A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"], args.format, np.float64, **opts, **o)
        t_w = time.time()
        while time.time() - t_w < 0.3:
            M.time_device(x.data_ptr(), y.data_ptr(), 20, s)
        ms = float(np.median([M.time_device(x.data_ptr(), y.data_ptr(), 20, s) for _ in range(7)]))
        print(f"{args.workload} {M.format_name} {name:10s} {ms*1e3:9.1f} us  {B/ms/1e6:8.1f} GB/s  {100*B/ms/1e6/8000:5.1f} %", flush=True)
        M.close()


if __name__ == "__main__":
    main()
