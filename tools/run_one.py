#!/usr/bin/env python3
"""Run ONE (workload, format) a few times on the GPU — the unit that rocprofv3 passes wrap."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="nlpkkt240")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--format", default="csr_stream")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[], help="key=value")
    ap.add_argument("--jitter", type=float, default=0.0, help="fraction of rows whose off-diagonal columns are perturbed by +-jitter-span")
    ap.add_argument("--jitter-span", type=int, default=3)
    ap.add_argument("--meta", default="", help="write what ran (format name, kernel, kernel-source fingerprint) to this JSON file")
    args = ap.parse_args()
    import torch
    import spmv_host as H
    import spmv_mi355x as E
    A = H.gen_named(args.workload, args.scale)
    if args.jitter:
        H.jitter_columns(A, args.jitter, args.jitter_span)
    npd = np.float64 if args.dtype == "f64" else np.float32
    td = torch.float64 if args.dtype == "f64" else torch.float32
    opts = {k: int(v) for k, v in (o.split("=") for o in args.opt)}
    sym = opts.pop("sym", 0)
    if sym:
        # row f4: the twin's LOWER triangle as the stored triangle of a symmetric matrix — sym=1: multiplied by the symmetric-storage
        # kernel without expanding it; sym=2: the same symmetric matrix expanded, through the general path (what it is compared with)
        import scipy.sparse as sp
        Mx = sp.csr_matrix((A["values"], A["col_idx"], A["row_ptr"]), shape=(A["m"], A["m"]))
        T = sp.tril(Mx).tocsr()
        T.sort_indices()
        Ex = (T + sp.tril(Mx, -1).T).tocsr()
        Ex.sort_indices()
        src = T if sym == 1 else Ex
        A = dict(A, row_ptr=src.indptr.astype(np.int32), col_idx=src.indices.astype(np.int32), values=src.data.astype(np.float64), nnz=int(Ex.nnz))
        if sym == 1:
            opts.update(symmetric_input=1, sell_window=1)
    M = E.Matrix(A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"], args.format, npd, **opts)
    M.upload_x(np.random.default_rng(14).uniform(-1, 1, A["n"]).astype(npd))     # the handle's own, engine-placed x / y (as bench.py)
    xp, yp = M.x_device(), M.y_device()
    s = torch.cuda.current_stream().cuda_stream
    import time
    t_w = time.time()
    while time.time() - t_w < 0.25 and args.iters > 5:          # profiling passes use --iters <= 5 and skip the warm-up
        M.time_device(xp, yp, args.iters, s)
    ms = float(np.median([M.time_device(xp, yp, args.iters, s) for _ in range(5 if args.iters > 5 else 1)]))
    vb = 8 if args.dtype == "f64" else 4
    B = A["nnz"] * (vb + 4) + (A["m"] + 1) * 4 + (A["n"] + A["m"]) * vb
    print(f"{args.workload} {M.format_name} {ms*1e3:.1f} us/launch {B/ms/1e6:.1f} GB/s algorithmic_bytes={B}")
    if args.meta:
        import json
        sys.path.insert(0, ROOT)
        import bench
        os.makedirs(os.path.dirname(args.meta), exist_ok=True)
        with open(args.meta, "w") as f:
            json.dump(dict(workload=args.workload + (f":sym{sym}" if sym else ""), dtype=args.dtype, format=args.format, opts=opts, scale=args.scale, jitter=args.jitter,
                           modes_off=int(os.environ.get("SPMV_MI355X_SELL_MODES_OFF", "0")), stored_bytes_per_nnz=M.mem_footprint / max(A["nnz"], 1),
                           format_name=M.format_name,
                           kernel=M.kernel_info()["name"], kernel_src_sha=bench.kernel_source_sha(), algorithmic_bytes=B, us_per_launch=ms * 1e3), f)


if __name__ == "__main__":
    main()
