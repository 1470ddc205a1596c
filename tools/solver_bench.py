#!/usr/bin/env python3
"""Time the device-resident solvers (row f3) against the SpMV they are built around and against the reference's call flow.

System: 27-point stencil on an N^3 grid (SPD, diagonally dominant), b = 1 (the reference's default right-hand side,
bench_cg.cpp:519-523). Reports, per format:
  * ms per CG / BiCGSTAB iteration with every vector in HBM (spmv_mi355x_pcg / _pbicgstab),
  * ms per bare SpMV launch (time_device) -> share of the iteration that is SpMV,
  * ms per iteration of the REFERENCE FLOW: host vectors, MF->spmv(host x, host y) with upload+download around every
    launch (what bench_cg.cpp does with a GPU backend) and numpy vector updates,
  * algorithmic GB/s of the whole iteration: SpMV bytes + the vector passes of the fused kernels.
Writes one JSON object to stdout (and --out).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))


def stencil27(N, diag=26.0001):
    """CSR of the 27-point stencil: `diag` on the diagonal, -1 on the up to 26 neighbours (diag > 26: strictly diagonally
    dominant -> SPD; close to 26 = a Neumann-like Laplacian that needs O(N) CG iterations); columns ascending."""
    idx = np.arange(N ** 3, dtype=np.int64)
    z, y, x = idx // (N * N), (idx // N) % N, idx % N
    counts = np.zeros(N ** 3, np.int32)
    offs = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    mask_all = np.empty((27, N ** 3), bool)
    for k, (dz, dy, dx) in enumerate(offs):
        ok = (z + dz >= 0) & (z + dz < N) & (y + dy >= 0) & (y + dy < N) & (x + dx >= 0) & (x + dx < N)
        mask_all[k] = ok
        counts += ok
    row_ptr = np.zeros(N ** 3 + 1, np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    nnz = int(row_ptr[-1])
    col = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float64)
    pos = row_ptr[:-1].copy()
    for k, (dz, dy, dx) in enumerate(offs):
        ok = mask_all[k]
        p = pos[ok]
        col[p] = (idx[ok] + (dz * N + dy) * N + dx).astype(np.int32)
        val[p] = diag if (dz, dy, dx) == (0, 0, 0) else -1.0
        pos[ok] += 1
    return row_ptr.astype(np.int32), col, val, N ** 3


def host_flow_cg(M, row_ptr, col, val, b, iters):
    """bench_cg.cpp's loop with host vectors (numpy) around Matrix_Format::spmv(host, host)."""
    m = len(b)
    diag = np.empty(m)
    for_rows = np.repeat(np.arange(m, dtype=np.int32), np.diff(row_ptr))
    d = col == for_rows
    diag[for_rows[d]] = val[d]
    x = np.zeros(m)
    r = b - M.spmv(x)
    z = r / diag
    p = z.copy()
    t0 = time.perf_counter()
    for _ in range(iters):
        Ap = M.spmv(p)
        zr = z @ r
        ak = zr / (p @ Ap)
        x += ak * p
        r -= ak * Ap
        z = r / diag
        bk = (z @ r) / zr
        p = z + bk * p
    return (time.perf_counter() - t0) / iters * 1e3, np.linalg.norm(b - M.spmv(x))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=160)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--formats", default="sell_c_sigma,csr_vector,csr_stream")
    ap.add_argument("--host-iters", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import spmv_mi355x as eng

    t0 = time.perf_counter()
    row_ptr, col, val, m = stencil27(args.grid)
    nnz = len(col)
    print(f"[solver_bench] stencil27 {args.grid}^3: m={m} nnz={nnz} generated in {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
    b = np.ones(m)
    spmv_bytes = nnz * 12 + (m + 1) * 4 + 2 * m * 8
    res = {"system": f"stencil27 {args.grid}^3", "rows": m, "nnz": nnz, "iters": args.iters, "formats": {}}
    for fmt in args.formats.split(","):
        M = eng.Matrix(row_ptr, col, val, m, m, fmt)
        xd = torch.ones(m, dtype=torch.float64, device="cuda")
        yd = torch.zeros(m, dtype=torch.float64, device="cuda")
        M.time_device(xd.data_ptr(), yd.data_ptr(), 50)
        spmv_ms = M.time_device(xd.data_ptr(), yd.data_ptr(), 200)
        del xd, yd
        out = {"format_name": M.format_name, "spmv_ms": round(spmv_ms, 5)}
        for name, fn, nspmv, passes in (("pcg", M.pcg, 1, 13), ("pbicgstab", M.pbicgstab, 2, 24)):
            fn(row_ptr, col, val, b, 10, history=False)                     # warm-up
            t = []
            for _ in range(3):
                r = fn(row_ptr, col, val, b, args.iters, history=False)
                t.append(r["seconds"])
            # subtract the fixed part (diagonal extraction, allocation, upload) measured with 0 iterations
            fixed = min(fn(row_ptr, col, val, b, 0, history=False)["seconds"] for _ in range(3))
            per_it = (min(t) - fixed) / max(r["iterations"], 1) * 1e3
            it_bytes = nspmv * spmv_bytes + passes * m * 8
            out[name] = {"iterations": r["iterations"], "error": r["error"], "seconds": round(min(t), 4),
                         "fixed_seconds": round(fixed, 4), "ms_per_iteration": round(per_it, 5),
                         "spmv_share": round(nspmv * spmv_ms / per_it, 3),
                         "algorithmic_GBps": round(it_bytes / per_it / 1e6, 1)}
            print(f"[solver_bench] {fmt} {name}: {out[name]}", file=sys.stderr, flush=True)
        if args.host_iters > 0:
            ms, err = host_flow_cg(M, row_ptr, col, val, b, args.host_iters)
            out["reference_flow_cg_ms_per_iteration"] = round(ms, 3)
            out["speedup_vs_reference_flow"] = round(ms / out["pcg"]["ms_per_iteration"], 1)
            print(f"[solver_bench] {fmt} reference flow: {ms:.2f} ms/iteration", file=sys.stderr, flush=True)
        res["formats"][fmt] = out
        M.close()
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
