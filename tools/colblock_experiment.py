#!/usr/bin/env python3
"""Would column blocking help the graph matrix (soc-LiveJournal1 twin: random gathers over a 39 MB x, bound by the fabric's
sector rate)? Emulated with what exists: B handles restricted to consecutive column ranges (col_filter_mode 1), launched
back to back with y += A_b x — each launch gathers from 1/B of x, which fits the 4 MiB L2 of an XCD for B >= 16.
Prints the total time per SpMV for several B and formats next to the unblocked kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))


def main():
    import torch
    import spmv_host as H
    import spmv_mi355x as E
    A = H.gen_named("soc-LiveJournal1", 1.0)
    m, n = A["m"], A["n"]
    x = torch.from_numpy(np.random.default_rng(14).uniform(-1, 1, n)).cuda()
    y = torch.zeros(m + 64, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def timed(handles, reps=20):
        for _ in range(3):
            for k, M in enumerate(handles):
                M.spmv_device(x.data_ptr(), y.data_ptr(), 1 if k else 0, s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            for k, M in enumerate(handles):
                M.spmv_device(x.data_ptr(), y.data_ptr(), 1 if k else 0, s)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    base = E.Matrix(A["row_ptr"], A["col_idx"], A["values"], m, n, "csr_merge", np.float64)
    print(f"unblocked {base.format_name}: {timed([base]):.1f} us", flush=True)
    y_ref = y[:m].clone()
    base.close()
    for fmt in ("csr_merge", "csr_scalar", "coo", "csr_vector"):
        for B in (4, 8, 16, 32):
            edges = np.linspace(0, n, B + 1).astype(np.int64)
            hs = [E.Matrix(A["row_ptr"], A["col_idx"], A["values"], m, n, fmt, np.float64, col_begin=int(edges[b]), col_end=int(edges[b + 1]),
                           col_filter_mode=1) for b in range(B)]
            t = timed(hs)
            err = float((y[:m] - y_ref).abs().max())
            print(f"{fmt} B={B}: {t:.1f} us total ({hs[0].format_name}), max |dy| {err:.2e}", flush=True)
            for h in hs:
                h.close()


if __name__ == "__main__":
    main()
