#!/usr/bin/env python3
"""Is the pool's fast/slow timing state a property of the PROCESS or of where one allocation landed? One process builds the
headline handle several times (destroying the previous one, optionally keeping a spacer allocation of varying size alive
so that the next arrays land elsewhere) and times each."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))


def main():
    import torch
    import spmv_host as H
    import spmv_mi355x as E
    A = H.gen_named("nlpkkt240", 1.0)
    n = A["n"]
    s = torch.cuda.current_stream().cuda_stream
    x = torch.from_numpy(np.random.default_rng(14).uniform(-1, 1, n)).cuda()
    y = torch.zeros(A["m"] + 64, dtype=torch.float64, device="cuda")
    spacers = []
    for trial, spacer_mb in enumerate([0, 0, 0, 1, 37, 512, 3000, 0]):
        if spacer_mb:
            spacers.append(torch.empty(spacer_mb << 20, dtype=torch.uint8, device="cuda"))
        M = E.Matrix(A["row_ptr"], A["col_idx"], A["values"], A["m"], n, "sell_c_sigma", np.float64)
        M.time_device(x.data_ptr(), y.data_ptr(), 20, s)
        ms = [M.time_device(x.data_ptr(), y.data_ptr(), 100, s) for _ in range(3)]
        print(f"trial {trial} spacer {spacer_mb} MiB: {min(ms) * 1e3:.1f} .. {max(ms) * 1e3:.1f} us", flush=True)
        M.close()
    # fresh x / y too
    x2 = x.clone(); y2 = torch.zeros_like(y)
    M = E.Matrix(A["row_ptr"], A["col_idx"], A["values"], A["m"], n, "sell_c_sigma", np.float64)
    ms = [M.time_device(x2.data_ptr(), y2.data_ptr(), 100, s) for _ in range(3)]
    print(f"fresh x/y: {min(ms) * 1e3:.1f} .. {max(ms) * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
