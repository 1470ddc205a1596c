#!/usr/bin/env python3
"""The probes behind profiles/r02_placement.md, on one MI355X (gpurun -- python tools/placement_probe.py <mode>):

  windows   the headline kernel with y placed every 3 GiB inside ONE 96 GiB allocation (same handle, same x): 32 GiB steps
  pairs     five handles x five (x, y) pairs allocated one after the other: which allocation decides the time
  map       the engine's own diagnostic (SPMV_MI355X_PLACEMENT=4): value array at a*16 GiB x y at b*16+8 GiB of one 160 GiB allocation
  streams   the same handle and vectors on ten HIP streams: the state is not the stream's
  engine    what the placement pass does (SPMV_MI355X_PLACEMENT=2 log) and the time on placed against torch-allocated vectors

All modes print microseconds per SpMV of the SELL-64-sigma-delta kernel on the nlpkkt240 twin; run a mode in several processes to see
the process-to-process differences."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))
GiB = 1 << 30


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "engine"
    if mode == "map":
        os.environ["SPMV_MI355X_PLACEMENT"] = "4"
    elif mode == "engine":
        os.environ["SPMV_MI355X_PLACEMENT"] = "2"
    else:
        os.environ["SPMV_MI355X_PLACEMENT"] = "0"            # the probes place the vectors themselves
    import torch
    import spmv_host as H
    import spmv_mi355x as E
    A = H.gen_kkt(240)
    m, n = A["m"], A["n"]
    xh = np.random.default_rng(14).uniform(-1, 1, n)

    def handle():
        return E.Matrix(A["row_ptr"], A["col_idx"], A["values"], m, n, "sell_c_sigma", np.float64)

    def t(M, xp, yp, stream=0, iters=25):
        M.time_device(xp, yp, 5, stream)
        return M.time_device(xp, yp, iters, stream) * 1e3

    if mode == "windows":
        M = handle()
        x = torch.from_numpy(xh).cuda()
        arena = torch.empty(96 * GiB // 8, dtype=torch.float64, device="cuda")
        print("y at k GiB of one 96 GiB allocation -> us per SpMV")
        for k in range(0, 95, 3):
            print(f"{k:3d} {t(M, x.data_ptr(), arena.data_ptr() + k * GiB):8.1f}", flush=True)
    elif mode == "pairs":
        Hs, Xs, Ys = [], [], []
        for k in range(5):
            Hs.append(handle())
            Xs.append(torch.from_numpy(xh).cuda())
            Ys.append(torch.zeros(m + 64, dtype=torch.float64, device="cuda"))
        print("rows: handle h; columns: (x_k, y_k)")
        for h in range(5):
            print(f"h{h}: " + " ".join(f"{t(Hs[h], Xs[k].data_ptr(), Ys[k].data_ptr()):6.0f}" for k in range(5)), flush=True)
        print("handle 0; rows: x_i; columns: y_j")
        for i in range(5):
            print(f"x{i}: " + " ".join(f"{t(Hs[0], Xs[i].data_ptr(), Ys[j].data_ptr()):6.0f}" for j in range(5)), flush=True)
    elif mode == "streams":
        M = handle()
        x = torch.from_numpy(xh).cuda()
        y = torch.zeros(m + 64, dtype=torch.float64, device="cuda")
        t(M, x.data_ptr(), y.data_ptr(), 0, 300)                 # settle first: the first few hundred launches of a process run a few % slower
        streams = [("null", 0)] + [(f"s{i}", torch.cuda.Stream().cuda_stream) for i in range(8)] + [("high priority", torch.cuda.Stream(priority=-1).cuda_stream)]
        for name, sp in streams:
            print(f"stream {name:14s} {t(M, x.data_ptr(), y.data_ptr(), sp, 50):8.1f} us", flush=True)
    else:                                                        # "map" and "engine": the pass runs at the first use of the handle's vectors
        M = handle()
        M.upload_x(xh)
        own = t(M, M.x_device(), M.y_device(), 0, 100)
        x = torch.from_numpy(xh).cuda()
        y = torch.zeros(m + 64, dtype=torch.float64, device="cuda")
        ext = t(M, x.data_ptr(), y.data_ptr(), 0, 100)
        print(f"handle's own (placed) x / y: {own:.1f} us | torch-allocated x / y: {ext:.1f} us", flush=True)


if __name__ == "__main__":
    main()
