#!/usr/bin/env python3
"""Randomised stress of the device-resident solvers against the oracle restatement (minutes of GPU time, not in the tiers):
random diagonally dominant systems (SPD for CG, non-symmetric for BiCGSTAB), every format, fp64."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "spmv-research_amd", "python")):
    sys.path.insert(0, p)


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    import oracle as orc
    import spmv_mi355x as eng
    orc.build()
    fmts = ["csr_scalar", "csr_vector", "csr_stream", "csr_merge", "sell_c_sigma", "coo"]
    t0 = time.time()
    for seed in range(seeds):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(50, 20000))
        dens = float(rng.uniform(2, 12)) / n
        R = sp.random(n, n, dens, random_state=int(rng.integers(1 << 30)), format="csr")
        R.data -= 0.4
        S = R + R.T
        d = np.asarray(abs(S).sum(axis=1)).ravel() * rng.uniform(1.02, 1.5) + 0.1
        A_spd = (S + sp.diags(d)).tocsr()
        A_ns = (R + sp.diags(np.asarray(abs(R).sum(axis=1)).ravel() * 1.3 + 0.1)).tocsr()
        for A in (A_spd, A_ns):
            A.sort_indices()
        b = rng.uniform(-1, 1, n)
        fmt = fmts[seed % len(fmts)]
        for name, A, iters in (("pcg", A_spd, 600), ("pbicgstab", A_ns, 120)):
            want = getattr(orc, name)(A.indptr, A.indices, A.data, b, iters)
            M = eng.Matrix(A.indptr, A.indices, A.data, n, n, fmt)
            got = getattr(M, name)(A.indptr, A.indices, A.data, b, iters)
            M.close()
            assert abs(got["iterations"] - want["iterations"]) <= max(2, 0.1 * want["iterations"]), (seed, name, got["iterations"], want["iterations"])
            err = np.linalg.norm(got["x"] - want["x"]) / max(np.linalg.norm(want["x"]), 1e-300)
            assert err <= 1e-8, (seed, name, fmt, err)
            k = min(10, got["iterations"], want["iterations"])
            assert np.allclose(got["history"][:k], want["history"][:k], rtol=1e-6), (seed, name)
        print(f"seed {seed}: n={n} {fmt} ok ({time.time() - t0:.0f}s)", flush=True)
    print("solver stress ok")


if __name__ == "__main__":
    main()
