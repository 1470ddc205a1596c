#!/usr/bin/env python3
"""Randomised stress of every kernel variant against the oracle (not part of the default tiers: minutes of GPU time).
Seeds x matrix kinds x random shapes x all VARIANTS of tests/test_gpu_parity.py x fp64/fp32, same bars as the parity tests."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "spmv-research_amd", "python")):
    sys.path.insert(0, p)


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    import oracle as orc
    import spmv_mi355x as eng
    from test_gpu_parity import VARIANTS, synth, check, _banded
    orc.build()
    t0 = time.time()
    n_checked = 0
    for seed in range(100, 100 + seeds):
        rng = np.random.default_rng(seed)
        cases = []
        for kind in ("powerlaw", "short", "regular"):
            m = int(rng.integers(200, 40000))
            n = int(rng.integers(max(64, m // 4), 2 * m + 64))
            cases.append((kind, m, n, synth(rng, m, n, kind)))
        mb = int(rng.integers(300, 6000))
        cases.append(("banded", mb, mb, _banded(rng, mb, int(rng.integers(3, 400)), int(rng.integers(1, 7)))))
        cases.append(("one_row", 50, 9000, synth(rng, 50, 9000, "one_row")))
        for kind, m, n, (rp, ci, a) in cases:
            x = rng.uniform(-1, 1, n)
            for dtype in (np.float64, np.float32):
                y_ref = orc.csr_spmv(rp, ci, a, x, dtype, num_threads=2)
                absrow = orc.csr_spmv(rp, ci, np.abs(a), np.abs(x))
                for fmt, opts, exact in VARIANTS:
                    A = eng.Matrix(rp, ci, a, m, n, fmt, dtype, **opts)
                    check(A.spmv(x), y_ref, absrow, dtype, exact, f"seed {seed} {kind} {m}x{n} {fmt}{opts} {np.dtype(dtype).name}")
                    A.close()
                    n_checked += 1
        print(f"seed {seed}: ok ({n_checked} handle checks, {time.time() - t0:.0f}s)", flush=True)
    print("stress ok")


if __name__ == "__main__":
    main()
