"""Pinning beyond the committed fixtures: where the genuine reference build is present (oracle/_ref: compiled in the build
container from /root/reference, travels with the repo), the oracle restatements are compared with it BIT FOR BIT on seeded
random matrices — csr (fp64/fp32), csr with Kahan summation, csr_vec (VEC_LEN 8/16), csr_sym (one thread). Skipped when the
reference libraries are absent."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import refdrv  # noqa: E402

FLAVOUR = "native" if refdrv.available("csr", "d", "native") else "v3"
pytestmark = pytest.mark.skipif(not refdrv.available("csr", "d", FLAVOUR), reason="oracle/_ref not built (needs /root/reference)")
VEC_LEN = {"d": 8, "f": 16} if FLAVOUR == "native" else {"d": 4, "f": 8}


def random_csr(rng, m, n, kind):
    if kind == "uniform":
        lens = rng.integers(0, min(n, 40) + 1, m)
    elif kind == "skewed":
        lens = np.minimum((rng.pareto(1.2, m) * 3).astype(np.int64), n)
        lens[rng.integers(0, m)] = n
    else:
        lens = np.where(rng.random(m) < 0.5, 0, rng.integers(1, min(n, 9) + 1, m))
    rp = np.zeros(m + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    ci = np.concatenate([np.sort(rng.choice(n, int(L), replace=False)) for L in lens]) if rp[-1] else np.zeros(0, np.int64)
    scale = 10.0 ** rng.integers(-8, 9, int(rp[-1])) if kind == "skewed" else 1.0
    a = rng.normal(size=int(rp[-1])) * scale
    return rp.astype(np.int32), ci.astype(np.int32), a


@pytest.mark.parametrize("kind", ["uniform", "skewed", "sparse_rows"])
def test_oracle_equals_reference_build_on_random_matrices(oracle, kind):
    rng = np.random.default_rng({"uniform": 1, "skewed": 2, "sparse_rows": 3}[kind])
    backends = {(name, prec): refdrv.RefBackend(name, prec, FLAVOUR, threads=3)
                for name, prec in (("csr", "d"), ("csr", "f"), ("csr_kahan", "d"), ("csr_vec", "d"), ("csr_vec", "f"))}
    for trial in range(25):
        m, n = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        rp, ci, a = random_csr(rng, m, n, kind)
        x = rng.uniform(-1, 1, n) * (10.0 ** rng.integers(-3, 4))
        for (name, prec), b in backends.items():
            dt = np.float64 if prec == "d" else np.float32
            b.csr_to_format(rp, ci, a, m, n)
            y_ref = b.spmv(x)
            if name == "csr":
                y = oracle.csr_spmv(rp, ci, a, x, dt, num_threads=3)
            elif name == "csr_kahan":
                y = oracle.csr_kahan_spmv(rp, ci, a, x)
            else:
                y = oracle.csr_vec_spmv(rp, ci, a, x, VEC_LEN[prec], dt)
            assert np.array_equal(y, y_ref), f"{kind} trial {trial}: {name}_{prec} differs from the reference build"


def test_csr_sym_oracle_equals_reference_build_on_random_triangles(oracle):
    if not refdrv.available("csr_sym", "d", FLAVOUR):
        pytest.skip("csr_sym reference library not built")
    rng = np.random.default_rng(7)
    backends = {prec: refdrv.RefBackend("csr_sym", prec, FLAVOUR, threads=1) for prec in ("d", "f")}
    for trial in range(25):
        m = int(rng.integers(1, 250))
        rows, cols = [], []
        for i in range(m):
            k = int(rng.integers(0, min(i + 1, 12) + 1))
            c = np.sort(rng.choice(i + 1, k, replace=False))
            rows += [i] * k
            cols += c.tolist()
        rp = np.zeros(m + 1, np.int32)
        np.add.at(rp, np.asarray(rows, np.int64) + 1, 1)
        rp = np.cumsum(rp).astype(np.int32)
        ci = np.asarray(cols, np.int32)
        a = rng.normal(size=len(ci))
        x = rng.uniform(-1, 1, m)
        for prec, b in backends.items():
            b.lib.ref_set_threads(1)
            b.csr_to_format(rp, ci, a, m, m, symmetric_unexpanded=True)
            dt = np.float64 if prec == "d" else np.float32
            assert np.array_equal(oracle.csr_sym_spmv(rp, ci, a, x, dt), b.spmv(x)), f"trial {trial} csr_sym_{prec}"
