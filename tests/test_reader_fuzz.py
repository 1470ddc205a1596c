"""Differential fuzz of the Matrix-Market path (SURVEY §8 rows a10 + a11): seeded random files in every field / symmetry
combination the reference accepts, with the lexical noise real files carry (comments, blank lines, tabs, '+' signs,
exponents, no banner). The product reader (`libspmv_host.so`) must agree bit for bit with the oracle restatement, and —
where the genuine reference build is present (`oracle/_ref`, built in the container, travels to the GPU box) — with the
reference's own `mtx_read -> coo_to_csr` on the resulting CSR. Files have no duplicate (row, col): the reference's
placement of duplicates inside a row depends on thread timing (csr_gen.c:178-213)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

N_FILES = 400


def _fmt_real(rng, v):
    style = rng.integers(0, 6)
    if style == 0:
        return repr(float(v))
    if style == 1:
        return f"{v:.17g}"
    if style == 2:
        return f"{v:.6E}"
    if style == 3:
        return f"{v:+.3e}"
    if style == 4:
        return f"{v:.1f}" if abs(v) < 1e6 else f"{v:.4e}"
    return f"{int(v)}." if abs(v) < 1e6 and rng.random() < 0.5 else f"{v:.12g}"


def make_file(rng):
    field = rng.choice(["real", "integer", "pattern", "complex"])
    symmetry = rng.choice(["general", "symmetric", "skew-symmetric", "Hermitian"]) if rng.random() < 0.6 else "general"
    banner = rng.random() < 0.9
    if not banner:
        field, symmetry = "real", "general"                      # matrix_market.c:171-176
    square = symmetry != "general" or rng.random() < 0.5
    m = int(rng.integers(1, 40))
    n = m if square else int(rng.integers(1, 40))
    cap = m * n if symmetry == "general" else m * (m + 1) // 2
    nnz = int(rng.integers(0, min(cap, 200) + 1))
    cells = set()
    while len(cells) < nnz:
        r, c = int(rng.integers(0, m)), int(rng.integers(0, n))
        if symmetry != "general" and c > r:
            r, c = c, r
        cells.add((r, c))
    cells = list(cells)
    rng.shuffle(cells)
    lines = []
    if banner:
        sep = "  " if rng.random() < 0.2 else " "
        lines.append(sep.join(["%%MatrixMarket", "matrix", "coordinate", field, symmetry]))
        for _ in range(int(rng.integers(0, 3))):
            lines.append("% " + "comment " * int(rng.integers(0, 4)))
    lines.append(f"{m} {n} {nnz}" if rng.random() < 0.7 else f"  {m}\t{n}   {nnz}  ")
    for r, c in cells:
        if rng.random() < 0.05:
            lines.append("")                                     # empty lines are skipped (matrix_market.c:239-240 counts non-empty)
        rs, cs = str(r + 1), str(c + 1)
        if rng.random() < 0.1:
            rs = "+" + rs
        sep = "\t" if rng.random() < 0.15 else (" " * int(rng.integers(1, 4)))
        lead = " " * int(rng.integers(0, 3))
        if field == "pattern":
            body = f"{lead}{rs}{sep}{cs}"
        elif field == "integer":
            body = f"{lead}{rs}{sep}{cs}{sep}{int(rng.integers(-10**6, 10**6))}"
        elif field == "complex":
            body = f"{lead}{rs}{sep}{cs}{sep}{_fmt_real(rng, rng.normal() * 10)}{sep}{_fmt_real(rng, rng.normal())}"
        else:
            scale = 10.0 ** rng.integers(-12, 12) if rng.random() < 0.3 else 1.0
            body = f"{lead}{rs}{sep}{cs}{sep}{_fmt_real(rng, rng.normal() * scale)}"
        lines.append(body + (" " * int(rng.integers(0, 2))))
    text = "\n".join(lines) + ("\n" if rng.random() < 0.9 else "")
    return text


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import refdrv
    if not refdrv.available("csr", "d", "native") and not refdrv.available("csr", "d", "v3"):
        return None
    flavour = "native" if refdrv.available("csr", "d", "native") else "v3"
    return refdrv.RefBackend("csr", "d", flavour, threads=2)


def test_reader_agrees_with_oracle_and_reference_on_random_files(oracle, ref, tmp_path):
    import spmv_host as H
    rng = np.random.default_rng(20260104)
    seen = set()
    for i in range(N_FILES):
        text = make_file(rng)
        path = str(tmp_path / f"f{i}.mtx")
        with open(path, "w") as f:
            f.write(text)
        hi, hr, hc, hv = H.mtx_read(path)
        oi, orr, oc, ov = oracle.mtx_read(path)
        assert {k: hi[k] for k in oi} == oi, text[:200]
        assert np.array_equal(hr, orr) and np.array_equal(hc, oc) and np.array_equal(hv, ov), text[:200]
        rp, ci, a = H.coo_to_csr(hr, hc, hv, hi["m"], hi["n"])
        orp, oci, oa = oracle.coo_to_csr(orr, oc, ov, oi["m"], oi["n"])
        assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and np.array_equal(a, oa)
        seen.add((hi["field"], hi["symmetric"], hi["skew"], hi["hermitian"]))
        if ref is not None:
            ri, rrp, rci, ra = ref.mtx_to_csr(path)
            assert (ri["m"], ri["n"], ri["nnz"], ri["symmetric"]) == (hi["m"], hi["n"], hi["nnz"], hi["symmetric"]), text[:200]
            assert (ri["nnz_diag"], ri["nnz_non_diag"]) == (hi["nnz_diag"], hi["nnz_non_diag"])
            assert np.array_equal(rrp, rp) and np.array_equal(rci, ci), text[:300]
            assert np.array_equal(ra, a), text[:300]
    assert len(seen) >= 8, f"the generator must cover the field x symmetry grid, saw {sorted(seen)}"
