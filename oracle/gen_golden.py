#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/ from the GENUINE reference build (oracle/_ref/native).

Run in the build container only (needs /root/reference to have been compiled by `make -C oracle ref`):

    python oracle/gen_golden.py

For every synthetic Matrix-Market input written below (inputs are ours; the reference ships no .mtx
file and no golden vectors) the reference's own code path
    mtx_read -> mtx_values_convert_to_real -> coo_to_csr -> csr_to_format -> spmv
is executed through oracle/ref_shim.cpp and its outputs are stored next to the input:

    tests/golden/<case>.mtx   input
    tests/golden/<case>.npz   row_ptr, col_idx, values (fp64), x_rand, and y_<backend>_<prec>_<ones|rand>;
                              symmetric files also hold the un-expanded CSR (sym_*) and y_csr_sym_<prec>_<ones|rand>
    tests/golden/manifest.json  per-case header info + the build facts the vectors depend on

Only data (inputs, outputs) is written; no reference source text.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refdrv  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SEED = 14          # the seed used in every twin parameter string of the reference (BENCH/config.sh:402-455)
THREADS = 2        # OMP threads for every reference run (SELL layout depends on it: sell_sorted.cpp:165-173)


def write(path, text):
    with open(path, "w") as f:
        f.write(text)


def fmt_val(v):
    return repr(float(v))


def coo_lines(rows, cols, vals=None, fmtv=fmt_val):
    out = []
    for k in range(len(rows)):
        if vals is None:
            out.append(f"{rows[k] + 1} {cols[k] + 1}")
        else:
            out.append(f"{rows[k] + 1} {cols[k] + 1} {fmtv(vals[k])}")
    return "\n".join(out) + "\n"


def rand_pattern(rng, m, n, nnz, lower_only=False, strict_lower=False):
    """nnz distinct coordinates (no duplicates: the reference's order among duplicates is unspecified)."""
    seen = set()
    while len(seen) < nnz:
        r = int(rng.integers(0, m))
        c = int(rng.integers(0, n))
        if lower_only and c > r:
            r, c = c, r
        if strict_lower and r == c:
            continue
        seen.add((r, c))
    coords = sorted(seen)
    perm = rng.permutation(len(coords))      # file order is NOT sorted
    rows = np.array([coords[p][0] for p in perm])
    cols = np.array([coords[p][1] for p in perm])
    return rows, cols


def make_cases(rng):
    cases = {}
    # 1. plain general real, shuffled entry order, m multiple of 16 (SELL-safe, see manifest note)
    m = 256
    r, c = rand_pattern(rng, m, m, 3000)
    v = rng.uniform(-1, 1, len(r))
    cases["general_real"] = (f"%%MatrixMarket matrix coordinate real general\n% synthetic, seed {SEED}\n"
                             f"{m} {m} {len(r)}\n" + coo_lines(r, c, v))
    # 2. symmetric real (lower triangle stored, diagonal included)
    m = 208
    r, c = rand_pattern(rng, m, m, 1500, lower_only=True)
    v = rng.uniform(-1, 1, len(r))
    cases["symmetric_real"] = (f"%%MatrixMarket matrix coordinate real symmetric\n{m} {m} {len(r)}\n" + coo_lines(r, c, v))
    # 3. skew-symmetric real (strict lower triangle)
    m = 96
    r, c = rand_pattern(rng, m, m, 400, lower_only=True, strict_lower=True)
    v = rng.uniform(-1, 1, len(r))
    cases["skew_real"] = (f"%%MatrixMarket matrix coordinate real skew-symmetric\n{m} {m} {len(r)}\n" + coo_lines(r, c, v))
    # 4. pattern general (soc-LiveJournal1 is a pattern matrix: values become 1.0)
    m = 320
    r, c = rand_pattern(rng, m, m, 2500)
    cases["pattern_general"] = (f"%%MatrixMarket matrix coordinate pattern general\n{m} {m} {len(r)}\n" + coo_lines(r, c))
    # 5. pattern symmetric
    m = 128
    r, c = rand_pattern(rng, m, m, 700, lower_only=True)
    cases["pattern_symmetric"] = (f"%%MatrixMarket matrix coordinate pattern symmetric\n{m} {m} {len(r)}\n" + coo_lines(r, c))
    # 6. integer general
    m = 112
    r, c = rand_pattern(rng, m, m, 900)
    v = rng.integers(-50, 50, len(r))
    cases["integer_general"] = (f"%%MatrixMarket matrix coordinate integer general\n{m} {m} {len(r)}\n"
                                + coo_lines(r, c, v, fmtv=lambda z: str(int(z))))
    # 7. complex general -> magnitudes
    m = 80
    r, c = rand_pattern(rng, m, m, 500)
    re, im = rng.uniform(-1, 1, len(r)), rng.uniform(-1, 1, len(r))
    body = "".join(f"{r[k] + 1} {c[k] + 1} {float(re[k])!r} {float(im[k])!r}\n" for k in range(len(r)))
    cases["complex_general"] = f"%%MatrixMarket matrix coordinate complex general\n{m} {m} {len(r)}\n" + body
    # 8. complex Hermitian
    m = 64
    r, c = rand_pattern(rng, m, m, 300, lower_only=True)
    re, im = rng.uniform(-1, 1, len(r)), rng.uniform(-1, 1, len(r))
    im = np.where(r == c, 0.0, im)
    body = "".join(f"{r[k] + 1} {c[k] + 1} {float(re[k])!r} {float(im[k])!r}\n" for k in range(len(r)))
    cases["complex_hermitian"] = f"%%MatrixMarket matrix coordinate complex Hermitian\n{m} {m} {len(r)}\n" + body
    # 9. no banner at all: silently 'coordinate real general' (Q5)
    m = 48
    r, c = rand_pattern(rng, m, m, 200)
    v = rng.uniform(-1, 1, len(r))
    cases["no_banner"] = f"% a comment instead of a banner\n{m} {m} {len(r)}\n" + coo_lines(r, c, v)
    # 10. many empty rows + comments + blank lines + odd number formats
    m = 160
    r, c = rand_pattern(rng, m, m, 300)
    keep = (r % 3 == 0)
    r, c = r[keep], c[keep]
    v = rng.uniform(-1e3, 1e3, len(r))
    lines = []
    for k in range(len(r)):
        s = ["%.17g", "%.6e", "%+.10E", "%.3f"][k % 4] % v[k]
        sep = ["  ", "\t", " ", "   "][k % 4]
        lines.append(f"{'  ' if k % 5 == 0 else ''}{r[k] + 1}{sep}{c[k] + 1}{sep}{s}")
        if k % 37 == 0:
            lines.append("")                       # blank lines are dropped by the line splitter
    cases["empty_rows_formats"] = (f"%%MatrixMarket matrix coordinate real general\n%\n% two comment lines\n\n"
                                   f"{m} {m} {len(r)}\n" + "\n".join(lines) + "\n")
    # 11. one huge row + power-law-ish short rows (load-balance stress, merge/partition edge cases)
    m = 512
    rows, cols = [], []
    for cc in range(m):
        rows.append(7); cols.append(cc)                  # full row
    rr, cc2 = rand_pattern(rng, m, m, 1200)
    seen = {(7, q) for q in range(m)}
    for a, b in zip(rr, cc2):
        if (int(a), int(b)) not in seen:
            rows.append(int(a)); cols.append(int(b)); seen.add((int(a), int(b)))
    p = rng.permutation(len(rows))
    r, c = np.array(rows)[p], np.array(cols)[p]
    v = rng.uniform(-1, 1, len(r))
    cases["huge_row"] = f"%%MatrixMarket matrix coordinate real general\n{m} {m} {len(r)}\n" + coo_lines(r, c, v)
    # 12. rectangular (m != n)
    m, n = 144, 37
    r, c = rand_pattern(rng, m, n, 800)
    v = rng.uniform(-1, 1, len(r))
    cases["rectangular"] = f"%%MatrixMarket matrix coordinate real general\n{m} {n} {len(r)}\n" + coo_lines(r, c, v)
    # 13. banded, ~27 nnz/row, symmetric (nlpkkt-like structure in miniature)
    m = 400
    rows, cols = [], []
    for i in range(m):
        for off in (0, 1, 2, 20, 21, 150):
            if i - off >= 0:
                rows.append(i); cols.append(i - off)
    r, c = np.array(rows), np.array(cols)
    v = np.where(r == c, 4.0, rng.uniform(-1, 1, len(r)))
    cases["banded_symmetric"] = f"%%MatrixMarket matrix coordinate real symmetric\n{m} {m} {len(r)}\n" + coo_lines(r, c, v)
    # 14. tiny 1x1 and a matrix with a trailing all-empty tail
    cases["tiny"] = "%%MatrixMarket matrix coordinate real general\n1 1 1\n1 1 2.5\n"
    m = 64
    r, c = rand_pattern(rng, 20, m, 90)
    v = rng.uniform(-1, 1, len(r))
    cases["empty_tail"] = f"%%MatrixMarket matrix coordinate real general\n{m} {m} {len(r)}\n" + coo_lines(r, c, v)
    # 15. pattern + skew-symmetric: dummy values are filled AFTER the expansion, so the mirrored entries are +1.0
    m = 72
    r, c = rand_pattern(rng, m, m, 260, lower_only=True, strict_lower=True)
    cases["pattern_skew"] = (f"%%MatrixMarket matrix coordinate pattern skew-symmetric\n{m} {m} {len(r)}\n" + coo_lines(r, c))
    # 16. integer symmetric with negative values (sign/width conversions)
    m = 88
    r, c = rand_pattern(rng, m, m, 420, lower_only=True)
    v = rng.integers(-2000000, 2000000, len(r))
    cases["integer_symmetric"] = (f"%%MatrixMarket matrix coordinate integer symmetric\n{m} {m} {len(r)}\n"
                                  + coo_lines(r, c, v, fmtv=lambda z: str(int(z))))
    return cases


# backends whose constructor/kernel misbehave on tiny inputs are skipped per case (reference quirks, recorded)
def sell_safe(m, threads, C):
    # sell_sorted.cpp:358-360 subtracts VEC_LEN (not 1) from the last slice index when the last thread's row range
    # is not a multiple of C, and reads out of bounds if that thread owns fewer than C slices. Keep to sizes where
    # the reference itself is well defined.
    return m % C == 0 and m >= C * C * threads


def main():
    if not refdrv.available("csr", "d", "native"):
        sys.exit("oracle/_ref/native is missing: run `make -C oracle ref` in the build container first")
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(SEED)
    cases = make_cases(rng)
    be = {}
    for name, prec in [("csr", "d"), ("csr", "f"), ("csr_kahan", "d"), ("csr_vec", "d"), ("csr_vec", "f"),
                       ("sell_sorted", "d"), ("sell_sorted", "f")]:
        be[(name, prec)] = refdrv.RefBackend(name, prec, "native", threads=THREADS)
    # symmetric storage (KEEP_SYMMETRY builds, csr_sym.cpp): ONE thread — with more the reference scatters through
    # compare-and-swap loops in a run-dependent order, so only T = 1 has a reproducible y
    # (the OpenMP runtime is shared by all the reference libraries of this process: the thread count is switched to 1
    # around the csr_sym calls only and restored, or sell_sorted's thread-dependent layout would change too)
    be_sym = {prec: refdrv.RefBackend("csr_sym", prec, "native") for prec in ("d", "f")}
    be_sym["d"].lib.ref_set_threads(THREADS)
    manifest = {"seed": SEED, "threads": THREADS, "flavour": "native (gcc -O3 -march=native, AVX-512 host)",
                "vec_len": {"d": 8, "f": 16}, "cases": {}}
    for cname, text in cases.items():
        mtx = os.path.join(OUT, cname + ".mtx")
        write(mtx, text)
        info, ia, ja, a = be[("csr", "d")].mtx_to_csr(mtx)
        m, n = info["m"], info["n"]
        xr = np.random.default_rng(SEED + 1).uniform(-1, 1, n)
        arrays = dict(row_ptr=ia, col_idx=ja, values=a, x_rand=xr)
        entry = dict(info)
        entry["backends"] = []
        for (name, prec), b in be.items():
            C = 8 if prec == "d" else 16
            if name == "sell_sorted" and not sell_safe(m, THREADS, C):
                continue
            b.csr_to_format(ia, ja, a, m, n)
            key = f"{name}_{prec}"
            arrays[f"y_{key}_ones"] = b.spmv(np.ones(n))
            arrays[f"y_{key}_rand"] = b.spmv(xr)
            entry["backends"].append(key)
            entry[f"format_name_{key}"] = b.format_name
            entry[f"mem_footprint_{key}"] = b.mem_footprint
            entry["csr_mem_footprint_" + prec] = b.csr_mem_footprint
        if info["symmetric"]:
            sinfo, sia, sja, sa = be_sym["d"].mtx_to_csr_keep_symmetry(mtx)
            arrays.update(sym_row_ptr=sia, sym_col_idx=sja, sym_values=sa)
            entry["sym_nnz"] = sinfo["nnz"]
            for prec, b in be_sym.items():
                b.lib.ref_set_threads(1)
                b.csr_to_format(sia, sja, sa, m, n, symmetric_unexpanded=True)
                arrays[f"y_csr_sym_{prec}_ones"] = b.spmv(np.ones(n))
                arrays[f"y_csr_sym_{prec}_rand"] = b.spmv(xr)
                entry[f"format_name_csr_sym_{prec}"] = b.format_name
                entry[f"mem_footprint_csr_sym_{prec}"] = b.mem_footprint
                b.lib.ref_set_threads(THREADS)
        np.savez_compressed(os.path.join(OUT, cname + ".npz"), **arrays)
        manifest["cases"][cname] = entry
        print(f"{cname:22s} m={m} n={n} nnz={info['nnz']} backends={len(entry['backends'])}")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
