"""GPU tier at BASELINE.json's FULL sizes (synthetic twins of the five configs): the oracle cannot finish those in seconds,
so parity is checked through size-independent properties of y = A x:

  * sampled rows against the CPU oracle (bit-exact for the row-sequential kernels, tolerance for the reordering ones);
  * checksum of checksums: sum_i y_i == sum_j (column sum of A)_j * x_j, both sides accumulated in fp64 on the host;
  * x = ones gives the row sums (Q1 of SURVEY §8: the reference driver's own input);
  * linearity: A(a*x + b*z) == a*A x + b*A z;
  * determinism: two launches give identical bits — for every kernel except the column-blocked ones, whose LDS atomics add a
    row's products in a run-dependent order (they keep the tolerance bar).
Tolerances: tol * sum_j |a_ij x_j| per row with tol = 1e-12 (fp64) / 1e-5 (fp32), the bar of BASELINE.json's north_star.

Which kernels run is DERIVED from bench.py (so that what is timed is what is validated): for every workload the kernel
BASELINE.json names (bench.NAMED_KERNEL), the kernel bench.py times by default (bench.DEFAULT_FORMAT / DEFAULT_OPTS, in
bench.DEFAULT_DTYPE), and one kernel of another family.
"""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu

EXTRA = {                        # a kernel of a different family per workload, on top of the named and the default one
    "cant": [("sell_c_sigma", {}), ("csr_stream", {})],
    "scircuit": [("csr_vector", {}), ("csr_vector", {"lanes_per_row": 64}), ("csr_stream", {}), ("csr_merge", {})],
    "pwtk": [("csr_stream", {})],
    "soc-LiveJournal1": [("coo", {}), ("csr_merge", {"col_blocks": -1})],      # + the merge-balanced column-blocked layout bench.py reports under "also"
    "nlpkkt240": [],
}


def _configs():
    out = []
    for w in bench.WORKLOADS:
        dts = bench.DEFAULT_DTYPE.get(w, "f64")
        seen, kernels = set(), []
        for fmt, opts in [bench.NAMED_KERNEL[w], (bench.DEFAULT_FORMAT[w], bench.DEFAULT_OPTS.get(w, {}))] + EXTRA[w]:
            key = (fmt, tuple(sorted(opts.items())))
            if key not in seen:
                seen.add(key)
                kernels.append((fmt, dict(opts)))
        out.append((w, np.float64 if dts == "f64" else np.float32, kernels))
    return out


CONFIGS = _configs()


def _id(c):
    return c[0] + "[" + ",".join(f + "".join(f":{k}={v}" for k, v in o.items()) for f, o in c[2]) + "]"


@pytest.fixture(scope="module")
def eng():
    import spmv_mi355x as E
    return E


def test_every_baseline_config_runs_its_named_and_its_default_kernel():
    by = {c[0]: c for c in CONFIGS}
    assert set(by) == set(bench.WORKLOADS)
    assert any(f == "csr_vector" and o.get("lanes_per_row") == 64 for f, o in by["scircuit"][2])    # config 2: one wavefront per row
    assert by["pwtk"][1] == np.float32 and ("sell_c_sigma", {}) in by["pwtk"][2]   # config 3
    assert ("csr_merge", {}) in by["soc-LiveJournal1"][2]                       # config 4
    assert ("coo", {"col_blocks": -1}) in by["soc-LiveJournal1"][2]            # what bench.py times for it
    for w, _, kernels in CONFIGS:
        assert (bench.DEFAULT_FORMAT[w], bench.DEFAULT_OPTS.get(w, {})) in kernels


@pytest.mark.parametrize("workload,dtype,kernels", CONFIGS, ids=[_id(c) for c in CONFIGS])
def test_full_size_properties(eng, oracle, workload, dtype, kernels):
    import spmv_host as H
    A = H.gen_named(workload, 1.0)
    rp, ci, a, m, n, nnz = A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"], A["nnz"]
    tol = 1e-12 if dtype == np.float64 else 1e-5
    rng = np.random.default_rng(2024)
    x = rng.uniform(-1, 1, n).astype(dtype)
    z = rng.uniform(-1, 1, n).astype(dtype)
    av = a.astype(dtype)                                     # csr.cpp:72: values are narrowed to ValueType once
    # host-side invariants in fp64
    col_sum = np.bincount(ci, weights=av.astype(np.float64), minlength=n)
    col_abs = np.bincount(ci, weights=np.abs(av).astype(np.float64), minlength=n)
    lens = np.diff(rp)
    nz_rows = np.nonzero(lens)[0]
    row_sum = np.zeros(m)
    row_sum[nz_rows] = np.add.reduceat(av.astype(np.float64), rp[:-1][nz_rows])
    row_abs = np.zeros(m)
    row_abs[nz_rows] = np.add.reduceat(np.abs(av).astype(np.float64), rp[:-1][nz_rows])
    sample = np.unique(np.concatenate([rng.integers(0, m, 4000), np.argsort(lens)[-8:], [0, m - 1]]))
    # the sampled rows as their own little CSR for the oracle
    s_rp = np.zeros(len(sample) + 1, np.int64)
    np.cumsum(lens[sample], out=s_rp[1:])
    s_idx = np.concatenate([np.arange(rp[i], rp[i + 1]) for i in sample]) if len(sample) else np.zeros(0, np.int64)
    s_ci, s_a = ci[s_idx], a[s_idx]
    y_sample = oracle.csr_spmv(s_rp.astype(np.int32), s_ci, s_a, x, dtype, num_threads=1)
    abs_sample = oracle.csr_spmv(s_rp.astype(np.int32), s_ci, np.abs(s_a), np.abs(x).astype(np.float64))
    for fmt, opts in kernels:
        M = eng.Matrix(rp, ci, a, m, n, fmt, dtype, **opts)
        y = M.spmv(x)
        what = f"{workload}/{M.format_name}"
        atomics = "COOB" in M.format_name or "MERGEB" in M.format_name      # LDS atomics: run-dependent summation order
        # sampled rows vs the oracle
        err = np.abs(y[sample].astype(np.float64) - y_sample.astype(np.float64))
        assert np.all(err <= tol * abs_sample + 1e-300), f"{what}: sampled rows off by {np.max(err / np.maximum(abs_sample, 1e-300)):.3g}"
        if (M.format_name.startswith("MI355X_SELLD_64") or M.format_name.startswith("MI355X_SELLW_64")) and "_w" not in M.format_name.split("64", 1)[1]:
            assert np.array_equal(y[sample], y_sample), f"{what}: one lane per row, left to right: must be bit-exact"
        # checksum of checksums
        lhs = float(np.sum(y.astype(np.float64)))
        rhs = float(col_sum @ x.astype(np.float64))
        scale = float(col_abs @ np.abs(x).astype(np.float64))
        assert abs(lhs - rhs) <= tol * scale, f"{what}: checksum {lhs} vs {rhs} (scale {scale})"
        # x = ones -> row sums
        y1 = M.spmv(np.ones(n, dtype))
        assert np.all(np.abs(y1.astype(np.float64) - row_sum) <= tol * row_abs + 1e-300), f"{what}: row sums"
        assert np.all(y1[lens == 0] == 0), f"{what}: empty rows must be written with 0 (the driver pre-fills y with 1.0)"
        # linearity (loose by one ulp-scale factor: three roundings on the right-hand side)
        yz = M.spmv(z)
        comb = M.spmv((dtype(0.75) * x + dtype(-1.5) * z).astype(dtype))
        ref = 0.75 * y.astype(np.float64) - 1.5 * yz.astype(np.float64)
        bound_vec = np.abs(x).astype(np.float64) * 0.75 + np.abs(z).astype(np.float64) * 1.5
        # |A| (0.75|x| + 1.5|z|) bounds every partial sum involved; computed exactly enough with the engine itself
        Mabs_rows = row_abs * float(np.max(bound_vec))
        lin_tol = (4 * tol if dtype == np.float64 else 8 * tol)
        assert np.all(np.abs(comb.astype(np.float64) - ref) <= lin_tol * Mabs_rows + 1e-300), f"{what}: linearity"
        # determinism
        y_again = M.spmv(x)
        if atomics:
            assert np.all(np.abs(y_again.astype(np.float64) - y.astype(np.float64)) <= 2 * tol * row_abs * float(np.max(np.abs(x))) + 1e-300), \
                f"{what}: two launches differ by more than the summation-order tolerance"
        else:
            assert np.array_equal(y_again, y), f"{what}: two launches differ"
        M.close()
