"""Solver callers of spmv(): Jacobi-preconditioned CG and BiCGSTAB.

CPU tier: the oracle restatement (oracle/solver_oracle.c; bench_cg.cpp:93-322, bench_bicg.cpp:149-459) against solver
properties. PARITY UNPINNED: the reference's two solver translation units do not compile in this image (they include
artificial_matrix_generation.h, absent from the reference tree) and the reference holds no solver fixtures, so the
oracle is checked against a dense solve and the algorithm's invariants, not against reference output.

GPU tier (-m gpu): the device-resident solvers of libspmv_mi355x.so through the C ABI against that oracle. Tolerances
(the reference's own dot products depend on the OpenMP thread count, so bit parity is not defined for this path):
  * the first HIST_ROWS rows of the per-iteration report (error, error_explicit, error_best) agree to 1e-9 relative
    (CG, fp64) / 1e-7 (BiCGSTAB, fp64);
  * iteration count at the `err < eps` break within +-2; solution within 1e-9 relative (fp64) / 2e-4 (fp32);
  * error == error_best bit for bit when the returned vector is the promoted one (same kernels on the same x).
"""
import numpy as np
import pytest
import scipy.sparse as sp

HIST_ROWS = 20


def laplace2d(k):
    T = sp.diags([-np.ones(k - 1), 4 * np.ones(k), -np.ones(k - 1)], [-1, 0, 1])
    S = sp.diags([-np.ones(k - 1), -np.ones(k - 1)], [-1, 1])
    A = (sp.kron(sp.eye(k), T) + sp.kron(S, sp.eye(k))).tocsr()
    A.sort_indices()
    return A


def random_spd(n, density, seed):
    A = sp.random(n, n, density, random_state=seed, format="csr")
    A = A + A.T
    d = np.asarray(abs(A).sum(axis=1)).ravel() + 1.0
    A = (A + sp.diags(d)).tocsr()
    A.sort_indices()
    return A


def nonsymmetric_dd(n, density, seed):
    """diagonally dominant, not symmetric: BiCGSTAB territory"""
    A = sp.random(n, n, density, random_state=seed, format="csr")
    A.data -= 0.3
    d = np.asarray(abs(A).sum(axis=1)).ravel() * 1.5 + 1.0
    A = (A + sp.diags(d)).tocsr()
    A.sort_indices()
    return A


def near_singular_neumann(n=100, delta=1e-13):
    """1-D Neumann Laplacian + delta*I: |x| ~ 1e13*|b| so the recurrence residual drifts from the true one and the CG
    restart rule of bench_cg.cpp:219-235 fires (found by search with the oracle: 4 restarts in 1000 iterations, stable
    under 1e-15 perturbations of b)."""
    d = np.ones(n) * 2
    d[0] = d[-1] = 1
    A = (sp.diags([-np.ones(n - 1), d, -np.ones(n - 1)], [-1, 0, 1]) + delta * sp.eye(n)).tocsr()
    A.sort_indices()
    b = np.ones(n) + np.random.default_rng(1).standard_normal(n)
    return A, b


SYSTEMS = {
    "laplace2d_40": lambda: laplace2d(40),
    "random_spd_2000": lambda: random_spd(2000, 0.004, 3),
    "nonsym_dd_1500": lambda: nonsymmetric_dd(1500, 0.005, 5),
}


def rhs(A, seed=11):
    return np.random.default_rng(seed).uniform(0.5, 1.5, A.shape[0])


# ------------------------------------------------------------------------------------------------ CPU tier: the oracle

@pytest.mark.parametrize("name", ["laplace2d_40", "random_spd_2000"])
def test_oracle_pcg_solves_spd_systems(oracle, name):
    A = SYSTEMS[name]()
    b = rhs(A)
    r = oracle.pcg(A.indptr, A.indices, A.data, b, 1000)
    x_ref = np.linalg.solve(A.toarray(), b)
    assert 0 < r["iterations"] < 1000, "CG must reach the err < eps break on a well-conditioned SPD system"
    assert np.linalg.norm(r["x"] - x_ref) <= 1e-12 * np.linalg.norm(x_ref)
    assert r["eps"] == pytest.approx(1e-15 * np.linalg.norm(b), rel=1e-14)
    assert r["eps_counter"] == pytest.approx(1e-7 * np.linalg.norm(b), rel=1e-14)
    h = r["history"]
    assert h.shape == (r["iterations"], 3)
    assert h[0, 0] == pytest.approx(np.linalg.norm(b), rel=1e-14)          # x0 = 0 -> r0 = b
    assert np.all(np.diff(h[:, 2]) <= 0), "error_best never grows"
    assert r["err_best"] == pytest.approx(np.linalg.norm(b - A @ r["x"]), rel=1e-6)


@pytest.mark.parametrize("name", ["laplace2d_40", "nonsym_dd_1500"])
def test_oracle_bicgstab_runs_every_iteration_and_keeps_best(oracle, name):
    A = SYSTEMS[name]()
    b = rhs(A)
    iters = 230
    r = oracle.pbicgstab(A.indptr, A.indices, A.data, b, iters)
    assert r["iterations"] == iters, "bench_bicg.cpp:319-320: the break is commented out"
    x_ref = np.linalg.solve(A.toarray(), b)
    # after convergence the recurrence divides 0/0 (the reference does too); x_best was saved at iteration 100 / 200
    assert np.linalg.norm(r["x"] - x_ref) <= 1e-10 * np.linalg.norm(x_ref)
    assert r["err_best"] <= 1e-10 * np.linalg.norm(b)


def test_oracle_restart_rule_fires_on_drifting_recurrence(oracle):
    A, b = near_singular_neumann()
    r = oracle.pcg(A.indptr, A.indices, A.data, b, 1000)
    assert r["restarts"] >= 2
    assert r["err_best"] < np.linalg.norm(b)


def test_oracle_error_paths(oracle):
    A = laplace2d(5).tolil()
    A[3, 3] = 0
    A = A.tocsr()
    A.eliminate_zeros()
    with pytest.raises(ValueError, match="zero in diagonal"):
        oracle.pcg(A.indptr, A.indices, A.data, np.ones(25), 10)
    with pytest.raises(ValueError, match="zero in diagonal"):
        oracle.pbicgstab(A.indptr, A.indices, A.data, np.ones(25), 10)
    B = sp.random(6, 9, 0.5, random_state=0, format="csr")
    with pytest.raises(ValueError, match="square"):
        oracle.pcg(B.indptr, B.indices, B.data, np.ones(9), 10)


def test_oracle_jacobi_takes_first_stored_diagonal_entry(oracle):
    # duplicates are legal in the reference's CSR (coo_to_csr keeps them): K = the FIRST (i,i) entry, bench_cg.cpp:122-129
    row_ptr = np.array([0, 3, 5], np.int32)
    col = np.array([0, 0, 1, 0, 1], np.int32)
    val = np.array([2.0, 5.0, 1.0, 1.0, 4.0])
    b = np.array([1.0, 2.0])
    r = oracle.pcg(row_ptr, col, val, b, 1)
    # one CG step by hand with K = diag(2, 4), A = [[7, 1], [1, 4]]
    A = np.array([[7.0, 1.0], [1.0, 4.0]])
    K = np.array([2.0, 4.0])
    z = b / K
    ak = (z @ b) / (z @ (A @ z))
    assert np.allclose(r["x"], ak * z, rtol=1e-15)


# ------------------------------------------------------------------------------------------------ GPU tier

FORMATS = ["csr_scalar", "csr_vector", "csr_stream", "csr_merge", "sell_c_sigma", "coo"]


def _matrix(A, fmt, dtype=np.float64, **opts):
    import spmv_mi355x as eng
    return eng.Matrix(A.indptr, A.indices, A.data, A.shape[0], A.shape[1], fmt, dtype, **opts)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("name", ["laplace2d_40", "random_spd_2000"])
def test_gpu_pcg_matches_oracle(oracle, name, fmt):
    A = SYSTEMS[name]()
    b = rhs(A)
    want = oracle.pcg(A.indptr, A.indices, A.data, b, 1000)
    M = _matrix(A, fmt)
    got = M.pcg(A.indptr, A.indices, A.data, b, 1000)
    assert abs(got["iterations"] - want["iterations"]) <= 2
    assert got["eps"] == pytest.approx(want["eps"], rel=1e-13)
    assert got["eps_counter"] == pytest.approx(want["eps_counter"], rel=1e-13)
    assert got["restarts"] == want["restarts"] == 0
    k = min(HIST_ROWS, got["iterations"], want["iterations"])
    np.testing.assert_allclose(got["history"][:k], want["history"][:k], rtol=1e-9)
    assert got["history"].shape == (got["iterations"], 3)
    assert np.linalg.norm(got["x"] - want["x"]) <= 1e-9 * np.linalg.norm(want["x"])
    true_err = np.linalg.norm(b - A @ got["x"])
    assert got["error"] == pytest.approx(true_err, rel=1e-3, abs=1e-13 * np.linalg.norm(b))
    assert got["error"] == got["error_best"], "same kernels on the same vector: bit-equal"
    # launches: r0, one per loop body, the explicit checks, the final residual; the host sees the break flag with a lag of
    # up to three polling windows, and the bodies enqueued meanwhile are predicated off on the device but still counted
    base = 1 + got["iterations"] + (got["iterations"] - 1) // 100 + 1
    assert base <= got["spmv_calls"] <= base + 3 * 32 + 1


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("name", ["laplace2d_40", "nonsym_dd_1500"])
def test_gpu_bicgstab_matches_oracle(oracle, name, fmt):
    A = SYSTEMS[name]()
    b = rhs(A)
    iters = 230
    want = oracle.pbicgstab(A.indptr, A.indices, A.data, b, iters)
    M = _matrix(A, fmt)
    got = M.pbicgstab(A.indptr, A.indices, A.data, b, iters)
    assert got["iterations"] == want["iterations"] == iters
    # BiCGSTAB's residual is not monotone (it grows 41 -> 55 in the first steps of the Laplacian) and amplifies the
    # last-bit differences of the dot products faster than CG: 1e-7 over the first rows instead of 1e-9
    np.testing.assert_allclose(got["history"][:HIST_ROWS], want["history"][:HIST_ROWS], rtol=1e-7)
    assert np.linalg.norm(got["x"] - want["x"]) <= 1e-9 * np.linalg.norm(want["x"])
    assert got["error_best"] <= 1e-10 * np.linalg.norm(b)
    assert got["error"] == got["error_best"]
    assert got["spmv_calls"] == 1 + 2 * iters + (iters - 1) // 100 + 1


@pytest.mark.gpu
def test_gpu_pcg_restart_path(oracle):
    A, b = near_singular_neumann()
    want = oracle.pcg(A.indptr, A.indices, A.data, b, 1000)
    M = _matrix(A, "csr_scalar")
    got = M.pcg(A.indptr, A.indices, A.data, b, 1000)
    # chaotic by construction (that is what makes the restart fire): only the rule itself is comparable
    assert want["restarts"] >= 1 and got["restarts"] >= 1
    h = got["history"]
    assert got["iterations"] == 1000 and h.shape == (1000, 3)
    assert np.all(np.diff(h[:, 2]) <= 0)
    assert got["error_best"] <= h[:, 1].min()
    assert got["error_best"] < np.linalg.norm(b)
    # explicit residual is re-evaluated only at multiples of 100
    for k in range(1, 1000):
        if k % 100:
            assert h[k, 1] == h[k - 1, 1]
    assert np.linalg.norm(b - A @ got["x"]) == pytest.approx(got["error"], rel=1e-2)


@pytest.mark.gpu
def test_gpu_pcg_fp32(oracle):
    A = SYSTEMS["random_spd_2000"]()
    b = rhs(A).astype(np.float32)
    want = oracle.pcg(A.indptr, A.indices, A.data, b.astype(np.float64), 60)
    M = _matrix(A, "csr_vector", np.float32)
    got = M.pcg(A.indptr, A.indices, A.data, b, 60)
    assert got["x"].dtype == np.float32
    # the RECURRENCE residual keeps shrinking below fp32 resolution, so the 1e-15*|b| break is still reached
    assert abs(got["iterations"] - want["iterations"]) <= 4
    assert np.linalg.norm(got["x"] - want["x"]) <= 2e-4 * np.linalg.norm(want["x"])
    np.testing.assert_allclose(got["history"][:5, 0], want["history"][:5, 0], rtol=1e-4)


@pytest.mark.gpu
def test_gpu_solver_breaks_at_iteration_zero_and_respects_max_iterations(oracle):
    A = SYSTEMS["laplace2d_40"]()
    b = rhs(A)
    M = _matrix(A, "csr_vector")
    for iters in (0, 1, 7, 33, 100, 101):
        want = oracle.pcg(A.indptr, A.indices, A.data, b, iters)
        got = M.pcg(A.indptr, A.indices, A.data, b, iters)
        assert got["iterations"] == want["iterations"] == iters
        assert np.linalg.norm(got["x"] - want["x"]) <= 1e-9 * max(np.linalg.norm(want["x"]), 1e-300)
        assert got["error_best"] == pytest.approx(want["err_best"], rel=1e-7)


@pytest.mark.gpu
def test_gpu_solver_error_paths():
    import spmv_mi355x as eng
    A = laplace2d(5).tolil()
    A[3, 3] = 0
    A = A.tocsr()
    A.eliminate_zeros()
    M = _matrix(A, "csr_vector")
    with pytest.raises(eng.SpmvError, match="zero in diagonal"):
        M.pcg(A.indptr, A.indices, A.data, np.ones(25), 10)
    with pytest.raises(eng.SpmvError, match="zero in diagonal"):
        M.pbicgstab(A.indptr, A.indices, A.data, np.ones(25), 10)
    B = sp.random(6, 9, 0.5, random_state=0, format="csr")
    MB = _matrix(B, "csr_vector")
    with pytest.raises(eng.SpmvError, match="square"):
        MB._solve(eng.lib().spmv_mi355x_pcg, np.zeros(7, np.int32), B.indices, B.data, np.ones(6), 10, False)


@pytest.mark.gpu
def test_gpu_driver_solver_mode(oracle, tmp_path):
    """spmv_mi355x_bench --cg / --bicgstab: the reference's solver drivers (bench_cg.cpp bench()+compute()): b from
    <matrix>_b.mtx, one CSV row per entry of CG_MAX_NUM_ITERS."""
    import os
    import subprocess
    import scipy.io
    from conftest import ROOT
    A = laplace2d(24)
    b = rhs(A)
    scipy.io.mmwrite(str(tmp_path / "sys.mtx"), sp.coo_matrix(A), symmetry="general")
    scipy.io.mmwrite(str(tmp_path / "sys_b.mtx"), sp.coo_matrix(b.reshape(-1, 1)))
    exe = os.path.join(ROOT, "spmv-research_amd", "bin", "spmv_mi355x_bench")
    for flag, fn in (("--cg", oracle.pcg), ("--bicgstab", oracle.pbicgstab)):
        env = dict(os.environ, CG_MAX_NUM_ITERS="5 150", SPMV_MI355X_FORMAT="sell_c_sigma")
        r = subprocess.run([exe, flag, str(tmp_path / "sys.mtx")], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        rows = [l.split(",") for l in r.stderr.strip().splitlines() if l.count(",") >= 16]
        assert len(rows) == 2
        for row, iters in zip(rows, (5, 150)):
            want = fn(A.indptr, A.indices, A.data, b, iters)
            assert abs(int(row[7]) - want["iterations"]) <= 2
            true = np.linalg.norm(b - A @ want["x"])
            assert float(row[6]) == pytest.approx(true, rel=1e-3, abs=1e-12 * np.linalg.norm(b))
            assert row[11].startswith("MI355X_SELL") and int(row[2]) == A.shape[0] and int(row[4]) == A.nnz
        assert "read vector file time" in r.stdout
