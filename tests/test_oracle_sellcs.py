"""Row a7 of SURVEY §8: the BSC SELL-C-sigma library's layout (C = 256, sigma = 16384 in the reference, sell_c_s.cpp:58-60).
The oracle's restatement (orc_sellcs_layout) is pinned BIT FOR BIT to the reference's own format code — sellcs_format.c,
radix_sort.c and sellcs_utils.c compiled from where they lie into oracle/_ref/v3/libref_sellcs.so (the RISC-V kernels are not
built) — on the golden inputs and on random matrices: sigma-window order (stable, descending), slice widths, slice pointers,
column-major fill with (column 0, value 0) padding. The GPU tier (tests/test_gpu_parity.py) then checks the engine's layout
against this oracle."""
import numpy as np
import pytest

from conftest import CASES, load_case


def _ref():
    import refdrv
    if not refdrv.sellcs_available():
        pytest.skip("oracle/_ref/v3/libref_sellcs.so not built (needs /root/reference: build container only)")
    return refdrv


def _same(a, b, what):
    for k in ("row_order", "widths", "slice_ptr", "col", "val"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=f"{what}: {k}")


@pytest.mark.parametrize("case", CASES)
def test_oracle_layout_equals_reference_on_golden_inputs(oracle, case):
    refdrv = _ref()
    info, g = load_case(case)
    rp, ci, a = g["row_ptr"], g["col_idx"], g["values"]
    for C_rows, sigma in ((256, 16384), (64, 16384), (64, 64), (256, 512), (16, 48)):
        _same(oracle.sellcs_layout(rp, ci, a, C_rows, sigma), refdrv.ref_sellcs_layout(rp, ci, a, info["n"], C_rows, sigma),
              f"{case} C={C_rows} sigma={sigma}")


@pytest.mark.parametrize("seed", range(6))
def test_oracle_layout_equals_reference_on_random_matrices(oracle, seed):
    refdrv = _ref()
    rng = np.random.default_rng(100 + seed)
    m = int(rng.integers(1, 40000))
    n = int(rng.integers(1, 5000))
    lens = np.minimum(rng.poisson(rng.uniform(0.5, 30), m) * (rng.random(m) > 0.1), n)
    if seed == 0:
        lens[rng.integers(0, m, 3)] = n                   # a few full rows: many radix digits
    rp = np.zeros(m + 1, np.int32)
    np.cumsum(lens, out=rp[1:])
    ci = np.concatenate([np.sort(rng.choice(n, L, replace=False)) for L in lens if L] or [np.zeros(0, np.int64)]).astype(np.int32)
    a = rng.uniform(-1, 1, len(ci))
    for C_rows, sigma in ((256, 16384), (64, 1024), (32, 32)):
        got, want = oracle.sellcs_layout(rp, ci, a, C_rows, sigma), refdrv.ref_sellcs_layout(rp, ci, a, n, C_rows, sigma)
        _same(got, want, f"seed {seed} C={C_rows} sigma={sigma}")
        # properties of the layout itself: a permutation, descending inside every window, width = the slice's first row
        assert sorted(got["row_order"].tolist()) == list(range(m))
        sl = lens[got["row_order"]]
        for k in range(0, m, sigma):
            assert np.all(np.diff(sl[k:k + sigma]) <= 0)
        assert np.array_equal(got["widths"], sl[::C_rows])
