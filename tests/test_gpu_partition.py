"""Graph partition on the GPU: every rank's kernels under both layouts against the oracle on the WHOLE matrix.
One process plays the ranks in turn (an 8-GPU node is the driver's to run): x holds only what the rank owns or receives —
NaN everywhere else, so a column the exchange lists missed poisons y — and the rank's y must be the oracle's rows.
 * layout "original": full-length x in original numbering, rows split by the OWNER of the column (local + remote handle)
 * layout "padded"  : P A P^T with padded x slices and the column-range split (col_filter_mode 1 / 2)
Tolerance 1e-12 / 1e-5 of sum|a||x| (the split and, for "padded", the renumbering reorder a row's sum)."""
import numpy as np
import pytest

import spmv_dist as D
import spmv_host as H

pytestmark = pytest.mark.gpu

FORMATS = ["sell_c_sigma", "csr_vector", "csr_stream", "csr_merge", "coo"]


def _matrix(name):
    return H.gen_kkt(11) if name == "kkt" else H.gen_named("scircuit", 0.05)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("name,world", [("kkt", 3), ("kkt", 8), ("scircuit", 4)])
def test_original_layout_rank_kernels_match_oracle(oracle, name, world, fmt, dtype):
    import spmv_mi355x as eng
    A = _matrix(name)
    m = A["m"]
    part = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, world, "graph")
    owner = part.owner()
    x = np.random.default_rng(5).uniform(-1, 1, m).astype(dtype)
    vals = A["values"].astype(dtype)
    y_ref = oracle.csr_spmv(A["row_ptr"], A["col_idx"], vals, x)
    absrow = oracle.csr_spmv(A["row_ptr"], A["col_idx"], np.abs(vals).astype(np.float64), np.abs(x).astype(np.float64))
    tol = 1e-12 if dtype == np.float64 else 1e-5
    for r in range(world):
        blk, rows = D.original_block(A["row_ptr"], A["col_idx"], A["values"], owner, r)
        if blk["m"] == 0:
            continue
        _send, recv = H.halo_lists(A["row_ptr"], A["col_idx"], owner, world, r)
        loc, rem = D.split_by_owner(blk, owner, r)
        xr = np.full(m, np.nan, dtype)
        xr[rows] = x[rows]
        y = None
        Ml = eng.Matrix(loc["row_ptr"], loc["col_idx"], loc["values"], blk["m"], m, fmt, dtype)
        y_loc = Ml.spmv(np.nan_to_num(xr))                       # before the halo arrives: owned columns only
        Ml.close()
        for q in range(world):
            xr[recv[q]] = x[recv[q]]
        Mr = eng.Matrix(rem["row_ptr"], rem["col_idx"], rem["values"], blk["m"], m, fmt, dtype)
        xs = xr.copy()
        untouched = np.isnan(xs)
        xs[untouched] = 0                                         # the kernels may prefetch, never use, other entries
        y = y_loc.astype(np.float64) + Mr.spmv(xs).astype(np.float64)
        Mr.close()
        assert not np.isnan(xr[blk["col_idx"]]).any()
        err = np.abs(y - y_ref[rows].astype(np.float64)) / np.maximum(absrow[rows], 1e-300)
        assert err.max() <= tol, (r, err.max())


@pytest.mark.parametrize("fmt", ["sell_c_sigma", "csr_vector"])
@pytest.mark.parametrize("world", [2, 5])
def test_padded_layout_rank_kernels_match_oracle(oracle, world, fmt):
    import spmv_mi355x as eng
    A = H.gen_kkt(11)
    m = A["m"]
    part = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, world, "graph")
    x_old = np.random.default_rng(6).uniform(-1, 1, m)
    y_ref = oracle.csr_spmv(A["row_ptr"], A["col_idx"], A["values"], x_old)[part.perm]
    absrow = oracle.csr_spmv(A["row_ptr"], A["col_idx"], np.abs(A["values"]), np.abs(x_old))[part.perm]
    off = part.offsets
    padded = D.padded_len(off)
    xp = D.scatter_x_padded(x_old[part.perm], off, padded)
    for r in range(world):
        blk = D.partition_block(A["row_ptr"], A["col_idx"], A["values"], part, r)
        D.to_padded_columns(blk["col_idx"], off, padded)
        r0, r1 = int(off[r]), int(off[r + 1])
        c0, c1 = r * padded, r * padded + (r1 - r0)
        Ml = eng.Matrix(blk["row_ptr"], blk["col_idx"], blk["values"], blk["m"], world * padded, fmt, np.float64,
                        col_begin=c0, col_end=c1, col_filter_mode=1)
        Mr = eng.Matrix(blk["row_ptr"], blk["col_idx"], blk["values"], blk["m"], world * padded, fmt, np.float64,
                        col_begin=c0, col_end=c1, col_filter_mode=2)
        # what the sub-range exchange delivers, nothing else
        xr = np.zeros(world * padded)
        xr[c0:c1] = xp[c0:c1]
        rg = D.needed_subranges(blk["col_idx"], padded, world, min_gap=8)
        for q in range(world):
            for k in range(rg.shape[1]):
                a, b = q * padded + int(rg[q, k, 0]), q * padded + int(rg[q, k, 1])
                xr[a:b] = xp[a:b]
        y = Ml.spmv(xr) + Mr.spmv(xr)
        Ml.close(); Mr.close()
        err = np.abs(y - y_ref[r0:r1]) / np.maximum(absrow[r0:r1], 1e-300)
        assert err.max() <= 1e-12, (r, err.max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("name,world", [("kkt", 2), ("kkt", 8), ("scircuit", 4)])
def test_interior_and_boundary_row_handles_match_the_undivided_matrix(oracle, name, world, fmt, dtype):
    """bench.py's original-numbering layout: an interior-row handle (runs while the halo is in flight: x holds NaN for
    everything not owned) and a boundary-row handle writing behind it in the same y. Rows are whole and in the matrix's own
    entry order, so y must equal the SAME kernel's result on the undivided matrix to the format's tolerance and the oracle's
    to 1e-12 / 1e-5."""
    import spmv_mi355x as eng
    A = _matrix(name)
    m = A["m"]
    part = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, world, "graph")
    owner = part.owner()
    x = np.random.default_rng(5).uniform(-1, 1, m).astype(dtype)
    vals = A["values"].astype(dtype)
    y_ref = oracle.csr_spmv(A["row_ptr"], A["col_idx"], vals, x)
    absrow = oracle.csr_spmv(A["row_ptr"], A["col_idx"], np.abs(vals).astype(np.float64), np.abs(x).astype(np.float64))
    tol = 1e-12 if dtype == np.float64 else 1e-5
    for r in range(world):
        B = D.interior_boundary_blocks(A["row_ptr"], A["col_idx"], A["values"], owner, r)
        rows, t = B["rows"], B["split"]
        _send, recv = H.halo_lists(A["row_ptr"], A["col_idx"], owner, world, r)
        xr = np.full(m, np.nan, dtype)
        xr[rows] = x[rows]
        y = np.zeros(B["m"], np.float64)
        if t > 0:
            Mi = eng.Matrix(B["interior"]["row_ptr"], B["interior"]["col_idx"], B["interior"]["values"], t, m, fmt, dtype)
            y[:t] = Mi.spmv(np.where(np.isnan(xr), 0, xr).astype(dtype))
            Mi.close()
            assert np.all(owner[B["interior"]["col_idx"]] == r)
        for q in range(world):
            xr[recv[q]] = x[recv[q]]
        assert not np.isnan(xr[B["col_idx"]]).any()
        if B["m"] - t > 0:
            Mb = eng.Matrix(B["boundary"]["row_ptr"], B["boundary"]["col_idx"], B["boundary"]["values"], B["m"] - t, m, fmt, dtype)
            y[t:] = Mb.spmv(np.where(np.isnan(xr), 0, xr).astype(dtype))
            Mb.close()
        err = np.abs(y - y_ref[rows].astype(np.float64)) / np.maximum(absrow[rows], 1e-300)
        assert err.max() <= tol, (r, err.max())
