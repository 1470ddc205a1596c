"""Host logic of the communication-aware row partition (host/graph_partition.cpp, spmv_dist.graph_partition): the engine
runs on P A P^T, so P must be a permutation, the blocks must be exactly the permuted rows, and the layout must keep what
peers read in a prefix of every slice. scipy is the independent check of the permutation algebra."""
import numpy as np
import pytest
import scipy.sparse as sp

import spmv_dist as D
import spmv_host as H


def _csr(A):
    return sp.csr_matrix((A["values"], A["col_idx"], A["row_ptr"]), shape=(A["m"], A["n"]))


def _cases():
    yield "kkt9", H.gen_kkt(9)
    yield "cant_twin", H.gen_named("cant", 0.02)
    rng = np.random.default_rng(3)                       # two components + isolated vertices + an empty row
    m = 300
    rows = rng.integers(0, 140, 900); cols = rng.integers(0, 140, 900)
    r2 = rng.integers(150, 290, 700); c2 = rng.integers(150, 290, 700)
    R = np.concatenate([rows, cols, r2, c2]); Cc = np.concatenate([cols, rows, c2, r2])
    S = sp.coo_matrix((rng.uniform(-1, 1, len(R)), (R, Cc)), shape=(m, m)).tocsr()
    S.sum_duplicates(); S.sort_indices()
    yield "two_components", dict(m=m, n=m, nnz=S.nnz, row_ptr=S.indptr.astype(np.int32), col_idx=S.indices.astype(np.int32),
                                 values=S.data.astype(np.float64))


@pytest.mark.parametrize("name,A", list(_cases()), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("world", [1, 2, 5])
def test_graph_partition_is_a_permutation_of_the_same_operator(name, A, world):
    m = A["m"]
    order = H.bfs_order(A["row_ptr"], A["col_idx"], m, m)
    assert sorted(order.tolist()) == list(range(m))
    part = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, world, "graph" if world > 1 else "auto")
    if world == 1:
        assert part.kind == "rows" and part.perm is None
        return
    perm, inv, off = part.perm, part.inv, part.offsets
    assert sorted(perm.tolist()) == list(range(m)) and off[0] == 0 and off[-1] == m and np.all(np.diff(off) >= 0)
    np.testing.assert_array_equal(perm[inv], np.arange(m))
    S = _csr(A)
    P = S[perm][:, perm].tocsr()
    P.sort_indices()
    x = np.random.default_rng(1).uniform(-1, 1, m)
    y_old = S @ x
    owner_new = np.repeat(np.arange(world), np.diff(off))
    for r in range(world):
        blk = D.partition_block(A["row_ptr"], A["col_idx"], A["values"], part, r)
        r0, r1 = int(off[r]), int(off[r + 1])
        assert blk["m"] == r1 - r0
        np.testing.assert_array_equal(blk["row_ptr"], P.indptr[r0:r1 + 1] - P.indptr[r0])
        np.testing.assert_array_equal(blk["col_idx"], P.indices[P.indptr[r0]:P.indptr[r1]])
        np.testing.assert_array_equal(blk["values"], P.data[P.indptr[r0]:P.indptr[r1]])
        B = sp.csr_matrix((blk["values"], blk["col_idx"], blk["row_ptr"]), shape=(blk["m"], m))
        np.testing.assert_allclose(B @ x[perm], y_old[perm][r0:r1], rtol=0, atol=1e-12)
        # volume = distinct remote columns of the block
        remote = np.unique(blk["col_idx"][owner_new[blk["col_idx"]] != r])
        assert part.volume[r] == len(remote)
    # layout: inside every slice, everything a peer reads sits before everything nobody else reads
    read_by_peer = np.zeros(m, bool)
    rows_new = np.repeat(np.arange(m), np.diff(P.indptr))
    cross = owner_new[rows_new] != owner_new[P.indices]
    read_by_peer[P.indices[cross]] = True
    for r in range(world):
        seg = read_by_peer[off[r]:off[r + 1]]
        k = int(seg.sum())
        assert seg[:k].all() and not seg[k:].any()
        # and each group keeps the matrix's original order
        assert np.all(np.diff(perm[off[r] + k:off[r + 1]]) > 0)


def test_auto_picks_the_partition_with_less_exchange():
    A = H.gen_kkt(12)
    m = A["m"]
    rows = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, 4, "rows")
    graph = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, 4, "graph")
    auto = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, 4, "auto")
    assert graph.volume.max() < rows.volume.max()               # KKT coupling: constraint rows read all of H's x
    assert auto.kind == "graph"
    np.testing.assert_array_equal(auto.perm, graph.perm)
    # a banded matrix is already in a good order: nothing to gain, stay with the reference's row blocks
    B = H.gen_named("cant", 0.05)
    auto = D.graph_partition(B["row_ptr"], B["col_idx"], B["m"], B["m"], 4, "auto")
    rows = D.graph_partition(B["row_ptr"], B["col_idx"], B["m"], B["m"], 4, "rows")
    assert auto.volume.max() <= rows.volume.max()


def test_non_square_falls_back_to_row_blocks():
    rp = np.array([0, 2, 3, 5], np.int32)
    ci = np.array([0, 4, 1, 2, 3], np.int32)
    part = D.graph_partition(rp, ci, 3, 5, 2, "auto")
    assert part.kind == "rows" and part.perm is None
    with pytest.raises(H.HostError, match="square"):
        H.bfs_order(rp, ci, 3, 5)


def test_balance_of_the_graph_partition():
    A = H.gen_kkt(14)
    m, nnz = A["m"], A["nnz"]
    part = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, 8, "graph")
    per = np.array([D.partition_block(A["row_ptr"], A["col_idx"], A["values"], part, r)["nnz"] for r in range(8)])
    assert per.sum() == nnz and per.max() <= 1.1 * nnz / 8


@pytest.mark.parametrize("world", [2, 4])
def test_halo_lists_and_original_numbering_blocks(world):
    """Original-numbering layout: send list of p towards q == receive list of q from p (both ascending, so packed buffers
    need no header); a rank's rows + its own x entries + what it receives reproduce its rows of y = A x exactly as
    local + remote; the received count is the partition's volume."""
    A = H.gen_kkt(12)
    m = A["m"]
    part = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, world, "graph")
    owner = part.owner()
    np.testing.assert_array_equal(np.bincount(owner, minlength=world), np.diff(part.offsets))
    S = _csr(A)
    x = np.random.default_rng(0).uniform(-1, 1, m)
    y = S @ x
    lists = [H.halo_lists(A["row_ptr"], A["col_idx"], owner, world, r) for r in range(world)]
    for p in range(world):
        send, recv = lists[p]
        assert len(send[p]) == 0 and len(recv[p]) == 0
        for q in range(world):
            np.testing.assert_array_equal(send[q], lists[q][1][p])
            assert np.all(np.diff(send[q]) > 0) and np.all(owner[send[q]] == p) and np.all(owner[recv[q]] == q)
        assert part.volume[p] == sum(len(l) for l in recv)
        blk, rows = D.original_block(A["row_ptr"], A["col_idx"], A["values"], owner, p)
        np.testing.assert_array_equal(rows, np.flatnonzero(owner == p))
        loc, rem = D.split_by_owner(blk, owner, p)
        assert loc["nnz"] + rem["nnz"] == blk["nnz"] and np.all(owner[loc["col_idx"]] == p) and np.all(owner[rem["col_idx"]] != p)
        have = np.full(m, np.nan)
        have[rows] = x[rows]
        for q in range(world):
            have[recv[q]] = x[recv[q]]
        assert not np.isnan(have[blk["col_idx"]]).any()              # everything the rows read is owned or received
        have = np.nan_to_num(have)
        mk = lambda b: sp.csr_matrix((b["values"], b["col_idx"], b["row_ptr"]), shape=(b["m"], m))
        np.testing.assert_allclose(mk(loc) @ have + mk(rem) @ have, y[rows], rtol=0, atol=1e-12)
        np.testing.assert_array_equal((mk(blk) @ have), (S[rows] @ x))


@pytest.mark.parametrize("name,world", [("kkt14", 2), ("kkt14", 4), ("cant_twin", 3)])
def test_interior_rows_first_then_boundary_rows(name, world):
    """What bench.py builds for the original-numbering layout: interior rows (every column owned) first, boundary rows after
    them, both ascending, every row WHOLE and in the matrix's own entry order — so each y entry is bit-identical to the same
    row of the undivided matrix summed the same way."""
    A = H.gen_kkt(14) if name == "kkt14" else H.gen_named("cant", 0.05)
    m = A["m"]
    part = D.graph_partition(A["row_ptr"], A["col_idx"], m, m, world, "graph")
    owner = part.owner()
    S = _csr(A)
    for r in range(world):
        B = D.interior_boundary_blocks(A["row_ptr"], A["col_idx"], A["values"], owner, r)
        rows, t = B["rows"], B["split"]
        np.testing.assert_array_equal(np.sort(rows), np.flatnonzero(owner == r))
        assert np.all(np.diff(rows[:t]) > 0) and np.all(np.diff(rows[t:]) > 0)
        I, Bd = B["interior"], B["boundary"]
        assert I["m"] == t and Bd["m"] == B["m"] - t and I["nnz"] + Bd["nnz"] == B["nnz"]
        assert np.all(owner[I["col_idx"]] == r)
        if Bd["m"]:
            remote = (owner[Bd["col_idx"]] != r).astype(np.int64)
            assert np.all(np.diff(Bd["row_ptr"]) > 0) and np.all(np.add.reduceat(remote, Bd["row_ptr"][:-1]) > 0)
        for blk, rr in ((I, rows[:t]), (Bd, rows[t:]), (B, rows)):
            want = S[rr]
            np.testing.assert_array_equal(blk["row_ptr"], want.indptr)
            np.testing.assert_array_equal(blk["col_idx"], want.indices)
            np.testing.assert_array_equal(blk["values"], want.data)


def test_diagonal_only_keeps_the_first_stored_diagonal_entry():
    # local rows are original rows 5, 2, 9; row 2 has its diagonal twice (first one counts), row 9 has none
    rows = np.array([5, 2, 9])
    blk = dict(row_ptr=np.array([0, 3, 6, 8], np.int32), col_idx=np.array([1, 5, 7, 2, 2, 4, 0, 3], np.int32),
               values=np.array([1., 50., 3., 20., 21., 5., 6., 7.]))
    d = D.diagonal_only(blk, rows)
    np.testing.assert_array_equal(d["row_ptr"], [0, 1, 2, 2])
    np.testing.assert_array_equal(d["col_idx"], [0, 1])
    np.testing.assert_array_equal(d["values"], [50., 20.])


@pytest.mark.parametrize("seed", range(12))
def test_random_graphs_fuzz(seed):
    """Random square patterns — unsymmetric, self loops, empty rows, duplicates, isolated vertices, more parts than busy
    vertices: every piece of the partition machinery must stay consistent with scipy's permutation algebra."""
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.integers(1, 400))
    nnz = int(rng.integers(0, 6 * m + 1))
    R = rng.integers(0, m, nnz)
    Cc = np.clip(R + rng.integers(-12, 13, nnz), 0, m - 1) if seed % 2 else rng.integers(0, m, nnz)
    order = np.lexsort((Cc, R))                                   # rows ascending, columns ascending, duplicates kept
    R, Cc = R[order], Cc[order]
    rp = np.zeros(m + 1, np.int32)
    np.add.at(rp, R + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    ci = Cc.astype(np.int32)
    va = rng.uniform(-1, 1, nnz)
    world = int(rng.integers(2, 9))
    S = sp.csr_matrix((va, ci, rp), shape=(m, m))                 # duplicates stay separate entries in .data order
    x = rng.uniform(-1, 1, m)
    y = np.array([np.dot(va[rp[i]:rp[i + 1]], x[ci[rp[i]:rp[i + 1]]]) for i in range(m)])
    bo = H.bfs_order(rp, ci, m, m)
    assert sorted(bo.tolist()) == list(range(m))
    part = D.graph_partition(rp, ci, m, m, world, "graph")
    owner = part.owner()
    assert sorted(part.perm.tolist()) == list(range(m)) and np.array_equal(np.bincount(owner, minlength=world), np.diff(part.offsets))
    vol = H.partition_volume(rp, ci, owner, world)
    np.testing.assert_array_equal(vol, part.volume)
    for r in range(world):
        send, recv = H.halo_lists(rp, ci, owner, world, r)
        assert sum(len(l) for l in recv) == vol[r]
        B = D.interior_boundary_blocks(rp, ci, va, owner, r)
        rows, t = B["rows"], B["split"]
        np.testing.assert_array_equal(np.sort(rows), np.flatnonzero(owner == r))
        have = np.full(m, np.nan)
        have[rows] = x[rows]
        if B["interior"]["nnz"]:
            assert not np.isnan(have[B["interior"]["col_idx"]]).any()
        for q in range(world):
            have[recv[q]] = x[recv[q]]
        got = np.array([np.dot(B["values"][B["row_ptr"][i]:B["row_ptr"][i + 1]], have[B["col_idx"][B["row_ptr"][i]:B["row_ptr"][i + 1]]])
                        for i in range(B["m"])])
        np.testing.assert_array_equal(got, y[rows])               # same entries in the same order: bit-identical
        # padded P A P^T layout
        blk = D.partition_block(rp, ci, va, part, r)
        r0, r1 = int(part.offsets[r]), int(part.offsets[r + 1])
        Bm = sp.csr_matrix((blk["values"], blk["col_idx"], blk["row_ptr"]), shape=(blk["m"], m))
        np.testing.assert_allclose(Bm @ x[part.perm], y[part.perm][r0:r1], rtol=0, atol=1e-12)
