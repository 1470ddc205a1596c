"""Row-block partition of y = A*x over the GPUs of one node — host logic shared by bench.py and the
world_size-2 gloo tests.

Partition: the reference's nnz-balanced contiguous row ranges (loop_partitioner_balance_prefix_sums,
lib/parallel_util.h:156-184, as csr.cpp:140 uses it per thread) applied with W = number of GPUs. Rank p owns rows
[offsets[p], offsets[p+1]) of A, the same slice of y and (square matrices) the same slice of x.

x layout: `world` slices padded to one common length, so that ONE equal-sized all_gather_into_tensor (RCCL allgather
over xGMI) fills it in place; column c owned by part p is renumbered p*padded + (c - offsets[p]) once at conversion
time (spmv_host.remap_columns), which costs nothing per SpMV.
"""
import numpy as np

import spmv_host as H


def row_partition(row_ptr, world):
    """offsets[world+1]: nnz-balanced contiguous row blocks."""
    m = len(row_ptr) - 1
    total = int(row_ptr[m]) - int(row_ptr[0])
    offsets = np.zeros(world + 1, np.int64)
    for p in range(world):
        s, e = H.partition_prefix_sums(world, p, row_ptr, m, total)
        offsets[p], offsets[p + 1] = s, e
    offsets[0], offsets[world] = 0, m
    assert np.all(np.diff(offsets) >= 0)
    return offsets


def padded_len(offsets, align=64):
    return int((int(np.diff(offsets).max()) + align - 1) // align * align) if len(offsets) > 1 else 0


def local_block(row_ptr, col_idx, values, offsets, rank):
    """Rows of `rank` as a local CSR (row_ptr from 0, GLOBAL column indices)."""
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    s, e = int(row_ptr[r0]), int(row_ptr[r1])
    return dict(m=r1 - r0, nnz=e - s, row_ptr=(row_ptr[r0:r1 + 1] - s).astype(np.int32),
                col_idx=np.ascontiguousarray(col_idx[s:e]).copy(), values=np.ascontiguousarray(values[s:e]).copy())


def to_padded_columns(col_idx, offsets, padded):
    """In place: global column -> position in the padded slice layout."""
    return H.remap_columns(col_idx, offsets, padded)


def padded_to_global(cols, offsets, padded):
    cols = np.asarray(cols, np.int64)
    p = cols // padded
    return offsets[p] + (cols - p * padded)


def scatter_x_padded(x_global, offsets, padded):
    """The padded x every rank holds after the allgather (host reference of the layout)."""
    world = len(offsets) - 1
    out = np.zeros(world * padded, x_global.dtype)
    for p in range(world):
        out[p * padded:p * padded + int(offsets[p + 1] - offsets[p])] = x_global[offsets[p]:offsets[p + 1]]
    return out
