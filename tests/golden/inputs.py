"""Seeded synthetic inputs shared by the golden-vector generator and the tests.

Everything here is plain numpy with explicit seeds so that the generator
(`make_golden.py`, run once in the build container where /root/reference is
mounted) and the tests (run anywhere, including the GPU box where the reference
does not exist) see bit-identical inputs.  Nothing here imports the reference,
the oracle or the product.
"""
import numpy as np

CORA_N = 2708
CORA_NFEAT = 1433
CORA_NCLASS = 7
CORA_NHID = 16


def cora_features(seed=42, n=CORA_N, nfeat=CORA_NFEAT, p=0.0127):
    """Synthetic stand-in for cora.content (absent from the reference tree):
    Bernoulli(p) bag-of-words rows, row-normalized the way utils.normalize does
    (reference pygcn/utils.py:390-397; empty rows -> 0)."""
    rng = np.random.default_rng(seed)
    x = (rng.random((n, nfeat)) < p).astype(np.float32)
    rowsum = x.sum(1, dtype=np.float64)
    with np.errstate(divide="ignore"):
        rinv = np.where(rowsum > 0, 1.0 / rowsum, 0.0)
    return (x * rinv[:, None]).astype(np.float32)


def cora_labels(seed=42, n=CORA_N, nclass=CORA_NCLASS):
    rng = np.random.default_rng(seed + 1)
    return rng.integers(0, nclass, size=n).astype(np.int64)


def cora_splits():
    # reference pygcn/utils.py:370-372
    return np.arange(140), np.arange(200, 500), np.arange(500, 1500)


def random_coo(n_rows, n_cols, nnz, seed, empty_rows=(), hub_row=None, hub_deg=0,
               duplicates=0):
    """Random COO (unsorted, possibly with duplicates) with fp32 values in (0,1]."""
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, n_rows, size=nnz)
    cols = rng.integers(0, n_cols, size=nnz)
    if hub_row is not None:
        rows = np.concatenate([rows, np.full(hub_deg, hub_row)])
        cols = np.concatenate([cols, rng.integers(0, n_cols, size=hub_deg)])
    if duplicates:
        pick = rng.integers(0, len(rows), size=duplicates)
        rows = np.concatenate([rows, rows[pick]])
        cols = np.concatenate([cols, cols[pick]])
    if len(empty_rows):
        keep = ~np.isin(rows, np.asarray(empty_rows))
        rows, cols = rows[keep], cols[keep]
    vals = (1.0 - rng.random(len(rows))).astype(np.float32)
    perm = rng.permutation(len(rows))
    return rows[perm].astype(np.int64), cols[perm].astype(np.int64), vals[perm]


def dense(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)
