"""Input contract of the GraphConvolution path: the helpers of the reference's utils.py that
define what adjacency the layer receives, plus the synthetic graph generators the benchmark
configurations need.

Reference counterparts (pygcn/utils.py): `encode_onehot` :22-28, `normalize` :390-397,
`accuracy` :400-404, `sparse_mx_to_torch_sparse_tensor` :407-414, and the Cora recipe that the
fork keeps as a comment at :356-383 (upstream `load_data(path, dataset)`).
"""
import os

import numpy as np
import scipy.sparse as sp
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# the re-indexed pairs of the reference's data/cora/cora.cites (a data file; upstream ships its
# dataset inside the repository too) — the package's own copy, byte-identical to the golden fixture
# tests/golden/cora_graph.npz that tests/golden/make_golden.py writes
DEFAULT_CORA = os.path.join(_HERE, "data", "cora_graph.npz")


def encode_onehot(labels):
    classes = sorted(set(labels))   # sorted: deterministic (the reference iterates a set)
    classes_dict = {c: np.identity(len(classes))[i, :] for i, c in enumerate(classes)}
    return np.array(list(map(classes_dict.get, labels)), dtype=np.int32)


def normalize(mx):
    """Row-normalize sparse matrix: D^-1 · mx, rows that sum to 0 stay 0 (utils.py:390-397)."""
    rowsum = np.array(mx.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.
    r_mat_inv = sp.diags(r_inv)
    return r_mat_inv.dot(mx)


def accuracy(output, labels):
    preds = output.max(1)[1].type_as(labels)
    correct = preds.eq(labels).double()
    correct = correct.sum()
    return correct / len(labels)


def sparse_mx_to_torch_sparse_tensor(sparse_mx):
    """scipy sparse -> torch sparse COO, int64 indices [2,nnz], fp32 values (utils.py:407-414)."""
    sparse_mx = sparse_mx.tocoo().astype(np.float32)
    indices = torch.from_numpy(np.vstack((sparse_mx.row, sparse_mx.col)).astype(np.int64))
    values = torch.from_numpy(sparse_mx.data)
    return torch.sparse_coo_tensor(indices, values, torch.Size(sparse_mx.shape))


def _cora_edges(path, dataset):
    cites = os.path.join(path, f"{dataset}.cites") if os.path.isdir(path) else path
    if cites.endswith(".npz"):
        z = np.load(cites, allow_pickle=False)
        return z["edges"].astype(np.int32), int(z["n"]), None
    raw = np.genfromtxt(cites, dtype=np.int32)
    content = cites[:-len(".cites")] + ".content"
    if os.path.exists(content):
        ifl = np.genfromtxt(content, dtype=np.dtype(str))
        ids = np.array(ifl[:, 0], dtype=np.int32)
    else:
        ifl, ids = None, np.unique(raw)
    idx_map = {j: i for i, j in enumerate(ids)}
    edges = np.array(list(map(idx_map.get, raw.flatten())), dtype=np.int32).reshape(raw.shape)
    return edges, len(ids), ifl


def synthetic_node_data(n, nfeat=1433, nclass=7, p=0.0127, seed=42):
    """Seeded stand-in for cora.content (absent from the reference tree): Bernoulli(p)
    bag-of-words features, row-normalized, and uniform random labels."""
    rng = np.random.default_rng(seed)
    x = (rng.random((n, nfeat)) < p).astype(np.float32)
    features = np.asarray(normalize(sp.csr_matrix(x)).todense(), dtype=np.float32)
    labels = np.random.default_rng(seed + 1).integers(0, nclass, size=n).astype(np.int64)
    return features, labels


def load_data(path=DEFAULT_CORA, dataset="cora"):
    """Upstream `load_data` (recipe at utils.py:356-383): citation graph -> symmetric adjacency
    -> normalize(A + I) -> torch sparse COO; features row-normalized; fixed index splits.
    `path` is a directory holding `<dataset>.cites` (and optionally `.content`), a `.cites` file,
    or the committed edge-list fixture.  Without `.content`, features/labels are synthetic."""
    print('Loading {} dataset...'.format(dataset))
    edges, n, ifl = _cora_edges(path, dataset)
    if ifl is not None:
        features = normalize(sp.csr_matrix(ifl[:, 1:-1], dtype=np.float32))
        features = np.array(features.todense(), dtype=np.float32)
        labels = np.where(encode_onehot(ifl[:, -1]))[1]
    else:
        features, labels = synthetic_node_data(n)
    adj = sp.coo_matrix((np.ones(edges.shape[0]), (edges[:, 0], edges[:, 1])), shape=(n, n),
                        dtype=np.float32)
    adj = adj + adj.T.multiply(adj.T > adj) - adj.multiply(adj.T > adj)   # symmetrize
    adj = normalize(adj + sp.eye(adj.shape[0]))
    idx_train, idx_val, idx_test = range(140), range(200, 500), range(500, 1500)
    return (sparse_mx_to_torch_sparse_tensor(adj), torch.from_numpy(features),
            torch.from_numpy(np.asarray(labels, dtype=np.int64)), torch.LongTensor(idx_train),
            torch.LongTensor(idx_val), torch.LongTensor(idx_test))


# ------------------------------------------------------------------ synthetic graphs (C3-C5)
def rmat_edge_chunks(n, n_edges, seed=42, abcd=(0.57, 0.19, 0.19, 0.05), device="cpu"):
    """Generator over the stream of `n_edges` directed R-MAT pairs (ids >= n rejected) in chunks
    of at most 2^27 candidates: yields (src, dst) int64 tensors.  The stream depends on
    (n, n_edges, seed, device type) only, so every process that iterates it sees the same edges —
    a rank of the sharded path keeps the pairs it owns from each chunk and drops the rest, holding
    O(chunk) + O(nnz / ranks) memory instead of the whole edge list."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    scale = max(1, int(np.ceil(np.log2(max(n, 2)))))
    a, b, c, _ = abcd
    have = 0
    while have < n_edges:
        m = int(min(max((n_edges - have) * 1.3 + 1024, 1024), 2 ** 27))
        src = torch.zeros(m, dtype=torch.int64, device=device)
        dst = torch.zeros(m, dtype=torch.int64, device=device)
        for _ in range(scale):
            u = torch.rand(m, generator=gen, device=device)
            sbit = (u >= a + b)
            dbit = ((u >= a) & (u < a + b)) | (u >= a + b + c)
            src = (src << 1) | sbit
            dst = (dst << 1) | dbit
        ok = (src < n) & (dst < n)
        src, dst = src[ok], dst[ok]
        take = min(int(src.numel()), n_edges - have)
        have += take
        yield src[:take], dst[:take]


def rmat_edges(n, n_edges, seed=42, abcd=(0.57, 0.19, 0.19, 0.05), device="cpu"):
    """`n_edges` directed R-MAT pairs over `n` vertices (ids >= n rejected), as int64 tensors."""
    parts = list(rmat_edge_chunks(n, n_edges, seed, abcd, device))
    return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])


def vertex_permutation(n, perm_seed, device):
    """The seeded vertex relabelling of the synthetic graphs (breaks id/degree correlation)."""
    gen = torch.Generator(device=torch.device(device))
    gen.manual_seed(perm_seed)
    return torch.randperm(n, generator=gen, device=device)


def rmat_block_keys(n, n_edges, r0, r1, seed=42, perm_seed=43, device="cpu"):
    """Stored entries of rows [r0, r1) of the normalized adjacency `rmat_graph(n, n_edges, seed,
    perm_seed)` as sorted unique GLOBAL keys row·n + col (self-loops included): the edge stream is
    replayed chunk by chunk and only pairs whose (relabelled) source lies in the block are kept —
    O(chunk) + O(block) memory."""
    device = torch.device(device)
    perm = vertex_permutation(n, perm_seed, device) if perm_seed is not None else None
    keys = []
    for src, dst in rmat_edge_chunks(n, n_edges, seed, device=device):
        if perm is not None:
            src, dst = perm[src], perm[dst]
        keep = (src >= r0) & (src < r1)
        keys.append(torch.unique(src[keep] * n + dst[keep]))          # (chunk-local dedupe)
    del perm
    diag = torch.arange(r0, r1, device=device, dtype=torch.int64)
    return torch.unique(torch.cat(keys + [diag * n + diag]))


def csr_from_keys(key, n, r0, r1):
    """Sorted unique global keys row·n + col of rows [r0, r1) -> (rowptr rebased to 0, col global
    ids int32, val fp32 = 1 / stored entries of the row): D^-1(A + I) for a 0/1 adjacency."""
    row = key // n - r0
    deg = torch.bincount(row, minlength=r1 - r0)
    col = (key - (row + r0) * n).to(torch.int32)
    rowptr = torch.zeros(r1 - r0 + 1, dtype=torch.int64, device=key.device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    val = (1.0 / deg.to(torch.float32))[row]
    if col.numel() < 2 ** 31 - 1:
        rowptr = rowptr.to(torch.int32)
    return rowptr, col, val


def rmat_row_block(n, n_edges, r0, r1, seed=42, perm_seed=43, device="cpu", counts_only=False):
    """Rows [r0, r1) of the SAME normalized adjacency `rmat_graph(n, n_edges, seed, perm_seed)`
    builds, generated without ever holding the other rows (rmat_block_keys).  Returns
    (rowptr rebased to 0, col global ids int32, val fp32); with `counts_only` just the stored
    entries per row (int64 [r1 - r0]) — all a partitioner needs."""
    key = rmat_block_keys(n, n_edges, r0, r1, seed, perm_seed, device)
    if counts_only:
        return torch.bincount(key // n - r0, minlength=r1 - r0)
    return csr_from_keys(key, n, r0, r1)


def normalized_adjacency_csr(src, dst, n, perm_seed=43):
    """Directed pairs -> seeded vertex permutation -> dedupe -> + I -> D^-1(A+I) as CSR arrays
    (rowptr int32/int64, col int32, val fp32) on the tensors' device.  Same semantics as
    `normalize(adj + I)` (utils.py:368,390-397) for a 0/1 adjacency."""
    device = src.device
    if perm_seed is not None:
        gen = torch.Generator(device=device)
        gen.manual_seed(perm_seed)
        perm = torch.randperm(n, generator=gen, device=device)
        src, dst = perm[src], perm[dst]
    diag = torch.arange(n, device=device, dtype=torch.int64)
    key = torch.cat([src * n + dst, diag * n + diag])
    del src, dst
    key = torch.unique(key)                      # sorted: row-major CSR order, duplicates gone
    row = key // n
    col = (key - row * n).to(torch.int32)
    del key
    deg = torch.bincount(row, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    val = (1.0 / deg.to(torch.float32))[row]
    if col.numel() < 2 ** 31 - 1:
        rowptr = rowptr.to(torch.int32)
    return rowptr, col, val


def rmat_graph(n, n_edges, seed=42, perm_seed=43, device="cpu"):
    """Config C3/C4 generator: R-MAT(0.57,0.19,0.19,0.05), permuted, deduped,
    self-loops, row-normalized."""
    src, dst = rmat_edges(n, n_edges, seed=seed, device=device)
    return normalized_adjacency_csr(src, dst, n, perm_seed=perm_seed)


def uniform_graph(n, degree=10, seed=46, device="cpu"):
    """The CACHE-HOSTILE counterpart of the R-MAT graphs: every vertex draws
    `degree` neighbours uniformly at random (seeded), duplicates removed, + I, row-normalized
    (`normalize(adj + I)`, utils.py:368,390-397) — no hubs, so no dense-operand row is re-read often
    enough for the 256 MiB Infinity Cache to matter: at 10^7 vertices x 1 KiB rows practically every
    gather is an HBM miss.  Returns CSR arrays like rmat_graph."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    row = torch.arange(n, device=device, dtype=torch.int64)
    keys = [row * n + row]
    for _ in range(degree):                       # (one column of neighbours at a time: O(n) temporaries)
        keys.append(row * n + torch.randint(0, n, (n,), generator=gen, device=device, dtype=torch.int64))
    key = torch.unique(torch.cat(keys))
    del keys, row
    return csr_from_keys(key, n, 0, n)
