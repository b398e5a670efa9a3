"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement (numpy + the plain-C kernels of spmm_oracle.c) of the GraphConvolution hot path
of LinChen-65/pygcn.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; pygcn_amd/ never does.

Reference lines each function follows (paths relative to /root/reference):
  * gc_forward / gc_backward ......... pygcn/layers.py:32-38 (mm -> spmm -> +bias) and the torch
                                        autograd formulas for mm / add it triggers
                                        (train.py:157 `loss.backward()`).
  * init_bounds ....................... pygcn/layers.py:23-29 (kaiming_uniform_ on [in,out] ->
                                        fan_in = size(1) = out; bias U(+-1/sqrt(out))).
  * gcn2_* ............................ upstream 2-layer model shape preserved in the comments at
                                        pygcn/models.py:23,48,50,68.
  * normalize / cora_adjacency ........ pygcn/utils.py:390-397 and the recipe at utils.py:356-368.
  * accuracy .......................... pygcn/utils.py:400-404.
  * adam_step / train_trajectory ...... pygcn/train.py:41-47,111-112 (Adam lr .01 wd 5e-4, seed 42)
                                        with the upstream epoch body named in the comments at
                                        train.py:140,150.
The sparse arithmetic itself lives in PyTorch (third-party; see spmm_oracle.c's header).

Parity pin: tests/test_oracle_golden.py checks every function here against the golden vectors
that tests/golden/make_golden.py captured by importing the reference in the build container.
"""
import ctypes
import os

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def lib():
    """Load oracle/liboracle_spmm.so (built by oracle/Makefile or __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_spmm.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle` or __graft_entry__.build()")
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# --------------------------------------------------------------------------- sparse products
def spmm_coo(row, col, val, B, n_rows):
    """out = A @ B for COO A in storage order (duplicates summed) — layers.py:34."""
    row, col, val, B = _c(row, np.int64), _c(col, np.int64), _c(val, np.float32), _c(B, np.float32)
    F = B.shape[1]
    C = np.empty((n_rows, F), np.float32)
    lib().oracle_spmm_coo_f32(ctypes.c_int64(len(val)), _p(row, _i64p), _p(col, _i64p),
                              _p(val, _f32p), _p(B, _f32p), ctypes.c_int64(F), _p(C, _f32p),
                              ctypes.c_int64(F), ctypes.c_int64(n_rows), ctypes.c_int64(F))
    return C


def spmm_csr(rowptr, col, val, B):
    """out = A @ B for CSR A, row-parallel, per-row sequential fp32 sum — layers.py:34."""
    rowptr, col, val, B = (_c(rowptr, np.int64), _c(col, np.int32), _c(val, np.float32),
                           _c(B, np.float32))
    n_rows, F = len(rowptr) - 1, B.shape[1]
    C = np.empty((n_rows, F), np.float32)
    lib().oracle_spmm_csr_f32(ctypes.c_int64(n_rows), _p(rowptr, _i64p), _p(col, _i32p),
                              _p(val, _f32p), _p(B, _f32p), ctypes.c_int64(F), _p(C, _f32p),
                              ctypes.c_int64(F), ctypes.c_int64(F))
    return C


def spmm_csr_f64acc(rowptr, col, val, B):
    rowptr, col, val, B = (_c(rowptr, np.int64), _c(col, np.int32), _c(val, np.float32),
                           _c(B, np.float32))
    n_rows, F = len(rowptr) - 1, B.shape[1]
    C = np.empty((n_rows, F), np.float64)
    lib().oracle_spmm_csr_f64acc(ctypes.c_int64(n_rows), _p(rowptr, _i64p), _p(col, _i32p),
                                 _p(val, _f32p), _p(B, _f32p), ctypes.c_int64(F),
                                 _p(C, _f64p), ctypes.c_int64(F), ctypes.c_int64(F))
    return C


def spmm_csr_t(rowptr, col, val, G, n_cols):
    """grad_support = A^T @ G (backward of the spmm; torch `mm` derivative for mat2)."""
    rowptr, col, val, G = (_c(rowptr, np.int64), _c(col, np.int32), _c(val, np.float32),
                           _c(G, np.float32))
    n_rows, F = len(rowptr) - 1, G.shape[1]
    out = np.empty((n_cols, F), np.float32)
    lib().oracle_spmm_csr_t_f32(ctypes.c_int64(n_rows), ctypes.c_int64(n_cols),
                                _p(rowptr, _i64p), _p(col, _i32p), _p(val, _f32p),
                                _p(G, _f32p), ctypes.c_int64(F), _p(out, _f32p),
                                ctypes.c_int64(F), ctypes.c_int64(F))
    return out


def csr_transpose(rowptr, col, val, n_cols):
    rowptr, col, val = _c(rowptr, np.int64), _c(col, np.int32), _c(val, np.float32)
    n_rows, nnz = len(rowptr) - 1, len(val)
    rp_t = np.empty(n_cols + 1, np.int64)
    col_t = np.empty(nnz, np.int32)
    val_t = np.empty(nnz, np.float32)
    lib().oracle_csr_transpose(ctypes.c_int64(n_rows), ctypes.c_int64(n_cols), _p(rowptr, _i64p),
                               _p(col, _i32p), _p(val, _f32p), _p(rp_t, _i64p),
                               _p(col_t, _i32p), _p(val_t, _f32p))
    return rp_t, col_t, val_t


def coo_to_csr(row, col, val, n_rows):
    """Stable row sort of a COO list; duplicates are kept as separate stored entries (their sum
    is what torch.spmm produces for an uncoalesced tensor)."""
    row = np.asarray(row, np.int64)
    order = np.argsort(row, kind="stable")
    rowptr = np.zeros(n_rows + 1, np.int64)
    np.add.at(rowptr, row + 1, 1)
    return np.cumsum(rowptr), np.asarray(col)[order].astype(np.int32), \
        np.asarray(val, np.float32)[order]


class CSR:
    """Plain CSR triple with shape; the oracle's adjacency type."""

    def __init__(self, rowptr, col, val, shape):
        self.rowptr, self.col, self.val = (_c(rowptr, np.int64), _c(col, np.int32),
                                            _c(val, np.float32))
        self.shape = tuple(int(s) for s in shape)

    @classmethod
    def from_coo(cls, row, col, val, shape):
        return cls(*coo_to_csr(row, col, val, shape[0]), shape)

    @classmethod
    def from_scipy(cls, m):
        m = sp.csr_matrix(m)
        m.sort_indices()
        return cls(m.indptr, m.indices, m.data.astype(np.float32), m.shape)

    @property
    def nnz(self):
        return len(self.val)

    def matmul(self, B):
        return spmm_csr(self.rowptr, self.col, self.val, B)

    def t_matmul(self, G):
        return spmm_csr_t(self.rowptr, self.col, self.val, G, self.shape[1])


# --------------------------------------------------------------------------- data recipe
def normalize(mx):
    """Row-normalize a scipy sparse matrix: D^-1 · M with inf -> 0 (utils.py:390-397)."""
    rowsum = np.array(mx.sum(1), dtype=np.float64).flatten()
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1)
    r_inv[np.isinf(r_inv)] = 0.0
    return sp.diags(r_inv).dot(mx)


def cora_adjacency(edges, n):
    """coo(ones) -> symmetrize -> normalize(A + I) (utils.py:360-368); float32 CSR out."""
    adj = sp.coo_matrix((np.ones(edges.shape[0]), (edges[:, 0], edges[:, 1])), shape=(n, n),
                        dtype=np.float32)
    adj = adj + adj.T.multiply(adj.T > adj) - adj.multiply(adj.T > adj)
    adj = normalize(adj + sp.eye(n))
    return CSR.from_scipy(adj.astype(np.float32))


def accuracy(output, labels):
    """utils.py:400-404."""
    return float((output.argmax(1) == labels).astype(np.float64).sum() / len(labels))


# --------------------------------------------------------------------------- the layer
def init_bounds(in_features, out_features):
    """(weight bound, bias bound) of layers.py:23-29: kaiming_uniform_(a=0) on a [in,out]
    tensor takes fan_in = size(1) = out_features -> sqrt(6/out); bias +-1/sqrt(out)."""
    return float(np.sqrt(6.0 / out_features)), float(1.0 / np.sqrt(out_features))


def gc_forward(x, weight, bias, adj):
    """layers.py:32-38.  Returns (output, support)."""
    support = (np.asarray(x, np.float32) @ np.asarray(weight, np.float32)).astype(np.float32)
    out = adj.matmul(support)
    if bias is not None:
        out = out + np.asarray(bias, np.float32)
    return out, support


def gc_backward(x, weight, has_bias, adj, grad_out, need_grad_x=True):
    """Autograd of layers.py:32-38: grad_bias = sum_rows(g); grad_support = A^T g;
    grad_W = x^T grad_support; grad_x = grad_support W^T."""
    g = np.asarray(grad_out, np.float32)
    grad_bias = g.sum(0, dtype=np.float32) if has_bias else None
    grad_support = adj.t_matmul(g)
    grad_w = (np.asarray(x, np.float32).T @ grad_support).astype(np.float32)
    grad_x = (grad_support @ np.asarray(weight, np.float32).T).astype(np.float32) \
        if need_grad_x else None
    return grad_x, grad_w, grad_bias, grad_support


def gc_backward_f64(x, weight, has_bias, adj, grad_out, need_grad_x=True):
    """gc_backward (autograd of layers.py:32-38) evaluated in FLOAT64 on the same float32 inputs:
    the arbiter for gradients that are float32 reductions over the graph's vertices, where the
    float32 reference arithmetic itself is ~1e-5 from exact (tests/conftest.py assert_parity).
    Returns (grad_x, grad_w, grad_bias)."""
    A = sp.csr_matrix((adj.val.astype(np.float64), adj.col, adj.rowptr), shape=adj.shape)
    g = np.asarray(grad_out, np.float64)
    grad_bias = g.sum(0) if has_bias else None
    grad_support = A.T.tocsr() @ g
    grad_w = np.asarray(x, np.float64).T @ grad_support
    grad_x = grad_support @ np.asarray(weight, np.float64).T if need_grad_x else None
    return grad_x, grad_w, grad_bias


# --------------------------------------------------------------------------- upstream 2-layer GCN
def log_softmax(z):
    m = z.max(1, keepdims=True)
    s = z - m
    return (s - np.log(np.exp(s).sum(1, keepdims=True))).astype(np.float32)


def gcn2_forward(x, adj, p, relu_out=False):
    """log_softmax(gc2(relu(gc1(x, adj)), adj)) — models.py:23,48,50(dropout off),68."""
    h1, _ = gc_forward(x, p["gc1.weight"], p.get("gc1.bias"), adj)
    a1 = np.maximum(h1, 0)
    h2, _ = gc_forward(a1, p["gc2.weight"], p.get("gc2.bias"), adj)
    return {"h1": h1, "a1": a1, "h2": h2, "logp": log_softmax(h2)}


def gcn2_loss_backward(x, adj, p, labels, idx, need_grad_x=False, relu_mask=None):
    """nll_loss(logp[idx], labels[idx]) and every gradient of it.  `relu_mask` (bool [n, hidden],
    optional): the ReLU derivative to use instead of (h1 > 0) — for comparing with an
    implementation whose pre-activations differ from this one's by rounding: an element within
    rounding of zero may sit on the other side of the ReLU there, which changes nothing visible in
    the forward pass (the value is ~0 either way) but switches one term of grad_W1 on or off; the
    caller asserts that the two masks differ on such elements only."""
    fw = gcn2_forward(x, adj, p)
    logp = fw["logp"]
    n_tr = len(idx)
    loss = float(-logp[idx, labels[idx]].astype(np.float64).mean())
    grad_logp = np.zeros_like(logp)
    np.add.at(grad_logp, (idx, labels[idx]), np.float32(-1.0 / n_tr))   # (a vertex listed twice counts twice)
    # log_softmax backward: g - softmax * sum(g)
    grad_h2 = (grad_logp - np.exp(logp) * grad_logp.sum(1, keepdims=True)).astype(np.float32)
    ga1, gw2, gb2, _ = gc_backward(fw["a1"], p["gc2.weight"], "gc2.bias" in p, adj, grad_h2)
    grad_h1 = (ga1 * ((fw["h1"] > 0) if relu_mask is None else relu_mask)).astype(np.float32)
    gx, gw1, gb1, _ = gc_backward(x, p["gc1.weight"], "gc1.bias" in p, adj, grad_h1,
                                  need_grad_x=need_grad_x)
    grads = {"gc1.weight": gw1, "gc2.weight": gw2}
    if gb1 is not None:
        grads["gc1.bias"] = gb1
    if gb2 is not None:
        grads["gc2.bias"] = gb2
    return loss, fw, grads, {"grad_h2": grad_h2, "grad_a1": ga1, "grad_h1": grad_h1, "grad_x": gx}


def gcn2_loss_backward_f64(x, adj, p, labels, idx, relu_mask=None):
    """The same training step (gcn2_loss_backward: models.py:23,48,68; train.py:153-157) evaluated
    in FLOAT64 on the same float32 inputs — the arbiter when two float32 results disagree at the
    1e-5 level: on hub columns of 10⁴–10⁵ entries the float32 CPU transpose product (the
    reference's own arithmetic) carries ~1e-5 of rounding error itself.  scipy CSR products and
    numpy GEMMs in float64.  Returns (loss, {"logp"}, grads)."""
    A = sp.csr_matrix((adj.val.astype(np.float64), adj.col, adj.rowptr), shape=adj.shape)
    At = A.T.tocsr()
    x = np.asarray(x, np.float64)
    W1, W2 = np.asarray(p["gc1.weight"], np.float64), np.asarray(p["gc2.weight"], np.float64)
    h1 = A @ (x @ W1)
    if "gc1.bias" in p:
        h1 = h1 + np.asarray(p["gc1.bias"], np.float64)
    mask = (h1 > 0) if relu_mask is None else relu_mask
    a1 = np.where(mask, h1, 0.0) if relu_mask is not None else np.maximum(h1, 0)
    h2 = A @ (a1 @ W2)
    if "gc2.bias" in p:
        h2 = h2 + np.asarray(p["gc2.bias"], np.float64)
    z = h2 - h2.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    n_tr = len(idx)
    loss = float(-logp[idx, labels[idx]].mean())
    g = np.zeros_like(logp)
    np.add.at(g, (idx, labels[idx]), -1.0 / n_tr)
    grad_h2 = g - np.exp(logp) * g.sum(1, keepdims=True)
    gs2 = At @ grad_h2
    grads = {"gc2.weight": a1.T @ gs2}
    if "gc2.bias" in p:
        grads["gc2.bias"] = grad_h2.sum(0)
    grad_h1 = (gs2 @ W2.T) * mask
    gs1 = At @ grad_h1
    grads["gc1.weight"] = x.T @ gs1
    if "gc1.bias" in p:
        grads["gc1.bias"] = grad_h1.sum(0)
    return loss, {"logp": logp}, grads


class Adam:
    """torch.optim.Adam(lr, betas=(.9,.999), eps=1e-8, weight_decay) — train.py:111-112."""

    def __init__(self, params, lr=0.01, weight_decay=5e-4, betas=(0.9, 0.999), eps=1e-8):
        self.p, self.lr, self.wd, self.betas, self.eps = params, lr, weight_decay, betas, eps
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}
        self.t = 0

    def step(self, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k, w in self.p.items():
            g = grads[k] + np.float32(self.wd) * w
            self.m[k] = (b1 * self.m[k] + (1 - b1) * g).astype(np.float32)
            self.v[k] = (b2 * self.v[k] + (1 - b2) * g * g).astype(np.float32)
            denom = (np.sqrt(self.v[k]) / np.float32(np.sqrt(bc2)) + np.float32(self.eps))
            self.p[k] = (w - np.float32(self.lr / bc1) * self.m[k] / denom).astype(np.float32)


def train_trajectory(x, adj, params, labels, idx_train, epochs, lr=0.01, weight_decay=5e-4):
    p = {k: np.array(v, np.float32) for k, v in params.items()}
    opt = Adam(p, lr=lr, weight_decay=weight_decay)
    losses, accs = [], []
    for _ in range(epochs):
        loss, fw, grads, _ = gcn2_loss_backward(x, adj, opt.p, labels, idx_train)
        accs.append(accuracy(fw["logp"][idx_train], labels[idx_train]))
        losses.append(loss)
        opt.step(grads)
    return np.array(losses), np.array(accs), opt.p


# ---------------------------------------------------------------- fused dropout (test checker)
def dropout_scale(p):
    """Scale of the kept elements: 1 / (keep probability of the QUANTISED threshold dropout_keep
    uses) = 65536 / (65536 - thresh), as float32 — 1 / (1 - p) at p = 1/2, within 2^-17 elsewhere
    (include/gcn_spmm.h, struct gcn_epilogue; upstream's F.dropout scales by 1 / (1 - p),
    pygcn/models.py:50)."""
    t = int(np.float64(np.float32(p)) * 65536.0 + 0.5)
    thresh = min(65535, max(1, t))
    return np.float32(65536.0) / np.float32(65536 - thresh)


def dropout_keep(seed, rows, F, p, row_base=0):
    """numpy restatement of the kernels' dropout keep function (include/gcn_spmm.h, struct
    gcn_epilogue, ABI 23) — the stand-in for the mask `F.dropout` draws in the reference model
    (pygcn/models.py:50 upstream).  bool [len(rows), F]; with thresh = clamp(round(p * 65536), 1,
    65535) and row = rows[i] + row_base, element (row, f):
      thresh != 32768 — eight 16-bit fields per Philox call:
        block = ((f >> 4) << 1) | ((f >> 2) & 1),   field = (((f >> 3) & 1) << 2) | (f & 3)
        w     = Philox4x32-10(counter = (row_lo, row_hi, block, 0), key = (seed_lo, seed_hi))
        keep  = ((w[field >> 1] >> 16 * (field & 1)) & 0xFFFF) >= thresh
      thresh == 32768 (p = 1/2, the reference's default: one bit decides) — 128 one-bit fields per call:
        block = ((f >> 8) << 1) | ((f >> 2) & 1),   index = (((f & 255) >> 3) << 2) | (f & 3)
        keep  = (w[index >> 5] >> (index & 31)) & 1
    Both are the same layout rule — a call covers the columns {2E·c + 8·q + 4·b + (0..3)} of one
    parity b of bit 2 of f, E = 8 or 128 fields — so that one lane of the MFMA GEMMs' transposed
    accumulator tile finds all its columns of a span in one call."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    mask32 = np.uint64(0xFFFFFFFF)
    rows = np.asarray(rows, np.int64) + np.int64(row_base)
    f = np.arange(F, dtype=np.int64)
    t = int(np.float64(np.float32(p)) * 65536.0 + 0.5)
    thresh = min(65535, max(1, t))
    one_bit = thresh == 32768
    if one_bit:
        blk = ((f >> 8) << 1) | ((f >> 2) & 1)
        fld = (((f & 255) >> 3) << 2) | (f & 3)                               # 0..127
    else:
        blk = ((f >> 4) << 1) | ((f >> 2) & 1)
        fld = (((f >> 3) & 1) << 2) | (f & 3)                                 # 0..7
    blocks = np.unique(blk)
    r = np.repeat(rows.astype(np.uint64), len(blocks))
    q = np.tile(blocks.astype(np.uint64), len(rows))
    c = [r & mask32, r >> np.uint64(32), q, np.zeros_like(q)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask32, p1 >> np.uint64(32), p1 & mask32
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + np.uint64(W0)) & mask32, (k1 + np.uint64(W1)) & mask32
    words = np.stack(c, axis=1).reshape(len(rows), len(blocks), 4)          # [row, block, word]
    bpos = np.searchsorted(blocks, blk)
    if one_bit:
        w = words[:, bpos, fld >> 5]                                          # [row, F]
        return ((w >> (fld & 31).astype(np.uint64)) & np.uint64(1)) != 0
    w = words[:, bpos, fld >> 1]                                              # [row, F]
    bits = (w >> (np.uint64(16) * (fld & 1).astype(np.uint64))) & np.uint64(0xFFFF)
    return bits >= np.uint64(thresh)
