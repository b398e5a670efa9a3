#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference on CPU.

Run once in the build container (the only place /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

What is executed from the reference (never copied): `layers.GraphConvolution`
(pygcn/layers.py:7-43), `models.GeneratorGCN` (pygcn/models.py:74-124),
`utils.normalize` / `utils.sparse_mx_to_torch_sparse_tensor` / `utils.accuracy`
(pygcn/utils.py:390-414).  The Cora recipe that the fork keeps only as a comment
(pygcn/utils.py:356-368) is re-stated here around those live helpers.  The arithmetic itself
is PyTorch's CPU `torch.mm` / `torch.spmm` (torch 2.10.0+rocm7.0 in this image).

Outputs are pure data (.npz): inputs that cannot be regenerated from a seed (the Cora
adjacency, which comes from the reference's data file data/cora/cora.cites) and expected
outputs.  Inputs that are seeded live in tests/golden/inputs.py.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PYGCN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, HERE)
import inputs as gin  # noqa: E402

_cwd = os.getcwd()
os.chdir(os.path.join(REF, "pygcn"))          # the reference imports flat, relative to cwd
sys.path.insert(0, os.getcwd())
import layers as ref_layers  # noqa: E402
import models as ref_models  # noqa: E402
import utils as ref_utils  # noqa: E402
os.chdir(_cwd)

torch.set_num_threads(1)   # fixed summation order inside MKL for reproducible fixtures


def to_np(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------- G1: parameter init
def g1_init():
    out = {}
    torch.manual_seed(42)
    gc1 = ref_layers.GraphConvolution(gin.CORA_NFEAT, gin.CORA_NHID)
    gc2 = ref_layers.GraphConvolution(gin.CORA_NHID, gin.CORA_NCLASS)
    out["gc1_weight"], out["gc1_bias"] = to_np(gc1.weight), to_np(gc1.bias)
    out["gc2_weight"], out["gc2_bias"] = to_np(gc2.weight), to_np(gc2.bias)
    torch.manual_seed(7)
    gcn = ref_layers.GraphConvolution(256, 256, bias=False)
    out["nobias_weight_head"] = to_np(gcn.weight)[:4]
    out["repr"] = np.array(repr(gc1))
    return out, gc1, gc2


# ---------------------------------------------------------------- Cora adjacency
def cora_adjacency():
    """Recipe of pygcn/utils.py:356-368 on data/cora/cora.cites; node order = np.unique(ids)
    (cora.content, which fixed the upstream order, is absent: .MISSING_LARGE_BLOBS)."""
    cites = np.genfromtxt(os.path.join(REF, "data/cora/cora.cites"), dtype=np.int32)
    ids = np.unique(cites)
    idx_map = {j: i for i, j in enumerate(ids)}
    edges = np.array([idx_map[v] for v in cites.flatten()], dtype=np.int32).reshape(cites.shape)
    n = len(ids)
    adj = sp.coo_matrix((np.ones(edges.shape[0]), (edges[:, 0], edges[:, 1])),
                        shape=(n, n), dtype=np.float32)
    adj = adj + adj.T.multiply(adj.T > adj) - adj.multiply(adj.T > adj)
    adj = ref_utils.normalize(adj + sp.eye(adj.shape[0]))
    adj_t = ref_utils.sparse_mx_to_torch_sparse_tensor(adj)
    return edges, adj_t


# ---------------------------------------------------------------- G2: Cora one step
def g2_cora(gc1, gc2, adj_t):
    x = torch.from_numpy(gin.cora_features()).requires_grad_(True)
    labels = torch.from_numpy(gin.cora_labels())
    idx_train = torch.from_numpy(gin.cora_splits()[0])
    h1 = gc1(x, adj_t)
    h1.retain_grad()
    a1 = F.relu(h1)
    a1.retain_grad()
    h2 = gc2(a1, adj_t)
    h2.retain_grad()
    logp = F.log_softmax(h2, dim=1)
    loss = F.nll_loss(logp[idx_train], labels[idx_train])
    loss.backward()
    return {
        "h1": to_np(h1), "h2": to_np(h2), "logp": to_np(logp), "loss": to_np(loss),
        "grad_h2": to_np(h2.grad), "grad_a1": to_np(a1.grad), "grad_h1": to_np(h1.grad),
        "grad_x_head": to_np(x.grad)[:64],
        "grad_x_abs_sum": np.float64(x.grad.double().abs().sum().item()),
        "gc1_weight_grad": to_np(gc1.weight.grad), "gc1_bias_grad": to_np(gc1.bias.grad),
        "gc2_weight_grad": to_np(gc2.weight.grad), "gc2_bias_grad": to_np(gc2.bias.grad),
        "features_abs_sum": np.float64(np.abs(gin.cora_features()).sum(dtype=np.float64)),
    }


# ---------------------------------------------------------------- G3: GeneratorGCN stack
def g3_generator():
    n, nfeat, nhid, nclass = 64, 8, 32, 32
    rows, cols, vals = gin.random_coo(n, n, 400, seed=300)
    adj = torch.sparse_coo_tensor(np.vstack([rows, cols]), vals, (n, n))
    x = torch.from_numpy(gin.dense((n, nfeat), 301)).requires_grad_(True)
    torch.manual_seed(3)
    m = ref_models.GeneratorGCN(nfeat, nhid, nclass, 0.5, 5)
    y = m(x, adj)
    g = torch.from_numpy(gin.dense((n, nclass), 302))
    y.backward(g)
    out = {"y": to_np(y), "grad_x": to_np(x.grad)}
    for name, p in m.named_parameters():
        out["param_" + name] = to_np(p)
        out["grad_" + name] = to_np(p.grad)
    return out


# ---------------------------------------------------------------- G4: edge cases via the layer
from make_golden_cases import EDGE_CASES  # noqa: E402


def g4_edge_cases():
    out = {}
    for k, (name, nr, nc, nnz, fin, fout, bias, kw) in enumerate(EDGE_CASES):
        seed = 400 + 10 * k
        rows, cols, vals = gin.random_coo(nr, nc, nnz, seed=seed, **kw)
        adj = torch.sparse_coo_tensor(np.vstack([rows, cols]), vals, (nr, nc))
        x = torch.from_numpy(gin.dense((nc, fin), seed + 1)).requires_grad_(True)
        torch.manual_seed(seed)
        layer = ref_layers.GraphConvolution(fin, fout, bias=bias)
        y = layer(x, adj)
        g = torch.from_numpy(gin.dense((nr, fout), seed + 2))
        y.backward(g)
        out[name + "/weight"] = to_np(layer.weight)
        if bias:
            out[name + "/bias"] = to_np(layer.bias)
            out[name + "/grad_bias"] = to_np(layer.bias.grad)
        out[name + "/y"] = to_np(y)
        out[name + "/grad_weight"] = to_np(layer.weight.grad)
        out[name + "/grad_x"] = to_np(x.grad)
        # the bare sparse product and its transpose product (a4 / a6), fp32 and fp64
        support = (x @ layer.weight).detach()
        out[name + "/spmm"] = to_np(torch.spmm(adj, support))
        out[name + "/spmm_t"] = to_np(torch.spmm(adj.t(), g))
        if name == "f256_hub":   # fp64 product kept for error attribution on the skewed case
            out[name + "/spmm_f64"] = to_np(torch.spmm(adj.double(), support.double()))
    return out


# ---------------------------------------------------------------- G5: 200-epoch trajectory
def g5_trajectory(adj_t, epochs=200):
    """Upstream-semantics 2-layer GCN (shape pinned by the comments at pygcn/models.py:23,48,
    50,68) assembled from the IMPORTED GraphConvolution; dropout 0 so CPU and GPU runs are
    comparable; Adam lr 0.01 wd 5e-4, seed 42 (pygcn/train.py:41-47,111-112)."""
    torch.manual_seed(42)
    gc1 = ref_layers.GraphConvolution(gin.CORA_NFEAT, gin.CORA_NHID)
    gc2 = ref_layers.GraphConvolution(gin.CORA_NHID, gin.CORA_NCLASS)
    params = list(gc1.parameters()) + list(gc2.parameters())
    opt = torch.optim.Adam(params, lr=0.01, weight_decay=5e-4)
    x = torch.from_numpy(gin.cora_features())
    labels = torch.from_numpy(gin.cora_labels())
    idx_train, idx_val, _ = [torch.from_numpy(i) for i in gin.cora_splits()]
    losses, accs, vlosses = [], [], []
    for _ in range(epochs):
        opt.zero_grad()
        out = F.log_softmax(gc2(F.relu(gc1(x, adj_t)), adj_t), dim=1)
        loss = F.nll_loss(out[idx_train], labels[idx_train])
        acc = ref_utils.accuracy(out[idx_train], labels[idx_train])
        loss.backward()
        opt.step()
        losses.append(loss.item())
        accs.append(acc.item())
        vlosses.append(F.nll_loss(out[idx_val], labels[idx_val]).item())
    return {
        "loss_train": np.array(losses, np.float64), "acc_train": np.array(accs, np.float64),
        "loss_val": np.array(vlosses, np.float64),
        "final_gc1_weight": to_np(gc1.weight), "final_gc1_bias": to_np(gc1.bias),
        "final_gc2_weight": to_np(gc2.weight), "final_gc2_bias": to_np(gc2.bias),
    }


def main():
    g1, gc1, gc2 = g1_init()
    np.savez_compressed(os.path.join(HERE, "g1_init.npz"), **g1)

    edges, adj_t = cora_adjacency()
    adj_c = adj_t.coalesce()
    csr = sp.csr_matrix((to_np(adj_t._values()), to_np(adj_t._indices())), shape=tuple(adj_t.shape))
    csr.sort_indices()
    np.savez_compressed(
        os.path.join(HERE, "cora_graph.npz"),
        edges=edges.astype(np.int32),                       # re-indexed cora.cites pairs (data)
        coo_row=to_np(adj_t._indices()[0]).astype(np.int32),  # exactly as the reference emits it
        coo_col=to_np(adj_t._indices()[1]).astype(np.int32),
        coo_val=to_np(adj_t._values()),
        csr_rowptr=csr.indptr.astype(np.int32), csr_col=csr.indices.astype(np.int32),
        csr_val=csr.data.astype(np.float32),
        n=np.int64(adj_t.shape[0]), nnz=np.int64(adj_c._nnz()),
    )
    np.savez_compressed(os.path.join(HERE, "g2_cora_step.npz"), **g2_cora(gc1, gc2, adj_t))
    np.savez_compressed(os.path.join(HERE, "g3_generator_gcn.npz"), **g3_generator())
    np.savez_compressed(os.path.join(HERE, "g4_edge_cases.npz"), **g4_edge_cases())
    np.savez_compressed(os.path.join(HERE, "g5_trajectory.npz"), **g5_trajectory(adj_t))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
    print("torch", torch.__version__, "nnz", adj_c._nnz())


if __name__ == "__main__":
    main()
