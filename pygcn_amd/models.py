"""GCN — the 2-layer model of upstream pygcn, which the reference fork keeps as comments around
its 3-layer edit (reference pygcn/models.py:23 `gc2 = GraphConvolution(nhid, nclass)`, :48
`F.relu(self.gc1(x, adj))`, :50 `F.dropout`, :68 `F.log_softmax(x, dim=1)`).

Parameter names gc1.weight/bias, gc2.weight/bias as in the reference (models.py:21-26).
"""
import os
import sys

import torch.nn as nn

if not __package__:   # flat import, the reference's convention (`from models import GCN`)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from layers import GraphConvolution
else:
    from pygcn_amd.layers import GraphConvolution
from pygcn_amd.tuning import ROWGRAD_MIN_ROWS  # noqa: E402


class GCN(nn.Module):
    def __init__(self, nfeat, nhid, nclass, dropout):
        super(GCN, self).__init__()
        self.gc1 = GraphConvolution(nfeat, nhid)
        self.gc2 = GraphConvolution(nhid, nclass)
        self.dropout = dropout

    def forward(self, x, adj, rows=None, keep_full=False):
        """`model(x, adj)`: the reference call — log-probabilities of every vertex.

        `model(x, adj, rows=idx)` (extension): the same forward pass, returning `output[idx]` —
        what upstream's epoch feeds to the loss, `F.nll_loss(output[idx_train], labels[idx_train])`
        (pygcn/train.py:153).  Telling the model which rows the loss reads lets the whole backward
        pass run on the rows that can be non-zero (pygcn_amd/fused.py): same gradients, no
        [N, ·]-sized zero fills, scatters or sweeps, no host synchronisation.  With
        `keep_full=True` the result is `(output[idx], output.detach())` — the full matrix for
        validation on other rows, as upstream's --fastmode uses it."""
        if rows is not None:
            return self._forward_rows(x, adj, rows, keep_full)
        # (below ROWGRAD_MIN_ROWS vertices an epoch is launch-bound — Cora: ~1 ms — and the plain
        #  layer-by-layer composition issues fewer launches than the node's generic fallbacks)
        graph = self._one_node_graph(x, adj) if (x.dim() == 2 and x.shape[0] >= ROWGRAD_MIN_ROWS) else None
        if graph is not None:
            # the whole model as ONE autograd node (pygcn_amd/fused.py): its backward pass takes the
            # row-restricted route when the caller selected `output[idx_train]` (upstream's next
            # line, pygcn/train.py:153) and the full-height route for a loss over all vertices
            fused, _, _, dropout_seed_for = self._imports()
            p = self.dropout if self.training else 0.0
            out = fused.gcn2_full(x, self.gc1, self.gc2, graph, p, dropout_seed_for(x) if p > 0.0 else 0)
        else:
            # F.dropout(F.relu(gc1(x, adj)), p, training) with ReLU and dropout fused into the SpMM store
            h = self.gc1(x, adj, relu=True, dropout=self.dropout if self.training else 0.0)
            # F.log_softmax(gc2(x, adj), dim=1) — in the SpMM's store when a row fits one wavefront
            # (dim=1 for the reference's [N, C]; the last dim if batched)
            out = self.gc2(h, adj, log_softmax=True)
        if (self.training and out.requires_grad and out.dim() == 2 and out.is_cuda
                and out.shape[0] >= ROWGRAD_MIN_ROWS     # (pygcn_amd/tuning.py)
                and type(adj).__name__ != "ShardedGraph"):
            # upstream's next line is `output[idx_train]`: let that selection hand the backward pass
            # the rows instead of a dense gradient (pygcn_amd/rowgrad.py)
            from pygcn_amd.rowgrad import RowSelectable
            out = out.as_subclass(RowSelectable)
        return out

    @staticmethod
    def _imports():
        import pygcn_amd.fused as fused
        from pygcn_amd.graph import CSRGraph, as_graph
        from pygcn_amd.spmm import dropout_seed_for
        return fused, CSRGraph, as_graph, dropout_seed_for

    def _one_node_graph(self, x, adj):
        """The prepared graph handle if the one-node path covers this call, else None (sharded /
        dense adjacency, batched input, class counts the fused log_softmax does not take)."""
        import torch
        fused, CSRGraph, as_graph, _ = self._imports()
        graph = adj
        if isinstance(adj, torch.Tensor) and adj.layout in (torch.sparse_coo, torch.sparse_csr) \
                and adj.is_cuda:
            graph = as_graph(adj)
        if (isinstance(graph, CSRGraph) and isinstance(x, torch.Tensor) and x.dim() == 2
                and x.dtype == self.gc1.weight.dtype and self.gc1.weight.is_cuda
                and fused.fusable(self.gc2.weight.dtype, self.gc2.out_features, graph, x)):
            return graph
        return None

    def _forward_rows(self, x, adj, rows, keep_full):
        graph = self._one_node_graph(x, adj)
        if graph is not None:
            fused, _, _, dropout_seed_for = self._imports()
            p = self.dropout if self.training else 0.0
            seed = dropout_seed_for(x) if p > 0.0 else 0
            out, full = fused.gcn2_rows(x, self.gc1, self.gc2, graph, rows, p, seed, keep_full)
            return (out, full) if keep_full else out
        full = self.forward(x, adj)          # layer-by-layer path (sharded / dense adjacency / odd shapes)
        return (full[rows], full.detach()) if keep_full else full[rows]


class GCNStack(nn.Module):
    """k x (GraphConvolution + ReLU), the shape of the fork's GeneratorGCN
    (reference pygcn/models.py:74-124: gc1..gc3, ReLU after each, no BatchNorm)."""

    def __init__(self, nfeat, nhid, nclass, dropout=0.0, nlayers=3):
        super(GCNStack, self).__init__()
        dims = [nfeat] + [nhid] * (nlayers - 1) + [nclass]
        for i in range(nlayers):
            setattr(self, f"gc{i + 1}", GraphConvolution(dims[i], dims[i + 1]))
        self.nlayers = nlayers
        self.dropout = dropout

    def forward(self, x, adj):
        for i in range(self.nlayers):
            x = getattr(self, f"gc{i + 1}")(x, adj, relu=True)
        return x
