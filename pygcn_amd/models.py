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


class GCN(nn.Module):
    def __init__(self, nfeat, nhid, nclass, dropout):
        super(GCN, self).__init__()
        self.gc1 = GraphConvolution(nfeat, nhid)
        self.gc2 = GraphConvolution(nhid, nclass)
        self.dropout = dropout

    def forward(self, x, adj):
        # F.dropout(F.relu(gc1(x, adj)), p, training) with ReLU and dropout fused into the SpMM store
        x = self.gc1(x, adj, relu=True, dropout=self.dropout if self.training else 0.0)
        # F.log_softmax(gc2(x, adj), dim=1) — in the SpMM's store when a row fits one wavefront
        # (dim=1 for the reference's [N, C]; the last dim if batched)
        return self.gc2(x, adj, log_softmax=True)


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
