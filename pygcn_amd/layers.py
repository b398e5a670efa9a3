"""GraphConvolution — same surface as the reference layer (reference pygcn/layers.py:7-43),
with the sparse product running on the MI355X HIP kernel instead of `torch.spmm`.

    GraphConvolution(in_features, out_features, bias=True)      layers.py:12
    .in_features .out_features .weight[in,out] .bias[out]|None  layers.py:14-20
    .reset_parameters()                                         layers.py:23-29
    .forward(input, adj)                                        layers.py:32-38
    repr -> "GraphConvolution (in -> out)"                      layers.py:40-43

state_dict keys (`weight`, `bias`) and shapes are the reference's, so its checkpoints load.
"""
import math
import os
import sys

import torch
from torch.nn.modules.module import Module
from torch.nn.parameter import Parameter

if not __package__:   # imported flat, the reference's convention (`from layers import ...`)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygcn_amd.graph import CSRGraph, _require_cuda, as_graph  # noqa: E402
from pygcn_amd.spmm import DenseMMFunction, SpMMFunction  # noqa: E402
from pygcn_amd.sharded import ShardedGraph, ShardedSpMMFunction  # noqa: E402


class GraphConvolution(Module):
    """Simple GCN layer, similar to https://arxiv.org/abs/1609.02907 (MI355X SpMM inside)."""

    def __init__(self, in_features, out_features, bias=True):
        super(GraphConvolution, self).__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = Parameter(torch.empty(in_features, out_features, dtype=torch.float32))
        if bias:
            self.bias = Parameter(torch.empty(out_features, dtype=torch.float32))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        # same RNG stream as the reference: kaiming_uniform_(weight) then bias.uniform_(+-stdv)
        # (layers.py:24,27,29); fan_in is taken from size(1) = out_features by torch.
        stdv = 1. / math.sqrt(self.weight.size(1))
        torch.nn.init.kaiming_uniform_(self.weight)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj, relu=False):
        """support = input @ W (MFMA GEMM via torch.mm), output = adj @ support (HIP SpMM),
        + bias fused into the SpMM's store.  `adj`: CSRGraph, torch sparse COO/CSR (converted
        once and cached on the tensor), a ShardedGraph (multi-GPU row block), or a dense [N,N]
        tensor (the fork's live scripts pass a dense adjacency, utils.py:124-131 — torch.spmm is
        a dense GEMM there, and so is this).

        `relu=True` (an extension; default is the reference's behaviour) applies the ReLU that
        follows the layer in the model (models.py:48 upstream) inside the kernel's store."""
        if isinstance(adj, ShardedGraph):
            # row-block shard of a multi-GPU run: all-gather + local HIP SpMM (pygcn_amd/sharded.py)
            return ShardedSpMMFunction.apply(adj, DenseMMFunction.apply(input, self.weight),
                                             self.bias, relu)
        _require_cuda(input, "input")
        _require_cuda(self.weight, "GraphConvolution.weight (call model.cuda())")
        if isinstance(adj, torch.Tensor) and adj.layout == torch.strided:
            output = torch.mm(adj, torch.mm(input, self.weight))
            output = output + self.bias if self.bias is not None else output
            return torch.relu(output) if relu else output
        support = DenseMMFunction.apply(input, self.weight)
        return SpMMFunction.apply(as_graph(adj), support, self.bias, relu)

    def __repr__(self):
        return self.__class__.__name__ + ' (' \
               + str(self.in_features) + ' -> ' \
               + str(self.out_features) + ')'
