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
from pygcn_amd.spmm import (DenseMMFunction, GraphConvFunction, SpMMFunction,  # noqa: E402
                            dropout_seed_for, log_softmax_fusable)
from pygcn_amd.sharded import (ShardedGraph, ShardedHiddenLayerFunction,  # noqa: E402
                               ShardedInputLayerFunction, ShardedSpMMFunction)


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

    def forward(self, input, adj, relu=False, dropout=0.0, log_softmax=False):
        """support = input @ W (MFMA GEMM via torch.mm), output = adj @ support (HIP SpMM),
        + bias fused into the SpMM's store.  `adj`: CSRGraph, torch sparse COO/CSR (converted
        once and cached on the tensor), a ShardedGraph (multi-GPU row block), or a dense [N,N]
        tensor (the fork's live scripts pass a dense adjacency, utils.py:124-131 — torch.spmm is
        a dense GEMM there, and so is this).

        `relu=True` / `dropout=p` (extensions; the defaults are the reference's behaviour) apply
        the ReLU and the training-mode dropout that follow the layer in the model (models.py:48,50
        upstream) inside the kernel's store; `dropout` requires `relu`.  `log_softmax=True` returns
        F.log_softmax(output, dim=-1) (models.py:52 upstream), computed in the same store when the
        row fits one wavefront and by torch otherwise."""
        if log_softmax:
            if relu or dropout > 0.0:
                raise RuntimeError("log_softmax cannot be combined with relu / dropout")
            if (input.dim() == 2 and not isinstance(adj, ShardedGraph) and input.is_cuda
                    and not (isinstance(adj, torch.Tensor) and adj.layout == torch.strided)
                    and log_softmax_fusable(self.out_features, input.dtype)):
                _require_cuda(self.weight, "GraphConvolution.weight (call model.cuda())")
                return GraphConvFunction.apply(input, self.weight, self.bias, as_graph(adj), False,
                                               0.0, 0, True)
            if (isinstance(adj, ShardedGraph) and input.dim() == 2 and adj.compress_hidden
                    and not adj.is_constant_input(input) and input.shape[1] % 32 == 0):
                # hidden activation, compressed exchange (ShardedGraph(compress_hidden=True))
                fuse = adj._hinted_product and input.is_cuda and log_softmax_fusable(self.out_features, input.dtype)
                out = ShardedHiddenLayerFunction.apply(adj, input, self.weight, self.bias, False, 0.0, 0,
                                                       fuse, True)
                return out if fuse else torch.nn.functional.log_softmax(out, dim=-1)
            if (isinstance(adj, ShardedGraph) and input.dim() == 2 and input.is_cuda and adj._hinted_product
                    and not adj.is_constant_input(input)
                    and log_softmax_fusable(self.out_features, input.dtype)):
                # last layer of a sharded model: log_softmax in the store of the launch that
                # completes the rank's rows (after the halo rows have arrived)
                return ShardedSpMMFunction.apply(adj, DenseMMFunction.apply(input, self.weight),
                                                 self.bias, False, 0.0, 0, True)
            if isinstance(adj, ShardedGraph) and input.dim() == 2 and not adj.is_constant_input(input):
                # (a class count the fused log_softmax does not take: still the model's LAST layer —
                #  the declared loss rows bound its gradient's rows)
                out = ShardedSpMMFunction.apply(adj, DenseMMFunction.apply(input, self.weight),
                                                self.bias, False, 0.0, 0, False, True)
                return torch.nn.functional.log_softmax(out, dim=-1)
            return torch.nn.functional.log_softmax(self.forward(input, adj), dim=-1)
        if input.dim() == 3:
            out = self._forward_batched(input, adj, relu)
            return torch.nn.functional.dropout(out, dropout, True) if dropout > 0.0 else out
        seed = dropout_seed_for(input) if dropout > 0.0 else 0
        if isinstance(adj, ShardedGraph):
            # row-block shard of a multi-GPU run: exchange + local HIP SpMM (pygcn_amd/sharded.py)
            if adj.is_constant_input(input):     # feature block: halo rows held, no exchange
                return ShardedInputLayerFunction.apply(adj, input, adj.constant_halo(input),
                                                       self.weight, self.bias, relu, dropout, seed)
            if adj.compress_hidden and input.dim() == 2 and input.shape[1] % 32 == 0:
                return ShardedHiddenLayerFunction.apply(adj, input, self.weight, self.bias, relu, dropout, seed)
            return ShardedSpMMFunction.apply(adj, DenseMMFunction.apply(input, self.weight),
                                             self.bias, relu, dropout, seed)
        _require_cuda(input, "input")
        _require_cuda(self.weight, "GraphConvolution.weight (call model.cuda())")
        if isinstance(adj, torch.Tensor) and adj.layout == torch.strided:
            output = torch.mm(adj, torch.mm(input, self.weight))
            output = output + self.bias if self.bias is not None else output
            output = torch.relu(output) if relu else output
            return torch.nn.functional.dropout(output, dropout, True) if dropout > 0.0 else output
        return GraphConvFunction.apply(input, self.weight, self.bias, as_graph(adj), relu, dropout,
                                       seed)

    def _forward_batched(self, input, adj, relu):
        """k samples over the same graph in ONE sparse product.  The fork runs
        its GCN once per sample in a Python loop ("cannot batch yet", reference
        pygcn/models.py:343-349); here input is [k, N, Fin]: the k supports are laid side by side
        as [N, k·Fout], so every gathered row of the dense operand is k·Fout wide (k launches of
        narrow rows become one launch of wide rows), and the result is returned as [k, N, Fout]."""
        k, n, _ = input.shape
        support = torch.matmul(input, self.weight)                       # [k, N, Fout]
        wide = support.permute(1, 0, 2).reshape(n, k * self.out_features)
        bias = self.bias.repeat(k) if self.bias is not None else None
        if isinstance(adj, torch.Tensor) and adj.layout == torch.strided:
            out = torch.mm(adj, wide)
            out = out + bias if bias is not None else out
            out = torch.relu(out) if relu else out
        elif isinstance(adj, ShardedGraph):
            out = ShardedSpMMFunction.apply(adj, wide.contiguous(), bias, relu)
        else:
            _require_cuda(input, "input")
            out = SpMMFunction.apply(as_graph(adj), wide, bias, relu)
        return out.view(n, k, self.out_features).permute(1, 0, 2)

    def __repr__(self):
        return self.__class__.__name__ + ' (' \
               + str(self.in_features) + ' -> ' \
               + str(self.out_features) + ')'
