"""pygcn_amd — MI355X-native (gfx950) implementation of the GraphConvolution hot path of
LinChen-65/pygcn: the normalized-adjacency SpMM forward and its transpose product backward,
behind the reference's `GraphConvolution` / `GCN` module surface.

Flat-module use (the reference's convention, pygcn/models.py:4, train.py:15-16) works with
cwd = this directory: `from layers import GraphConvolution`, `from models import GCN`,
`from utils import load_data, accuracy`.  Package use: `from pygcn_amd import GCN, ...`.
"""
from .graph import CSRGraph, as_graph
from .spmm import spmm_csr
from .ops import sparse_mm          # also registers torch.ops.pygcn_amd.spmm_csr
from .layers import GraphConvolution
from .models import GCN

# `pygcn_amd.spmm` is the MODULE (kernels' Python launchers + autograd nodes); the drop-in for
# `torch.spmm` / `torch.sparse.mm` is exported as `sparse_mm`.
__all__ = ["CSRGraph", "as_graph", "sparse_mm", "spmm_csr", "GraphConvolution", "GCN"]
