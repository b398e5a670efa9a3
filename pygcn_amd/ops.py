"""`torch.ops.pygcn_amd.spmm_csr` — the sparse x dense product of the GraphConvolution layer as a
REGISTERED PyTorch operator ("operator / FFI" row).

Replaces the call `torch.spmm(adj, support)` at reference pygcn/layers.py:34 and, through the
registered autograd formula, the product `adj.t() @ grad_output` PyTorch's `mm` derivative runs for
it when `loss.backward()` is called (pygcn/train.py:157):

    C       = spmm_csr(rowptr, col, val, B, bias, n_cols, relu)          A · B (+ bias, ReLU)
    grad_B  = spmm_csr(rowptr_T, col_T, val_T, grad_C, None, n_rows)     Aᵀ · grad_C
    grad_b  = column sums of grad_C

The operator takes plain tensors (CSR arrays of Â on the HIP device), so it is visible to the
dispatcher, `torch.library.opcheck`, and `torch.compile` (a fake/meta kernel is registered).  The
HIP kernel behind it is the C-ABI launcher `gcn_spmm_csr_ep` (include/gcn_spmm.h); the schedule
and CSR(Âᵀ) for a given set of arrays are built once and cached (pygcn_amd/graph.py,
`graph_for_arrays`).  There is NO CPU kernel: calling the operator with CPU tensors raises.
"""
from typing import Optional

import torch

from . import spmm as _spmm
from .graph import CSRGraph, graph_for_arrays

_NO_CPU = ("pygcn_amd::spmm_csr must run on a HIP device: the MI355X path has no CPU "
           "implementation; move the tensors with .cuda().")


@torch.library.custom_op("pygcn_amd::spmm_csr", mutates_args=(), device_types="cuda")
def spmm_csr_op(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, B: torch.Tensor,
                bias: Optional[torch.Tensor], n_cols: int, relu: bool = False) -> torch.Tensor:
    graph = graph_for_arrays(rowptr, col, val, (rowptr.numel() - 1, n_cols))
    return _spmm.spmm_csr(graph, B, bias=bias, relu=relu)


@spmm_csr_op.register_kernel("cpu")
def _(rowptr, col, val, B, bias, n_cols, relu=False):
    raise RuntimeError(_NO_CPU)


@spmm_csr_op.register_fake
def _(rowptr, col, val, B, bias, n_cols, relu=False):
    torch._check(B.dim() == 2, lambda: "dense operand must be 2-D")
    torch._check(B.shape[0] == n_cols,
                 lambda: f"size mismatch, adj [*, {n_cols}] x dense {tuple(B.shape)}")
    return B.new_empty((rowptr.shape[0] - 1, B.shape[1]))


@torch.library.custom_op("pygcn_amd::csr_transpose", mutates_args=(), device_types="cuda")
def csr_transpose_op(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor,
                     n_cols: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CSR(Aᵀ) of the adjacency held in these arrays: (rowptr_T [n_cols+1], col_T, val_T), built by
    the native device transpose (`gcn_csr_transpose_device`) the first time and cached with the
    prepared adjacency — PyTorch re-derives the transposed operand on every backward call of
    `torch.spmm`; here it is a lookup."""
    gt = graph_for_arrays(rowptr, col, val, (rowptr.numel() - 1, n_cols)).t()
    # (fresh tensor objects over the cached storage: an operator must not hand out its inputs or
    #  the same Python object twice; nothing ever writes to these arrays)
    return gt.rowptr.view(-1), gt.col.view(-1), gt.val.view(-1)


@csr_transpose_op.register_kernel("cpu")
def _(rowptr, col, val, n_cols):
    raise RuntimeError(_NO_CPU.replace("spmm_csr", "csr_transpose"))


@csr_transpose_op.register_fake
def _(rowptr, col, val, n_cols):
    return rowptr.new_empty((n_cols + 1,)), col.new_empty(col.shape), val.new_empty(val.shape)


@torch.library.custom_op("pygcn_amd::sddmm_csr", mutates_args=(), device_types="cuda")
def sddmm_csr_op(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, G: torch.Tensor,
                 B: torch.Tensor, n_cols: int) -> torch.Tensor:
    """grad_val[e] = < G[row(e), :], B[col[e], :] > on the pattern of the adjacency held in these
    arrays (C-ABI gcn_sddmm_csr): the gradient of the adjacency values of spmm_csr — only computed
    when a caller sets `val.requires_grad` (the reference never does: pygcn/train.py:80,123)."""
    graph = graph_for_arrays(rowptr, col, val, (rowptr.numel() - 1, n_cols))
    return _spmm.sddmm_csr(graph, G, B).to(val.dtype)


@sddmm_csr_op.register_kernel("cpu")
def _(rowptr, col, val, G, B, n_cols):
    raise RuntimeError(_NO_CPU.replace("spmm_csr", "sddmm_csr"))


@sddmm_csr_op.register_fake
def _(rowptr, col, val, G, B, n_cols):
    return val.new_empty(val.shape)


def _setup_context(ctx, inputs, output):
    rowptr, col, val, B, bias, n_cols, relu = inputs
    ctx.n_cols, ctx.relu = n_cols, bool(relu)
    ctx.bias_dtype = bias.dtype if bias is not None else None
    ctx.val_grad = bool(val.requires_grad)
    # (B is kept only when the adjacency values themselves need a gradient: grad_val is an SDDMM
    #  of grad_C and B; the reference's adjacency is a constant and never asks for it)
    ctx.save_for_backward(rowptr, col, val, *([output] if relu else []), *([B] if ctx.val_grad else []))


def _backward(ctx, grad_out):
    # written with operators only (no raw pointers), so AOT autograd can trace it
    rowptr, col, val = ctx.saved_tensors[:3]
    grad_B = grad_bias = grad_val = None
    if ctx.relu:
        grad_out = torch.where(ctx.saved_tensors[3] > 0, grad_out, torch.zeros_like(grad_out))
    if ctx.val_grad and ctx.needs_input_grad[2]:
        grad_val = torch.ops.pygcn_amd.sddmm_csr(rowptr, col, val, grad_out.contiguous(),
                                                 ctx.saved_tensors[-1], ctx.n_cols)
    if ctx.needs_input_grad[3]:
        rp_t, col_t, val_t = torch.ops.pygcn_amd.csr_transpose(rowptr, col, val, ctx.n_cols)
        grad_B = torch.ops.pygcn_amd.spmm_csr(rp_t, col_t, val_t, grad_out.contiguous(), None,
                                              rowptr.shape[0] - 1, False)
    if ctx.bias_dtype is not None and ctx.needs_input_grad[4]:
        grad_bias = grad_out.sum(0).to(ctx.bias_dtype)
    return None, None, grad_val, grad_B, grad_bias, None, None


spmm_csr_op.register_autograd(_backward, setup_context=_setup_context)


def sparse_mm(adj, dense, bias=None):
    """Drop-in for `torch.spmm(adj, dense)` / `torch.sparse.mm` (+ optional fused bias) with
    autograd, through the registered operator.  `adj`: CSRGraph or a torch sparse COO/CSR tensor
    on the HIP device (converted once, cached on the tensor)."""
    from .graph import as_graph
    g = adj if isinstance(adj, CSRGraph) else as_graph(adj)
    return torch.ops.pygcn_amd.spmm_csr(g.rowptr, g.col, g.val, dense, bias, g.shape[1], False)
