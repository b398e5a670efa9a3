"""Row-sparse gradients between autograd nodes, so that upstream's UNCHANGED lines

    output = model(features, adj)                                   (pygcn/train.py:150)
    loss_train = F.nll_loss(output[idx_train], labels[idx_train])   (pygcn/train.py:153)
    loss_train.backward()                                           (pygcn/train.py:157)

run their backward pass on the rows that can be non-zero — what `model(x, adj, rows=idx)` does
inside one node (pygcn_amd/fused.py) — without the model being told the rows.

* In training mode `GCN.forward` returns its log-probabilities as a `RowSelectable`: an ordinary
  tensor in every respect except that `output[idx]` with a 1-D int64 index tensor goes through
  `SelectRowsFunction`, whose backward hands autograd a `RowGrad` instead of a dense [N, C] tensor
  of zeros with |idx| rows scattered into it.
* `RowGrad` is a wrapper subclass (shape [N, C], no storage) carrying (rows, values).  The
  layer's autograd node (`GraphConvFunction.backward`, pygcn_amd/spmm.py) recognises it and takes the
  structural route: the cached row sets and transpose block of (graph, rows), compact operands,
  and — for the layer below — another `RowGrad` over the rows R2.  Anything else that meets a
  `RowGrad` (a second consumer of `output`, a hook, an optimizer) sees the dense tensor: every
  operator other than the ones above materialises it first (`__torch_dispatch__`), so results never
  depend on who consumes the gradient.
"""
import torch
import torch.utils._pytree as pytree


class RowGrad(torch.Tensor):
    """A gradient of shape [n, C] that is zero outside `rows` (int64 [k], any order, duplicates
    add up): `values` [k, C] holds its rows.  `meta`: optional structure the producer already has
    for these rows (e.g. the padded int32 list of a sorted-unique row set)."""

    @staticmethod
    def __new__(cls, rows, values, n, meta=None):
        t = torch.Tensor._make_wrapper_subclass(cls, (n, values.shape[1]), dtype=values.dtype,
                                                device=values.device, requires_grad=False)
        t.rows, t.values, t.meta = rows, values, meta
        return t

    def dense(self):
        out = torch.zeros(tuple(self.shape), dtype=self.dtype, device=self.device)
        return out.index_add_(0, self.rows, self.values) if self.rows.numel() else out

    def __repr__(self):
        return f"RowGrad(shape={tuple(self.shape)}, rows={self.rows.numel()}, dtype={self.dtype})"

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        if func is torch.ops.aten.isnan.default and len(args) == 1 and isinstance(args[0], RowGrad):
            # anomaly mode (`torch.autograd.set_detect_anomaly(True)` around `backward()`, reference
            # pygcn/policy-generator.py:419-420) asks every gradient a node returns whether it holds a
            # NaN: answered on the compact rows — the structurally-zero rows hold none — so the
            # [n, C] tensor is not materialised for the question
            return torch.isnan(args[0].values)
        conv = lambda a: a.dense() if isinstance(a, RowGrad) else a     # noqa: E731
        return func(*pytree.tree_map(conv, args), **pytree.tree_map(conv, kwargs or {}))


class SelectRowsFunction(torch.autograd.Function):
    """full[idx] whose gradient is a RowGrad."""

    @staticmethod
    def forward(ctx, full, idx):
        ctx.n = full.shape[0]
        ctx.idx = idx                       # (the caller's tensor object: row-set caches key on it)
        return full.index_select(0, idx)

    @staticmethod
    def backward(ctx, grad_rows):
        return RowGrad(ctx.idx, grad_rows, ctx.n), None


class LossRows(torch.Tensor):
    """`output[idx_train]` of the model's RowSelectable output: an ordinary tensor, except that
    upstream's very next call — `F.nll_loss(output[idx_train], labels[idx_train])`
    (pygcn/train.py:153) with its default arguments — runs as the gather / scatter pair of
    pygcn_amd.functional.nll_loss instead of torch's one-thread-per-row kernels (2 ms of a 50 ms
    epoch at C4, 10 ms of 106 at C5).  Same value, same gradient, `ignore_index` honoured; any other
    use of the tensor is the plain one."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if (func is torch.nn.functional.nll_loss and len(args) >= 2 and isinstance(args[0], LossRows)
                and len(args) == 2 and args[0].dim() == 2 and isinstance(args[1], torch.Tensor)
                and args[1].dim() == 1 and args[1].dtype == torch.int64
                and kwargs.get("weight") is None and kwargs.get("size_average") is None
                and kwargs.get("reduce") is None and kwargs.get("reduction", "mean") == "mean"
                and set(kwargs) <= {"weight", "size_average", "reduce", "reduction", "ignore_index"}):
            from pygcn_amd.functional import nll_loss
            return nll_loss(args[0].as_subclass(torch.Tensor), args[1], kwargs.get("ignore_index", -100))
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return pytree.tree_map(lambda o: o.as_subclass(torch.Tensor) if isinstance(o, LossRows) else o, out)


_NONNEG = {}     # rows_key of an index tensor -> are all its entries >= 0 (one host read per tensor)


def _known_nonnegative(idx):
    """`output[idx]` wraps negative indices, `index_select` (the structural route) does not — and on
    ROCm its range assertion is compiled out.  Whether an index tensor is free of
    negative entries is read ONCE per tensor identity (the row-set cache of pygcn_amd/fused.py
    reads the same tensor on the host at that point anyway); a tensor with negative entries takes
    the ordinary indexing path."""
    from pygcn_amd.fused import rows_key
    import weakref
    key = rows_key(idx)
    ent = _NONNEG.get(key)
    if ent is None or ent[1]() is not idx:       # (an address reused by another tensor is not a hit)
        if len(_NONNEG) >= 16:
            _NONNEG.clear()
        ent = _NONNEG[key] = (bool((idx >= 0).all()) if idx.numel() else True, weakref.ref(idx))
    return ent[0]


class RowSelectable(torch.Tensor):
    """Marker subclass of the model's output (see the module docstring)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if (func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[0], RowSelectable)
                and isinstance(args[1], torch.Tensor) and not isinstance(args[1], RowSelectable)
                and args[1].dim() == 1 and args[1].dtype == torch.int64 and args[0].dim() == 2
                and args[0].requires_grad and torch.is_grad_enabled()
                and args[1].device == args[0].device and _known_nonnegative(args[1])):
            return SelectRowsFunction.apply(args[0].as_subclass(torch.Tensor), args[1]).as_subclass(LossRows)
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        # results are ordinary tensors: only the model's own output is selectable
        return pytree.tree_map(lambda o: o.as_subclass(torch.Tensor) if isinstance(o, RowSelectable) else o, out)
