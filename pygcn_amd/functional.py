"""Loss head of the reference's training step, `F.nll_loss(output[idx_train], labels[idx_train])`
(reference pygcn/train.py:153, mean reduction), as a gather / scatter pair.

torch's `nll_loss` kernels walk one row per thread: on the [|idx_train|, 256] slice of config C4
(517 k rows) forward + backward take 2.0 ms per epoch; a gather of one element per row and a
scatter of one element per row into a zero tensor take 0.3 ms.  Same value, same gradient.
"""
import torch
import torch.utils._pytree as pytree


class NLLGrad(torch.Tensor):
    """The gradient of a mean NLL loss with respect to its [n, C] input: ONE non-zero per row,
    `coef` (device float32 [1]) at column target[r] (a negative target marks an ignored row: zeros).  A wrapper tensor without storage, like
    rowgrad.RowGrad: the model's one-node backward pass (pygcn_amd/fused.py) recognises it and
    forms coef·(onehot − exp(logp)) in one sweep without the [n, C] gradient ever existing
    (10 GB at config C4); every other consumer sees the dense tensor (any operator on it
    materialises it first)."""

    @staticmethod
    def __new__(cls, target, coef, shape, dtype):
        t = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=dtype, device=target.device,
                                                requires_grad=False)
        t.target, t.coef = target, coef
        return t

    def dense(self):
        n = self.shape[0]
        out = torch.zeros(tuple(self.shape), dtype=self.dtype, device=self.device)
        keep = self.target >= 0                                   # (negative label = ignored row)
        safe = torch.where(keep, self.target, torch.zeros_like(self.target))
        return out.scatter_(1, safe.view(-1, 1), (self.coef.to(self.dtype) * keep).view(n, 1))

    def __repr__(self):
        return f"NLLGrad(shape={tuple(self.shape)}, dtype={self.dtype})"

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        if func is torch.ops.aten.isnan.default and len(args) == 1 and isinstance(args[0], NLLGrad):
            # anomaly mode's NaN question (see rowgrad.RowGrad): the gradient's only non-zero values
            # are copies of `coef` — answered without materialising [n, C]
            return torch.isnan(args[0].coef)
        conv = lambda a: a.dense() if isinstance(a, NLLGrad) else a     # noqa: E731
        return func(*pytree.tree_map(conv, args), **pytree.tree_map(conv, kwargs or {}))


class _NLLMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, target, structural=False, ignore_index=-100):
        if log_probs.dim() != 2 or target.dim() != 1 or target.shape[0] != log_probs.shape[0]:
            raise RuntimeError("nll_loss: expected log_probs [n, C] and target [n]")
        ctx.structural = bool(structural)
        ctx.shape, ctx.dtype = log_probs.shape, log_probs.dtype
        # `ignore_index` exactly as torch: such rows add nothing and do not count in the mean
        # (elementwise work on [n] only; no host synchronisation)
        keep = target != ignore_index
        safe = torch.where(keep, target, torch.zeros_like(target))
        picked = log_probs.gather(1, safe.view(-1, 1)).float().squeeze(1)
        count = keep.sum().float()
        ctx.save_for_backward(safe, keep, count)
        return -(picked * keep).sum() / count

    @staticmethod
    def backward(ctx, grad):
        safe, keep, count = ctx.saved_tensors
        n = ctx.shape[0]
        if ctx.structural and grad.is_cuda and n > 0:   # nothing of size [n, C] is written here
            # (rows carrying the — negative — ignore_index keep their negative label: the consumer
            #  gives them a zero gradient row, and `count` excludes them)
            return NLLGrad(torch.where(keep, safe, torch.full_like(safe, -1)),
                           (-grad / count).float().reshape(1), ctx.shape, ctx.dtype), None, None, None
        g = torch.zeros(ctx.shape, dtype=ctx.dtype, device=grad.device)
        coef = ((-grad / count).float() * keep).to(ctx.dtype).view(n, 1)
        g.scatter_(1, safe.view(-1, 1), coef)
        return g, None, None, None


def nll_loss(log_probs, target, ignore_index=-100):
    """Drop-in for `torch.nn.functional.nll_loss(log_probs, target)` (mean over the rows, no class
    weights; `ignore_index` as in torch) — the form the reference's training step uses
    (pygcn/train.py:153).  fp32 or bf16 log-probabilities (the loss value is fp32 either way).

    When `log_probs` is the model's own full output (`model(x, adj)` in training mode: a
    rowgrad.RowSelectable), its gradient travels in STRUCTURAL form (NLLGrad: the label vector and
    one coefficient) — the only consumer is then the model's autograd node, which declared it can
    take it; any other tensor gets the ordinary dense gradient."""
    from pygcn_amd.rowgrad import RowSelectable
    structural = (isinstance(log_probs, RowSelectable) and log_probs.dim() == 2 and log_probs.is_cuda
                  and log_probs.requires_grad and target.dtype == torch.int64
                  and target.device == log_probs.device and ignore_index < 0)
    if isinstance(log_probs, torch.Tensor) and type(log_probs) is not torch.Tensor:
        log_probs = log_probs.as_subclass(torch.Tensor)
    return _NLLMean.apply(log_probs, target.contiguous(), structural, int(ignore_index))
