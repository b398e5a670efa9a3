"""Shared helper of the full-size tests: recompute sampled rows of a product A·B with the CPU
oracle from the rows' own stored entries (the whole product does not fit a CPU in test time)."""
import numpy as np
import torch


def sampled_rows_reference(oracle, g, B, rows, round_to=None, f64=False):
    """oracle.spmm_csr on the sub-problem made of `rows` of CSRGraph `g` and the rows of B they
    reference.  `round_to`: optional torch dtype the operand is stored in (bf16): the reference then
    works on the rounded values in fp32.  `f64`: accumulate in float64 (oracle.spmm_csr_f64acc) —
    the arbiter for rows of 10⁴–10⁵ entries, where a single float32 chain (the CPU reference's own
    arithmetic) carries ~1e-5 of rounding error itself."""
    dev = g.device
    rows = rows.to(dev).long()
    starts, ends = g.rowptr[rows].long(), g.rowptr[rows + 1].long()
    lens = ends - starts
    total = int(lens.sum())
    idx = torch.repeat_interleave(starts - torch.cumsum(lens, 0) + lens, lens) + torch.arange(total, device=dev)
    cols, vals = g.col[idx].long(), g.val[idx]
    ucols, inv = torch.unique(cols, return_inverse=True)
    rp = torch.zeros(len(rows) + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=rp[1:])
    Bs = B[ucols]
    if round_to is not None:
        Bs = Bs.to(round_to)
    fn = oracle.spmm_csr_f64acc if f64 else oracle.spmm_csr
    return fn(rp.cpu().numpy(), inv.cpu().numpy().astype(np.int32), vals.cpu().numpy(),
              Bs.float().cpu().numpy())


def heavy_and_random_rows(g, n_heavy, n_random, gen):
    """Row ids: the `n_heavy` longest rows of `g` (the chunked long-row path), `n_random` uniform
    ones, the first and the last."""
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    top = torch.topk(deg, n_heavy).indices
    rnd = torch.randint(0, g.shape[0], (n_random,), generator=gen, device=g.device)
    ends = torch.tensor([0, g.shape[0] - 1], device=g.device)
    return torch.unique(torch.cat([top, rnd, ends])), int(deg[top].max())


def device_relu_mask(oracle, model, x, g, a, tol=1e-5):
    """The ReLU derivative the DEVICE used for the hidden layer: (h1 > 0) of the layer-level call
    (the same kernels as inside the model's one-node path; call with dropout 0).  The oracle's
    pre-activations differ from the device's by fp32 summation order (the device may evaluate the
    first layer as (Â·X)·W), so an element within rounding of zero can sit on the other side of the
    ReLU: invisible in the forward pass, but it switches one term of grad_W1 / grad_b1 on or off —
    over 10⁶ x 256 hidden units a few dozen such terms move a weight gradient by ~1e-3 of its
    norm.  The derivative of ReLU at 0 is a convention, not arithmetic: the comparison is made
    well-posed by handing the oracle the device's mask, AFTER asserting that the two masks differ
    only where the oracle's own pre-activation is within `tol` of zero (relative to the largest).
    Returns (mask bool [n, hidden] numpy, number of differing elements)."""
    was = model.training
    model.eval()
    with torch.no_grad():
        h1 = model.gc1(x, g, relu=True)
    model.train(was)
    mask = (h1 > 0).cpu().numpy()
    p = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    pre, _ = oracle.gc_forward(x.detach().float().cpu().numpy(), p["gc1.weight"], p.get("gc1.bias"), a)
    flips = mask != (pre > 0)
    if flips.any():
        assert np.abs(pre[flips]).max() <= tol * np.abs(pre).max(), "masks differ away from the ReLU boundary"
    assert flips.mean() < 1e-4
    return mask, int(flips.sum())
