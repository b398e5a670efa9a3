"""Shared helper of the full-size tests: recompute sampled rows of a product A·B with the CPU
oracle from the rows' own stored entries (the whole product does not fit a CPU in test time)."""
import numpy as np
import torch


def sampled_rows_reference(oracle, g, B, rows, round_to=None):
    """oracle.spmm_csr on the sub-problem made of `rows` of CSRGraph `g` and the rows of B they
    reference.  `round_to`: optional torch dtype the operand is stored in (bf16): the reference then
    works on the rounded values in fp32."""
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
    return oracle.spmm_csr(rp.cpu().numpy(), inv.cpu().numpy().astype(np.int32), vals.cpu().numpy(),
                           Bs.float().cpu().numpy())


def heavy_and_random_rows(g, n_heavy, n_random, gen):
    """Row ids: the `n_heavy` longest rows of `g` (the chunked long-row path), `n_random` uniform
    ones, the first and the last."""
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    top = torch.topk(deg, n_heavy).indices
    rnd = torch.randint(0, g.shape[0], (n_random,), generator=gen, device=g.device)
    ends = torch.tensor([0, g.shape[0] - 1], device=g.device)
    return torch.unique(torch.cat([top, rnd, ends])), int(deg[top].max())
