"""Full-size (BASELINE config C4: 10^7 vertices / 10^8 sampled edges, F = 256) checks of the HIP
path through properties that do not need a full-size CPU product:

  * Â is row-normalized, so Â · 1 = 1;  Âᵀ · 1 = the column sums of Â (fp64 bincount);
  * linearity: Â·(αB1 + B2) = α·Â·B1 + Â·B2;
  * sampled rows — uniformly random ones plus the longest rows (which take the chunked long-row
    path) — recomputed by the CPU oracle from their own stored entries.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import assert_normwise, assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c4():
    assert torch.cuda.is_available()
    from pygcn_amd import CSRGraph
    from pygcn_amd.utils import rmat_graph
    dev = torch.device("cuda:0")
    n, e = 10_000_000, 100_000_000
    rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    yield g, n
    del g
    torch.cuda.empty_cache()


def test_row_and_column_sums(c4):
    from pygcn_amd import spmm_csr
    g, n = c4
    ones = torch.ones(n, 256, device=g.device)
    out = spmm_csr(g, ones)
    assert float((out - 1).abs().max()) <= 1e-5          # rows of D^-1(A+I) sum to 1
    del out
    colsum = torch.zeros(n, dtype=torch.float64, device=g.device).index_add_(
        0, g.col.long(), g.val.double())
    out_t = spmm_csr(g.t(), ones)
    err = (out_t[:, 0].double() - colsum).abs().max().item()
    assert err <= 1e-5 * colsum.max().item()
    assert torch.equal(out_t[:, 0], out_t[:, 255])       # every feature column sees the same sum


def test_linearity(c4):
    from pygcn_amd import spmm_csr
    g, n = c4
    gen = torch.Generator(device=g.device).manual_seed(7)
    b1 = torch.randn(n, 256, generator=gen, device=g.device)
    b2 = torch.randn(n, 256, generator=gen, device=g.device)
    lhs = spmm_csr(g, 0.75 * b1 + b2)
    rhs = spmm_csr(g, b1).mul_(0.75).add_(spmm_csr(g, b2))
    scale = float(rhs.abs().max())
    assert float((lhs - rhs).abs().max()) <= 1e-5 * scale


def test_sampled_rows_against_oracle(c4, oracle):
    from pygcn_amd import spmm_csr
    g, n = c4
    gen = torch.Generator(device=g.device).manual_seed(8)
    B = torch.randn(n, 256, generator=gen, device=g.device)
    out = spmm_csr(g, B)
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    top = torch.topk(deg, 40).indices                      # hubs: tens of chunks each
    rnd = torch.randint(0, n, (3000,), generator=gen, device=g.device)
    rows = torch.unique(torch.cat([top, rnd, torch.tensor([0, n - 1], device=g.device)]))
    assert int(deg[top].max()) > 20000
    # gather the sampled rows' entries and the B rows they reference; remap to a small problem
    starts, ends = g.rowptr[rows].long(), g.rowptr[rows + 1].long()
    lens = ends - starts
    idx = torch.repeat_interleave(starts - torch.cumsum(lens, 0) + lens, lens) + torch.arange(
        int(lens.sum()), device=g.device)
    cols, vals = g.col[idx].long(), g.val[idx]
    ucols, inv = torch.unique(cols, return_inverse=True)
    rp = torch.zeros(len(rows) + 1, dtype=torch.int64, device=g.device)
    torch.cumsum(lens, 0, out=rp[1:])
    ref = oracle.spmm_csr(rp.cpu().numpy(), inv.cpu().numpy().astype(np.int32),
                          vals.cpu().numpy(), B[ucols].cpu().numpy())
    assert_normwise(out[rows].cpu(), ref, 1e-5, "sampled rows incl. hubs")


def test_sampled_rows_of_the_transpose_product_against_oracle(c4, oracle):
    """Âᵀ·G with a random G at full size — the backward product of a dense-gradient epoch —
    against the oracle on >= 3 000 sampled rows of CSR(Âᵀ) incl. its 40 heaviest (the hub COLUMNS
    of Â: tens of thousands of entries each, the chunked long-row path)."""
    from pygcn_amd import spmm_csr
    from _sampling import heavy_and_random_rows, sampled_rows_reference
    g, n = c4
    gt = g.t()
    gen = torch.Generator(device=g.device).manual_seed(18)
    G = torch.randn(n, 256, generator=gen, device=g.device)
    out = spmm_csr(gt, G)
    rows, heaviest = heavy_and_random_rows(gt, 40, 3000, gen)
    assert heaviest > 20000
    # Rows of Âᵀ are not normalized: a hub column sums 10⁴–10⁵ terms to magnitudes ~10², and the
    # float32 CPU chain loses ~1e-5 there by itself.  Arbiter: float64 accumulation of the same
    # float32 products; the HIP result must be within the contract's 1e-5 of it, and within 1e-5 +
    # (the float32 oracle's own measured distance from float64) of the float32 oracle
    # (conftest.assert_parity).
    got = out[rows].cpu().numpy()
    ref64 = sampled_rows_reference(oracle, gt, G, rows, f64=True)
    ref32 = sampled_rows_reference(oracle, gt, G, rows)
    assert_parity(got, ref32, ref64, "C4 transpose product: sampled rows incl. the heaviest columns")


def test_transpose_block_equals_the_full_transpose_product(c4):
    """The [|R2|, |R|] block of Âᵀ the one-node backward pass multiplies (cut from the rows R of
    CSR(Â)) against the full cached CSR(Âᵀ) on an operand that is zero outside R: the same rows,
    and nothing outside R2."""
    from pygcn_amd import fused, spmm_csr
    g, n = c4
    dev = g.device
    rows = torch.arange(n * 140 // 2708, device=dev)                 # the bench's idx_train share
    rs = fused.row_sets(g, rows)
    assert rs.at_block.shape == (rs.n2, rs.n_u) and rs.n_u == rows.numel()
    gen = torch.Generator(device=dev).manual_seed(11)
    gp = torch.randn(rs.n_u, 256, generator=gen, device=dev)
    small = spmm_csr(rs.at_block, gp)
    operand = torch.zeros(n, 256, device=dev)
    operand[rows] = gp
    full = spmm_csr(g.t(), operand)
    del operand
    scale = float(full.abs().max())
    assert float((full.index_select(0, rs.rows2) - small).abs().max()) <= 1e-5 * scale
    outside = torch.ones(n, dtype=torch.bool, device=dev)
    outside[rs.rows2] = False
    assert float(full[outside].abs().max()) == 0.0                   # R2 is exactly where it can be non-zero


def test_training_step_routes_agree(c4):
    """One training step of the 2-layer model at C4 by two routes through the HIP kernels:
    `model(x, adj, rows=idx)` (one autograd node: transpose block, gather-fused weight gradients,
    masks in the GEMM stores) and upstream's unchanged lines `model(x, adj)[idx]` (one node per
    layer: dense gradient through autograd, row bitmaps found at run time, row-restricted
    launches).  Same forward kernels, different backward routes: loss, selected rows and all four
    parameter gradients must agree."""
    from pygcn_amd import GCN
    g, n = c4
    dev = g.device
    gen = torch.Generator(device=dev).manual_seed(44)
    x = torch.randn(n, 256, generator=gen, device=dev)
    labels = torch.randint(0, 256, (n,), generator=gen, device=dev)
    idx = torch.arange(n * 140 // 2708, device=dev)
    torch.manual_seed(42)
    model = GCN(256, 256, 256, dropout=0.0).to(dev)
    model.train()
    out_rows = model(x, g, rows=idx)
    loss = torch.nn.functional.nll_loss(out_rows, labels[idx])
    loss.backward()
    fused_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    sel = out_rows.detach().clone()
    model.zero_grad(set_to_none=True)
    del out_rows, loss
    full = model(x, g)
    loss2 = torch.nn.functional.nll_loss(full[idx], labels[idx])
    loss2.backward()
    assert torch.equal(full.detach()[idx], sel)                      # (the same forward kernels)
    del full
    for k, p in model.named_parameters():
        a, b = fused_grads[k].double(), p.grad.double()
        assert torch.isfinite(a).all() and float(b.abs().max()) > 0
        err = float((a - b).abs().max())
        # (two float32 routes, each within the contract's 1e-5 of exact arithmetic)
        assert err <= 2e-5 * float(b.abs().max()), f"{k}: {err:.3e} vs {float(b.abs().max()):.3e}"
