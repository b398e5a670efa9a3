"""Row f2: the hand-written MFMA GEMMs Y = X[M,256] · W[256,256] (three bf16 parts / two scaled fp16
parts) against an fp64 product, at the edges of the fp32 range, and through the layer's autograd
against the oracle."""
import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_normwise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(scheme, X, W, bound=None):
    """Y through one of the two kernels; for "h2" the bound defaults to the exact max|X|."""
    from pygcn_amd import spmm as S
    from pygcn_amd.spmm import gemm_xw256
    if scheme == "bf16x3":
        S.set_gemm_scheme("bf16x3")
        try:
            return gemm_xw256(X, W)
        finally:
            S.set_gemm_scheme("h2")
    finite = torch.where(torch.isfinite(X), X.abs(), torch.zeros_like(X))
    b = (finite.max() if bound is None else torch.as_tensor(bound, device=X.device)).float().reshape(1)
    return gemm_xw256(X, W, x_bound=b)


@pytest.mark.parametrize("scheme", ["bf16x3", "h2"])
@pytest.mark.parametrize("M", [1, 31, 32, 255, 257, 1000, 65537])
def test_gemm_xw256_matches_fp64(dev, M, scheme):
    gen = torch.Generator(device=dev).manual_seed(M)
    X = torch.randn(M, 256, generator=gen, device=dev) * (10 ** (4 * torch.rand(M, 1, generator=gen,
                                                                                device=dev) - 2))
    W = torch.randn(256, 256, generator=gen, device=dev)
    Y = _run(scheme, X, W)
    assert Y is not None and Y.shape == (M, 256)
    ref = X.double() @ W.double()
    # per-row check (rows span 4 orders of magnitude): fp32-level accuracy relative to the ROW's
    # scale — 2e-6 for three bf16 parts (24 bits), 4e-6 for two fp16 parts (22 bits)
    err = (Y.double() - ref).abs().amax(1)
    scale = ref.abs().amax(1)
    tol = 2e-6 if scheme == "bf16x3" else 4e-6
    assert bool((err <= tol * scale).all()), float((err / scale).max())
    torch_err = ((X @ W).double() - ref).abs().amax(1)
    assert float(err.max() / scale.max()) <= 4 * float(torch_err.max() / scale.max()) + 1e-7
    # strided rows (a column slice of a wider buffer) and special values
    big = torch.randn(M, 512, generator=gen, device=dev)
    Ys = _run(scheme, big[:, 256:], W)
    assert_normwise(Ys.cpu(), (big[:, 256:].double() @ W.double()).cpu().numpy(), tol, "strided")
    Xz = torch.zeros(M, 256, device=dev)
    Xz[0, 3] = float("inf")
    out = _run(scheme, Xz, W, bound=1.0)
    assert bool(torch.isinf(out[0]).any() or torch.isnan(out[0]).any())
    assert bool((out[1:] == 0).all())


@pytest.mark.parametrize("scheme", ["bf16x3", "h2"])
def test_gemm_precision_at_the_edges_of_fp32(dev, scheme):
    """the decompositions are sold as fp32 — check them where a reduced-range
    format would break, each case against an fp64 product AND against torch.mm (hipBLASLt fp32)."""
    gen = torch.Generator(device=dev).manual_seed(9)
    M = 512
    base = torch.randn(M, 256, generator=gen, device=dev)
    W = torch.randn(256, 256, generator=gen, device=dev) * 0.1

    def check(X, Wm, what, tol=4e-6):
        Y = _run(scheme, X, Wm)
        ref = X.double() @ Wm.double()
        scale = float(ref.abs().max())
        assert scale > 0 and bool(torch.isfinite(Y).all()), what
        err = float((Y.double() - ref).abs().max()) / scale
        t_err = float(((X @ Wm).double() - ref).abs().max()) / scale
        assert err <= tol, f"{what}: {err:.3e} (torch.mm: {t_err:.3e})"
        return err, t_err
    check(base, W, "N(0,1)")
    # whole operand near the top of the fp32 range: products reach 1e38 without overflowing
    check(base * 1e37 / 16, W * 0.01, "|x| ~ 1e36..1e37")
    # whole operand deep below the bf16 / fp16 normal range (fp32 normal numbers)
    check(base * 1e-30, W, "|x| ~ 1e-30")
    if scheme == "h2":
        check(base * 1e-37, W * 1e3, "|x| ~ 1e-37 (next to the fp32 denormals)")
        # fp32 denormal inputs (|x| < 1.18e-38): exact scaling keeps their leading bits
        den = base * 1e-39
        assert bool((den.abs() < 1.2e-38).all()) and bool((den != 0).any())
        Y = _run(scheme, den, W * 1e6)
        ref = den.double() @ (W * 1e6).double()
        assert float((Y.double() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())   # 2^-10: few bits exist
    else:
        # the unscaled three-part scheme's envelope ends here: below ~1e-30 the second and third
        # bf16 parts fall into the bf16 denormals (measured 2.5e-4 at 1e-37) — which is why the
        # product path uses the scaled scheme; the error is still bounded by the first part's 2^-8
        Y = _run(scheme, base * 1e-37, W * 1e3)
        ref = (base * 1e-37).double() @ (W * 1e3).double()
        assert float((Y.double() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())
    # heavy cancellation: every row is (v, -v) against a W with duplicated rows -> the true result is
    # the rounding residue; the error must stay relative to the SUMMANDS' size, like torch.mm's
    v = torch.randn(M, 128, generator=gen, device=dev) * 100
    Xc = torch.cat([v, -v], 1)
    Wc = torch.cat([W[:128], W[:128] + 1e-3 * torch.randn(128, 256, generator=gen, device=dev)], 0)
    Yc = _run(scheme, Xc, Wc)
    refc = Xc.double() @ Wc.double()
    summands = float((Xc.abs().double() @ Wc.abs().double()).max())
    assert float((Yc.double() - refc).abs().max()) <= 3e-7 * summands
    assert float(((Xc @ Wc).double() - refc).abs().max()) <= 3e-7 * summands       # (same yardstick)
    if scheme == "h2":
        # a LOOSE bound (2^10 above the maximum) costs dynamic range, not correctness
        Yl = _run(scheme, base, W, bound=float(base.abs().max()) * 1024)
        assert_normwise(Yl.cpu(), (base.double() @ W.double()).cpu().numpy(), 4e-6, "loose bound")
        # documented envelope: a row 2^20 below the tensor maximum keeps an ABSOLUTE error bound
        mixed = base.clone()
        mixed[1::2] *= 2.0 ** -20
        Ym = _run(scheme, mixed, W)
        refm = mixed.double() @ W.double()
        assert float((Ym.double() - refm).abs().max()) <= 4e-6 * float(refm.abs().max())
        small_rows = (Ym[1::2].double() - refm[1::2]).abs().amax(1) / refm[1::2].abs().amax(1)
        assert float(small_rows.max()) <= 2e-3          # ~11 bits left on those rows (documented)


def test_gemm_output_maximum_side_channel(dev):
    """gcn_gemm_xw256_f32_h2 reports max|Y| (the next layer's bound) without a pass over Y; and
    without a caller-supplied bound the wrapper computes max|X| itself."""
    from pygcn_amd.spmm import gemm_xw256
    X = torch.randn(3000, 256, device=dev) * 3
    W = torch.randn(256, 256, device=dev)
    ymax = torch.zeros(1, device=dev)
    Y = gemm_xw256(X, W, x_bound=X.abs().max().reshape(1), y_absmax=ymax)
    assert float(ymax) == float(Y.abs().max())
    assert torch.equal(gemm_xw256(X, W), Y)


def test_gemm_declines_other_shapes(dev):
    from pygcn_amd.spmm import gemm_xw256
    assert gemm_xw256(torch.randn(10, 128, device=dev), torch.randn(128, 256, device=dev)) is None
    assert gemm_xw256(torch.randn(10, 256, device=dev).bfloat16(),
                      torch.randn(256, 256, device=dev).bfloat16()) is None
    odd = torch.randn(10, 257, device=dev)[:, 1:]            # rows start 4 bytes off a 16-byte boundary
    assert gemm_xw256(odd, torch.randn(256, 256, device=dev)) is None


def test_layer_256_to_256_through_custom_gemm(oracle, dev):
    """The C3/C4 layer shape: forward and backward of GraphConvolution(256, 256) run the custom
    GEMM (forward and grad_input) and must still match the oracle at the north-star tolerance."""
    from pygcn_amd import CSRGraph, GraphConvolution
    from pygcn_amd.utils import rmat_graph
    n = 20000
    rowptr, col, val = rmat_graph(n, 200000, seed=8, device="cpu")
    a = oracle.CSR(rowptr.numpy(), col.numpy(), val.numpy(), (n, n))
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    torch.manual_seed(1)
    layer = GraphConvolution(256, 256).to(dev)
    x = gin.dense((n, 256), 70)
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    y = layer(xg, g)
    go = gin.dense((n, 256), 71)
    y.backward(torch.from_numpy(go).to(dev))
    w, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    y_ref, _ = oracle.gc_forward(x, w, b, a)
    gx, gw, gb, _ = oracle.gc_backward(x, w, True, a, go)
    assert_normwise(y.detach().cpu(), y_ref, 1e-5, "y")
    assert_normwise(xg.grad.cpu(), gx, 1e-5, "grad_x")
    assert_normwise(layer.weight.grad.cpu(), gw, 2e-5, "grad_w")
    assert_normwise(layer.bias.grad.cpu(), gb, 2e-5, "grad_b")
