"""Row f2: the hand-written MFMA GEMMs Y = X[M,256] · W[256,256] (three bf16 parts / two scaled fp16
parts) against an fp64 product, at the edges of the fp32 range, and through the layer's autograd
against the oracle."""
import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_normwise, assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(scheme, X, W, bound=None):
    """Y through one of the two kernels; for "h2" the bound defaults to the exact max|X|."""
    from pygcn_amd import spmm as S
    from pygcn_amd.spmm import gemm_xw256
    before = S.gemm_scheme()
    S.set_gemm_scheme(scheme)
    try:
        if scheme == "bf16x3":
            return gemm_xw256(X, W)
        finite = torch.where(torch.isfinite(X), X.abs(), torch.zeros_like(X))
        b = (finite.max() if bound is None else torch.as_tensor(bound, device=X.device)).float().reshape(1)
        return gemm_xw256(X, W, x_bound=b)
    finally:
        S.set_gemm_scheme(before)


@pytest.mark.parametrize("M", [1, 257, 4099, 600_001])
def test_three_part_pipeline_equals_the_round_one_kernel_bitwise(dev, M):
    """gcn_gemm_xw256_f32_b3 through a ROW LIST (gemm_xw256_h2_kernel<., 1>: the fp32-equivalent scheme
    in round 3's DMA / persistent pipeline, 32x32x16 MFMAs) issues the six MFMAs of a product in the
    order of round 1's gcn_gemm_xw256_f32: same bits, at heights that give a persistent workgroup one
    ragged tile, several tiles, and a cross-tile prefetch.  Contiguous rows take gemm_xw256_s16_kernel
    (16x16x32 MFMAs, K chunks of 32: another fp32 summation order, stores under the next tile's MFMAs):
    held against the listed result to fp32 rounding of the ROW's scale, and bit-for-bit repeatable."""
    from pygcn_amd import _native, spmm as S
    L = _native.lib()
    g = torch.Generator(device=dev).manual_seed(M)
    X = torch.randn(M, 256, generator=g, device=dev) * (10 ** (4 * torch.rand(M, 1, generator=g, device=dev) - 2))
    W = torch.randn(256, 256, generator=g, device=dev)
    old = torch.empty(M, 256, device=dev)
    ws = torch.empty(L.gcn_gemm_xw256_workspace_bytes(), dtype=torch.uint8, device=dev)
    _native.check(L.gcn_gemm_xw256_f32(X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), old.data_ptr(),
                                       old.stride(0), M, ws.data_ptr(), ws.numel(),
                                       torch.cuda.current_stream().cuda_stream), "gcn_gemm_xw256_f32")
    before = S.gemm_scheme()
    S.set_gemm_scheme("bf16x3")
    try:
        listed = S.gemm_xw256(X, W, rows=torch.arange(M, device=dev, dtype=torch.int32))
    finally:
        S.set_gemm_scheme(before)
    assert torch.equal(listed, old)
    new = _run("bf16x3", X, W)
    err = (new.double() - old.double()).abs().amax(1)
    scale = (X.double() @ W.double()).abs().amax(1)
    assert bool((err <= 1e-6 * scale).all()), float((err / scale).max())
    assert torch.equal(new, _run("bf16x3", X, W))           # (no atomics, fixed order: the same bits every run)


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


def test_gemm_output_maximum_side_channel(dev, gemm_scheme):
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


def test_layer_256_to_256_through_custom_gemm(oracle, dev, gemm_scheme):
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
    _, gw64, gb64 = oracle.gc_backward_f64(x, w, True, a, go, need_grad_x=False)
    assert_normwise(y.detach().cpu(), y_ref, 1e-5, "y")
    assert_normwise(xg.grad.cpu(), gx, 1e-5, "grad_x")
    assert_parity(layer.weight.grad.cpu(), gw, gw64, "grad_w")      # (20 000-term float32 reductions)
    assert_parity(layer.bias.grad.cpu(), gb, gb64, "grad_b")


@pytest.mark.parametrize("M", [1, 15, 16, 17, 100, 4097, 70001])
def test_weight_gradient_over_row_lists_matches_fp64(dev, M, gemm_scheme):
    """gcn_gemm_atg256_f32: Σ_r A[ra[r]]ᵀ ⊗ G[rg[r]] with and without row lists (the weight
    gradient of pygcn/layers.py:33 over the rows on which the gradient can be non-zero)."""
    from pygcn_amd.spmm import weight_grad_rows
    gen = torch.Generator(device=dev).manual_seed(M)
    n_a, n_g = 3 * M + 5, 2 * M + 3
    A = torch.randn(n_a, 256, generator=gen, device=dev) * 3
    G = torch.randn(n_g, 256, generator=gen, device=dev) * 0.01
    ra = torch.randint(0, n_a, (M,), generator=gen, device=dev).to(torch.int32)
    rg = torch.randperm(n_g, generator=gen, device=dev)[:M].to(torch.int32)
    got = weight_grad_rows(A, G, ra, rg)
    ref = A[ra.long()].double().t() @ G[rg.long()].double()
    summands = float((A[ra.long()].abs().double().t() @ G[rg.long()].abs().double()).max())
    assert got.shape == (256, 256)
    assert float((got.double() - ref).abs().max()) <= 3e-7 * summands
    torch_err = float(((A[ra.long()].t() @ G[rg.long()]).double() - ref).abs().max())
    assert float((got.double() - ref).abs().max()) <= 4 * torch_err + 1e-7 * summands
    # one list only / no list, strided operands, and bitwise determinism
    Ac = A[ra.long()].contiguous()
    assert torch.equal(weight_grad_rows(Ac, G, None, rg), got)
    wide = torch.randn(M, 512, generator=gen, device=dev)
    g2 = weight_grad_rows(Ac, wide[:, 256:])
    ref2 = Ac.double().t() @ wide[:, 256:].double()
    assert float((g2.double() - ref2).abs().max()) <= 3e-7 * float((Ac.abs().double().t() @ wide[:, 256:].abs().double()).max())
    assert torch.equal(weight_grad_rows(Ac, wide[:, 256:]), g2)
    # rows outside the list must not be read: poison them
    Ap = torch.full_like(A, float("nan"))
    Ap[ra.long()] = A[ra.long()]
    assert torch.equal(weight_grad_rows(Ap, G, ra, rg), got)


@pytest.mark.parametrize("M", [1, 17, 100, 5000, 400_003])
def test_weight_gradient_with_the_bias_gradient_as_a_side_result(dev, M):
    """gcn_gemm_atg256_f32_b3_colsum: the weight gradient unchanged (same bits) and Σ_r G[rg[r]] — the layer's
    bias gradient (pygcn/layers.py:36) — from the rows the kernel loads anyway: against a float64 sum, with and
    without row lists, poisoned unlisted rows, lists that end inside a 32-row super-step and a 16-entry pad."""
    from pygcn_amd import spmm as S
    gen = torch.Generator(device=dev).manual_seed(M + 7)
    n_g = 2 * M + 3
    A = torch.randn(M, 256, generator=gen, device=dev)
    G = torch.randn(n_g, 256, generator=gen, device=dev) * 0.01 + 0.003          # (sums that do not cancel to zero)
    rg = torch.randperm(n_g, generator=gen, device=dev)[:M].to(torch.int32)
    before = S.gemm_scheme()
    S.set_gemm_scheme("bf16x3")
    try:
        plain = S.weight_grad_rows(A, G, None, rg)
        both = S.weight_grad_rows(A, G, None, rg, colsum_g=True)
        assert both is not None and torch.equal(both[0], plain)
        ref = G[rg.long()].double().sum(0)
        scale = float(G[rg.long()].double().abs().sum(0).max())
        assert both[1].shape == (256,) and both[1].dtype == torch.float32
        assert float((both[1].double() - ref).abs().max()) <= 2e-7 * scale, float((both[1].double() - ref).abs().max()) / scale
        Gp = torch.full_like(G, float("nan"))
        Gp[rg.long()] = G[rg.long()]
        again = S.weight_grad_rows(A, Gp, None, rg, colsum_g=True)
        assert torch.equal(again[0], plain) and torch.equal(again[1], both[1])
        Gc = G[rg.long()].contiguous()                                           # no list at all
        dense = S.weight_grad_rows(A, Gc, colsum_g=True)
        assert torch.equal(dense[0], plain) and float((dense[1].double() - ref).abs().max()) <= 2e-7 * scale
        S.set_gemm_scheme("h2")                                                  # (the two-part scheme has no such form)
        assert S.weight_grad_rows(A, G, None, rg, colsum_g=True) is None
    finally:
        S.set_gemm_scheme(before)


def test_weight_gradient_degenerate_lists(dev):
    from pygcn_amd.spmm import weight_grad_rows
    A, G = torch.randn(10, 256, device=dev), torch.randn(10, 256, device=dev)
    empty = torch.empty(0, dtype=torch.int32, device=dev)
    assert bool((weight_grad_rows(A, G, empty, empty) == 0).all())
    with pytest.raises(RuntimeError, match="different numbers of rows"):
        weight_grad_rows(A, G, empty, None)
    assert weight_grad_rows(A[:, :128], G) is None


def test_gemm_with_row_list(dev, gemm_scheme):
    """gcn_gemm_xw256_f32_h2 with x_rows: output row r = X[rows[r]] · W, unlisted rows never read."""
    from pygcn_amd.spmm import gemm_xw256
    X = torch.randn(5000, 256, device=dev)
    W = torch.randn(256, 256, device=dev)
    rows = torch.randperm(5000, device=dev)[:1237].to(torch.int32)
    # (the three-part scheme runs listed and contiguous rows on two kernels with different fp32
    #  summation orders: the bit-exact reference is the gathered matrix through an identity list)
    Xg = X[rows.long()].contiguous()
    ident = torch.arange(1237, device=dev, dtype=torch.int32)
    want = gemm_xw256(Xg, W, x_bound=X.abs().max().reshape(1), rows=ident)
    Xp = torch.full_like(X, float("nan"))
    Xp[rows.long()] = X[rows.long()]
    got = gemm_xw256(Xp, W, x_bound=X.abs().max().reshape(1), rows=rows)
    assert got.shape == (1237, 256) and torch.equal(got, want)
    dense = gemm_xw256(Xg, W, x_bound=X.abs().max().reshape(1))
    assert_normwise(got.cpu(), dense.double().cpu().numpy(), 1e-6, "listed vs contiguous rows")


@pytest.mark.parametrize("K,N", [(128, 128), (128, 256), (256, 128)])
@pytest.mark.parametrize("M", [1, 33, 1000, 65537])
def test_bf16_gemm_matches_fp32_on_rounded_inputs(dev, K, N, M):
    """gcn_gemm_xw_bf16 (config C5: 128 -> 128): bf16 products are exact in fp32, so against an fp64
    product of the same bf16 inputs only the fp32 accumulation and the final rounding to bf16
    (2^-9 relative) differ."""
    from pygcn_amd.spmm import gemm_bf16
    gen = torch.Generator(device=dev).manual_seed(K + N + M)
    X = torch.randn(M, K, generator=gen, device=dev).bfloat16()
    W = (torch.randn(K, N, generator=gen, device=dev) * 0.2).bfloat16()
    Y = gemm_bf16(X, W)
    assert Y is not None and Y.dtype == torch.bfloat16 and Y.shape == (M, N)
    ref = X.double() @ W.double()
    err = (Y.double() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-5 * float(ref.abs().max())).all())
    assert torch.equal(Y, gemm_bf16(X, W))
    # the same numbers as torch.mm at bf16 up to the final rounding
    assert float((Y.float() - (X @ W).float()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    wide = torch.randn(M, 2 * K, generator=gen, device=dev).bfloat16()
    Ys = gemm_bf16(wide[:, K:], W)                       # strided rows
    refs = wide[:, K:].double() @ W.double()
    assert bool(((Ys.double() - refs).abs() <= 2.0 ** -8 * refs.abs() + 1e-5 * float(refs.abs().max())).all())
    assert gemm_bf16(X[:, :64], W[:64]) is None and gemm_bf16(X.float(), W.float()) is None


def test_gemm_with_fused_relu_dropout_mask(dev, gemm_scheme):
    """mask_src: y = mask_src[input row] > 0 ? y * scale : 0 in the GEMM's own store (the backward
    of the fused ReLU / dropout epilogue on the grad_input GEMM), with and without a row list."""
    from pygcn_amd.spmm import gemm_xw256
    gen = torch.Generator(device=dev).manual_seed(4)
    X = torch.randn(4000, 256, generator=gen, device=dev)
    H = torch.randn(4000, 256, generator=gen, device=dev).clamp_min(0) * (torch.rand(4000, 256, generator=gen, device=dev) > 0.5)
    W = torch.randn(256, 256, generator=gen, device=dev)
    b = X.abs().max().reshape(1)
    plain = gemm_xw256(X, W, x_bound=b)
    ymax = torch.zeros(1, device=dev)
    got = gemm_xw256(X, W, x_bound=b, y_absmax=ymax, mask_src=H, mask_scale=2.0)
    want = torch.where(H > 0, plain * 2.0, torch.zeros_like(plain))
    assert torch.equal(got, want) and float(ymax) == float(want.abs().max())
    rows = torch.randperm(4000, device=dev)[:999].to(torch.int32)
    got_r = gemm_xw256(X, W, x_bound=b, rows=rows, mask_src=H, mask_scale=2.0)
    plain_r = gemm_xw256(X, W, x_bound=b, rows=rows)            # (the listed rows' own kernel, see test_gemm_with_row_list)
    assert torch.equal(got_r, torch.where(H[rows.long()] > 0, plain_r * 2.0, torch.zeros_like(plain_r)))
    assert_normwise(got_r.cpu(), want[rows.long()].double().cpu().numpy(), 1e-6, "listed vs contiguous rows")


@pytest.mark.parametrize("M", [1, 129, 4099, 300_007])
@pytest.mark.parametrize("p", [0.0, 0.5])
def test_keep_bits_carry_the_backward_mask(dev, M, p):
    """gcn_gemm_epilogue.keep_bits_out / mask_bits (ABI 25): the forward launch writes `out > 0` as one bit per
    element, the grad_input launch masks from those bits — the SAME bits as masking from the fp32 activations,
    with and without a mask row list; launches that cannot take the contiguous-row kernel read mask_src."""
    from pygcn_amd import spmm as S
    from pygcn_amd.spmm import gemm_xw256
    gen = torch.Generator(device=dev).manual_seed(M + int(10 * p))
    X = torch.randn(M, 256, generator=gen, device=dev)
    W = torch.randn(256, 256, generator=gen, device=dev) * 0.1
    G = torch.randn(M, 256, generator=gen, device=dev)
    bias = torch.randn(256, generator=gen, device=dev) * 0.5
    before = S.gemm_scheme()
    S.set_gemm_scheme("bf16x3")
    try:
        assert S.gemm_keep_bits_usable(X, None, p)
        bits = torch.full((M, 8), -1, dtype=torch.int32, device=dev)
        H = gemm_xw256(X, W, bias=bias, relu=True, dropout_p=p, seed=77, keep_bits_out=bits)
        assert torch.equal(H, gemm_xw256(X, W, bias=bias, relu=True, dropout_p=p, seed=77))    # Y itself unchanged
        assert int((H > 0).sum()) == int(sum(((bits >> k) & 1).sum() for k in range(32)))     # as many bits as positives
        want = gemm_xw256(G, W, mask_src=H, mask_scale=1.5)
        got = gemm_xw256(G, W, mask_src=H, mask_scale=1.5, mask_bits=bits)
        assert torch.equal(got, want)
        poisoned = torch.full_like(H, float("nan"))                      # with bits, the activations are not read
        assert torch.equal(gemm_xw256(G, W, mask_src=poisoned, mask_scale=1.5, mask_bits=bits), want)
        perm = torch.randperm(M, device=dev).to(torch.int32)
        want_p = gemm_xw256(G, W, mask_src=H, mask_rows=perm, mask_scale=1.5)
        assert torch.equal(gemm_xw256(G, W, mask_src=poisoned, mask_rows=perm, mask_scale=1.5, mask_bits=bits), want_p)
        rows = torch.arange(M, device=dev, dtype=torch.int32)            # a row list: the other kernel, mask_src is read
        assert torch.equal(gemm_xw256(G, W, rows=rows, mask_src=H, mask_scale=1.5, mask_bits=bits),
                           gemm_xw256(G, W, rows=rows, mask_src=H, mask_scale=1.5))
        with pytest.raises(RuntimeError):
            gemm_xw256(X, W, bias=bias, relu=True, dropout_p=0.3, seed=77, keep_bits_out=bits)
    finally:
        S.set_gemm_scheme(before)


@pytest.mark.parametrize("p", [0.0, 0.3, 0.5])
def test_gemm_forward_epilogue_matches_the_spmm_epilogue(dev, p, gemm_scheme):
    """bias + ReLU + inverted dropout in the GEMM's store (a layer evaluated as (Â·X)·W + b): the
    same values — and the SAME Philox keep bits for a given (seed, row, column) — as the SpMM
    epilogue applied to the plain product (checked through an identity adjacency)."""
    from pygcn_amd import CSRGraph, spmm_csr
    from pygcn_amd.spmm import gemm_xw256
    gen = torch.Generator(device=dev).manual_seed(12)
    M = 3001
    X = torch.randn(M, 256, generator=gen, device=dev)
    W = torch.randn(256, 256, generator=gen, device=dev) * 0.1
    bias = torch.randn(256, generator=gen, device=dev)
    b = X.abs().max().reshape(1)
    # (three-part scheme: dropout at p != 1/2 has no instantiation of the contiguous-row kernel and
    #  runs the listed rows' one — take the plain product from the same kernel)
    from pygcn_amd import spmm as S
    other = S.gemm_scheme() == "bf16x3" and p not in (0.0, 0.5)
    plain = gemm_xw256(X, W, x_bound=b, rows=torch.arange(M, device=dev, dtype=torch.int32) if other else None)
    ident = CSRGraph(torch.arange(M + 1, device=dev, dtype=torch.int32),
                     torch.arange(M, device=dev, dtype=torch.int32), torch.ones(M, device=dev), (M, M))
    seed = 0x1234_5678_9ABC_DEF1
    want = spmm_csr(ident, plain, bias=bias, relu=True, dropout_p=p, seed=seed)
    ymax = torch.zeros(1, device=dev)
    got = gemm_xw256(X, W, x_bound=b, y_absmax=ymax, bias=bias, relu=True, dropout_p=p, seed=seed)
    assert got is not None and torch.equal(got, want)
    assert float(ymax) == float(want.abs().max())
    if p > 0:
        kept = (got != 0).float().mean().item()
        assert abs(kept - 0.5 * (1 - p)) < 0.02          # ~half survive the ReLU, (1 - p) of those the dropout
        other = gemm_xw256(X, W, x_bound=b, bias=bias, relu=True, dropout_p=p, seed=seed + 1)
        assert not torch.equal(other, got)
    # bias only / relu only
    dense = gemm_xw256(X, W, x_bound=b)
    assert torch.equal(gemm_xw256(X, W, x_bound=b, bias=bias), dense + bias)
    assert torch.equal(gemm_xw256(X, W, x_bound=b, relu=True), dense.clamp_min(0))
    # a device-resident seed gives the same mask as the same host seed
    sd = torch.tensor([seed & (2 ** 63 - 1)], dtype=torch.int64, device=dev)
    a1 = gemm_xw256(X, W, x_bound=b, bias=bias, relu=True, dropout_p=0.25, seed=sd)
    a2 = gemm_xw256(X, W, x_bound=b, bias=bias, relu=True, dropout_p=0.25, seed=int(sd.item()))
    assert torch.equal(a1, a2)


@pytest.mark.parametrize("K,N", [(128, 128), (128, 256), (256, 128)])
@pytest.mark.parametrize("p", [0.0, 0.4, 0.5])
def test_bf16_gemm_forward_epilogue(dev, K, N, p):
    """bias + ReLU + inverted dropout on the fp32 accumulators of the bf16 GEMM, rounded to bf16
    once; the keep bits are those of the SpMM epilogue for the same (seed, row, column) — read off
    an identity-adjacency product of an all-ones operand."""
    from pygcn_amd import CSRGraph, spmm_csr
    from pygcn_amd.spmm import gemm_bf16
    gen = torch.Generator(device=dev).manual_seed(K + N)
    M = 2777
    X = torch.randn(M, K, generator=gen, device=dev).bfloat16()
    W = (torch.randn(K, N, generator=gen, device=dev) * 0.2).bfloat16()
    bias = torch.randn(N, generator=gen, device=dev).bfloat16()
    seed = 0x0BAD_5EED_1234_5678
    got = gemm_bf16(X, W, bias=bias, relu=True, dropout_p=p, seed=seed)
    assert got is not None and got.dtype == torch.bfloat16 and got.shape == (M, N)
    pre = (X.double() @ W.double() + bias.double()).clamp_min(0)
    keep = torch.ones(M, N, dtype=torch.bool, device=dev)
    if p > 0:
        ident = CSRGraph(torch.arange(M + 1, device=dev, dtype=torch.int32),
                         torch.arange(M, device=dev, dtype=torch.int32), torch.ones(M, device=dev), (M, M))
        keep = spmm_csr(ident, torch.ones(M, N, device=dev).bfloat16(), relu=True, dropout_p=p, seed=seed) != 0
        assert abs(keep.float().mean().item() - (1 - p)) < 0.01
    want = torch.where(keep, pre / (1 - p), torch.zeros_like(pre))
    err = (got.double() - want).abs()
    assert bool((err <= 2.0 ** -8 * want.abs() + 1e-5 * float(want.abs().max())).all())
    clear = pre > 1e-3 * float(pre.max())                   # (away from the ReLU's rounding edge)
    assert torch.equal((got != 0) & clear, keep & clear)
    assert torch.equal(got, gemm_bf16(X, W, bias=bias, relu=True, dropout_p=p, seed=seed))
    if p > 0:
        assert not torch.equal(got, gemm_bf16(X, W, bias=bias, relu=True, dropout_p=p, seed=seed + 1))
    # bias only
    b_only = gemm_bf16(X, W, bias=bias)
    ref = X.double() @ W.double() + bias.double()
    assert bool(((b_only.double() - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-5 * float(ref.abs().max())).all())


def test_a_bound_that_is_too_small_is_never_silent(dev, h2_scheme):
    """the scaled fp16 GEMM trusts the caller's upper bound of max|X|.  A bound that is
    too small overflows the fp16 parts; that must surface — y_absmax becomes non-finite (its
    integer maximum keeps inf / NaN patterns), a consumer scaled by such a bound stores NaN, and
    the debug switch names the launch — never plausible-looking numbers."""
    import pygcn_amd.spmm as S
    from pygcn_amd.spmm import gemm_xw256, weight_grad_rows
    torch.manual_seed(0)
    M = 4096
    X = torch.randn(M, 256, device=dev)
    X[17, 3] = 5000.0                                   # the element the bound forgets
    W = torch.randn(256, 256, device=dev) * 0.06
    good, bad = X.abs().max().reshape(1), torch.full((1,), 0.05, device=dev)   # 5000 / 0.05 >> 2^17... and >> 4
    ymax = torch.zeros(1, device=dev)
    Y = gemm_xw256(X, W, x_bound=good, y_absmax=ymax)
    assert torch.isfinite(Y).all() and torch.isfinite(ymax).all()
    assert abs(ymax.item() - Y.abs().max().item()) <= 1e-6 * ymax.item()
    ymax_bad = torch.zeros(1, device=dev)
    Yb = gemm_xw256(X, W, x_bound=bad, y_absmax=ymax_bad)
    assert not torch.isfinite(Yb).all()                 # the overflow is visible in the output ...
    assert not torch.isfinite(ymax_bad).all()           # ... and in the reported maximum
    # downstream: a GEMM / weight gradient scaled by the non-finite bound stores NaN, not zeros
    Z = gemm_xw256(Y, W, x_bound=ymax_bad)
    assert torch.isnan(Z).all()
    gw = weight_grad_rows(Y, Y, a_bound=ymax_bad, g_bound=ymax)
    assert torch.isnan(gw).all()
    S.set_bound_check(True)
    try:
        with pytest.raises(RuntimeError, match="x_bound was smaller"):
            gemm_xw256(X, W, x_bound=bad, y_absmax=torch.zeros(1, device=dev))
        gemm_xw256(X, W, x_bound=good, y_absmax=torch.zeros(1, device=dev))
    finally:
        S.set_bound_check(False)


@pytest.mark.parametrize("m,listed", [(5000, False), (70000, True), (33, True), (16, False), (1, True)])
def test_bf16_weight_gradient_over_row_lists(dev, m, listed):
    """gcn_gemm_atg_bf16 (config C5's weight gradients): Σ_r A[ra[r]]ᵀ ⊗ G[rg[r]] for bf16 [*, 128]
    operands, fp32 accumulation — against an fp64 product of the same bf16-rounded values, with
    and without row lists (gathers fused into the loads), odd list lengths, duplicates."""
    from pygcn_amd.spmm import padded_row_list, weight_grad_rows
    torch.manual_seed(m)
    n = max(m, 64) * 3
    A = torch.randn(n, 128, device=dev).bfloat16()
    G = (torch.randn(n, 128, device=dev) * 0.01).bfloat16()
    if listed:
        ra = torch.randint(0, n, (m,), device=dev)
        rg = torch.randint(0, n, (m,), device=dev)
        got = weight_grad_rows(A, G, padded_row_list(ra), padded_row_list(rg), n_list=m)
        ref = A[ra].double().t() @ G[rg].double()
    else:
        got = weight_grad_rows(A[:m], G[:m])
        ref = A[:m].double().t() @ G[:m].double()
    assert got is not None and got.dtype == torch.bfloat16 and tuple(got.shape) == (128, 128)
    # fp32 accumulation of exact bf16 products, one final rounding to bf16
    err = (got.double() - ref).abs().max().item()
    assert err <= 2.0 ** -8 * ref.abs().max().item() + 1e-30, (err, ref.abs().max().item())
    # deterministic: the same launch twice gives the same bits
    again = weight_grad_rows(A, G, padded_row_list(ra), padded_row_list(rg), n_list=m) if listed else \
        weight_grad_rows(A[:m], G[:m])
    assert torch.equal(got, again)
    # shapes the kernel does not carry decline (the caller falls back to a library GEMM)
    assert weight_grad_rows(A[:, :64], G[:, :64]) is None


def test_bf16_gemm_with_backward_mask_in_the_store(dev):
    """gcn_gemm_xw_bf16 with `mask_src` / `mask_rows`: the backward of a fused ReLU / dropout epilogue
    (y = mask > 0 ? y * scale : 0) in the grad_input GEMM's own store at bf16 (config C5), the mask
    read through a row list — against the unfused composition on the same bf16 values."""
    from pygcn_amd.spmm import gemm_bf16
    torch.manual_seed(5)
    M, n = 7001, 30000
    X = torch.randn(M, 128, device=dev).bfloat16()
    W = (torch.randn(128, 128, device=dev) * 0.1).bfloat16()
    H = torch.relu(torch.randn(n, 128, device=dev)).bfloat16()            # ~half zeros, like h1
    rows = torch.randint(0, n, (M,), device=dev, dtype=torch.int32)
    plain = gemm_bf16(X, W)
    got = gemm_bf16(X, W, mask_src=H, mask_rows=rows, mask_scale=2.0)
    assert got is not None and got.dtype == torch.bfloat16
    acc = X.float() @ W.float()
    want = torch.where(H[rows.long()] > 0, acc * 2.0, torch.zeros_like(acc)).bfloat16()
    assert torch.equal(got == 0, want == 0) or ((got == 0) != (want == 0)).sum() <= 2      # (products that round to 0)
    err = (got.float() - want.float()).abs().max().item()
    assert err <= 2.0 ** -7 * want.float().abs().max().item()
    # without a row list the mask row is the output row; and the mask excludes the forward epilogue
    got2 = gemm_bf16(X, W, mask_src=H[:M].contiguous(), mask_scale=1.0)
    want2 = torch.where(H[:M] > 0, plain.float(), torch.zeros_like(acc))
    assert (got2.float() - want2).abs().max().item() <= 2.0 ** -7 * want2.abs().max().item()
    assert gemm_bf16(X, W, relu=True, mask_src=H[:M].contiguous()) is None


@pytest.mark.parametrize("M", [98_305, 400_003, 196_640])
def test_bf16_gemm_pipeline_across_tiles(dev, M):
    """The bf16 GEMM addresses its tiles through buffer descriptors (the range check replaces the
    tail branch), swaps two fragment register sets between tiles, requests the backward mask ahead
    of the next tile's prefetch, and leans on counted waits the compiler derives from that order
    (gcn_gemm.hip).  Heights that give the persistent waves one tile and a second only for some
    (98 305 = 3 072 waves x 32 rows + 1), several tiles with a ragged last one, and an even split:
    every store variant must repeat its bits across launches, the masked forms must equal the
    plain product masked afterwards (scale 1: the same rounding), the row-list form the
    own-row form, and the plain product an fp64 product of the same bf16 values."""
    from pygcn_amd.spmm import gemm_bf16
    g = torch.Generator(device=dev).manual_seed(M)
    X = torch.randn(M, 128, device=dev, generator=g).bfloat16()
    W = (torch.randn(128, 128, device=dev, generator=g) * 0.1).bfloat16()
    H = torch.relu(torch.randn(M, 128, device=dev, generator=g)).bfloat16()
    bias = torch.randn(128, device=dev, generator=g) * 0.1
    perm = torch.randperm(M, device=dev, generator=g).to(torch.int32)
    plain = gemm_bf16(X, W)
    idx = torch.cat([torch.arange(0, 4096, device=dev), torch.arange(M - 4096, M, device=dev)])
    ref = X[idx].double() @ W.double()
    assert (plain[idx].double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item()
    masked = gemm_bf16(X, W, mask_src=H, mask_scale=1.0)
    assert torch.equal(masked, torch.where(H > 0, plain, torch.zeros_like(plain)))
    listed = gemm_bf16(X, W, mask_src=H, mask_rows=perm, mask_scale=1.0)
    assert torch.equal(listed, torch.where(H[perm.long()] > 0, plain, torch.zeros_like(plain)))
    fwd = gemm_bf16(X, W, bias=bias, relu=True, dropout_p=0.5, seed=99)
    nodrop = gemm_bf16(X, W, bias=bias, relu=True)
    kept = fwd != 0
    assert torch.equal(nodrop == 0, nodrop == 0) and bool((nodrop[kept] > 0).all())
    # kept elements are the undropped values times 1 / (1 - p) = 2 (exact in bf16 up to the one rounding)
    assert (fwd[kept].float() - 2.0 * nodrop[kept].float()).abs().max().item() <= 2.0 ** -7 * nodrop.float().abs().max().item() * 2
    share = kept.float().sum().item() / max(1.0, (nodrop > 0).float().sum().item())
    assert 0.49 < share < 0.51
    for _ in range(4):
        assert torch.equal(gemm_bf16(X, W), plain)
        assert torch.equal(gemm_bf16(X, W, mask_src=H, mask_scale=1.0), masked)
        assert torch.equal(gemm_bf16(X, W, mask_src=H, mask_rows=perm, mask_scale=1.0), listed)
        assert torch.equal(gemm_bf16(X, W, bias=bias, relu=True, dropout_p=0.5, seed=99), fwd)


@pytest.mark.parametrize("K,N", [(128, 256), (256, 128)])
def test_bf16_gemm_mask_at_the_wider_shapes(dev, K, N):
    """The shapes whose accumulators leave no registers for a tile's whole mask fetch it per column
    block in the store section: same result as the plain product masked afterwards (scale 1), with
    and without a row list, ragged height, repeatable."""
    from pygcn_amd.spmm import gemm_bf16
    M = 70_003
    g = torch.Generator(device=dev).manual_seed(K + N)
    X = torch.randn(M, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(K, N, device=dev, generator=g) * 0.1).bfloat16()
    H = torch.relu(torch.randn(M, N, device=dev, generator=g)).bfloat16()
    perm = torch.randperm(M, device=dev, generator=g).to(torch.int32)
    plain = gemm_bf16(X, W)
    assert plain is not None
    masked = gemm_bf16(X, W, mask_src=H, mask_scale=1.0)
    listed = gemm_bf16(X, W, mask_src=H, mask_rows=perm, mask_scale=1.0)
    assert torch.equal(masked, torch.where(H > 0, plain, torch.zeros_like(plain)))
    assert torch.equal(listed, torch.where(H[perm.long()] > 0, plain, torch.zeros_like(plain)))
    assert torch.equal(gemm_bf16(X, W, mask_src=H, mask_rows=perm, mask_scale=1.0), listed)


def test_dma_pipeline_is_deterministic_across_tiles_and_launches(dev, gemm_scheme):
    """The fp32 GEMM moves X and W by asynchronous HBM -> LDS DMA with hand-counted waits
    (gcn_gemm.hip, GEMM_H2_XLDS): a wait that is one too weak would show as run-to-run noise.
    Many launches over inputs that give every persistent workgroup several tiles (cross-tile
    prefetch), a ragged last tile, a row list and both epilogues — all launches must store the same
    bits, and the plain result must match an fp64 product."""
    from pygcn_amd.spmm import gemm_xw256
    torch.manual_seed(11)
    for M in (300_007, 70_001, 257):
        X = torch.randn(M, 256, device=dev)
        W = torch.randn(256, 256, device=dev) * 0.06
        bias = torch.randn(256, device=dev) * 0.1
        b = X.abs().max().reshape(1)
        rows = torch.randint(0, M, (M // 2,), device=dev, dtype=torch.int32)
        H = torch.relu(torch.randn(M, 256, device=dev))
        perm = torch.randperm(M, device=dev).to(torch.int32)
        variants = {
            "plain": lambda: gemm_xw256(X, W, x_bound=b),
            "epilogue": lambda: gemm_xw256(X, W, x_bound=b, bias=bias, relu=True, dropout_p=0.3, seed=99),
            # (the contiguous-row kernel of the three-part scheme has its own store sections: dropout at 1/2,
            #  the backward mask read at the row and through a mask row list)
            "epilogue 1/2": lambda: gemm_xw256(X, W, x_bound=b, bias=bias, relu=True, dropout_p=0.5, seed=98),
            "mask": lambda: gemm_xw256(X, W, x_bound=b, mask_src=H, mask_scale=1.5),
            "mask rows": lambda: gemm_xw256(X, W, x_bound=b, mask_src=H, mask_rows=perm, mask_scale=1.5),
            "rows+mask": lambda: gemm_xw256(X, W, x_bound=b, rows=rows, mask_src=H, mask_scale=1.5),
        }
        results = {}
        for name, fn in variants.items():
            first = fn()
            results[name] = first
            assert torch.isfinite(first).all(), (name, M)
            for _ in range(12):
                assert torch.equal(fn(), first), (name, M)
        plain = results["plain"]
        assert torch.equal(results["mask"], torch.where(H > 0, plain * 1.5, torch.zeros_like(plain))), M
        assert torch.equal(results["mask rows"], torch.where(H[perm.long()] > 0, plain * 1.5, torch.zeros_like(plain))), M
        keep = results["epilogue 1/2"] != 0
        want = ((plain + bias).clamp_min(0) * 2.0)
        assert torch.equal(results["epilogue 1/2"][keep], want[keep]), M
        assert 0.2 < keep.float().mean().item() < 0.3, M          # ~half survive the ReLU, half of those the dropout
        ref = X[:2048].double() @ W.double()
        got = variants["plain"]()[:2048].double()
        assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-6
