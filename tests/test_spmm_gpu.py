"""GPU parity tests (run with -m gpu on an MI355X).  Every product call goes through the C-ABI
`gcn_spmm_csr` in pygcn_amd/csrc/libgcn_spmm.so; the checker is the CPU oracle (oracle/) and the
golden vectors captured from the imported reference (tests/golden/).

Tolerance: BASELINE.json north_star — within 1e-5 relative fp32, measured normwise
(max|got-ref| <= 1e-5 * max|ref|, conftest.assert_normwise; SURVEY §7 explains why elementwise
relative error is meaningless here).  bf16 storage is outside that contract: 2^-8 relative on
the final rounding, written at the test.
"""
import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_normwise, assert_parity, load_golden
from make_golden_cases import EDGE_CASES

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from pygcn_amd import _native
    _native.lib()   # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _graph(oracle_csr, dev, **kw):
    from pygcn_amd import CSRGraph
    rp = oracle_csr.rowptr.astype(np.int32)
    return CSRGraph(torch.from_numpy(rp).to(dev), torch.from_numpy(oracle_csr.col).to(dev),
                    torch.from_numpy(oracle_csr.val).to(dev), oracle_csr.shape, **kw)


# ------------------------------------------------------------------ golden edge cases
@pytest.mark.parametrize("case", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_layer_matches_reference_golden(oracle, dev, case):
    from pygcn_amd import GraphConvolution, sparse_mm as spmm
    g4 = load_golden("g4_edge_cases.npz")
    k = EDGE_CASES.index(case)
    name, nr, nc, nnz, fin, fout, bias, kw = case
    seed = 400 + 10 * k
    rows, cols, vals = gin.random_coo(nr, nc, nnz, seed=seed, **kw)
    # the adjacency exactly as the reference feeds it: uncoalesced torch sparse COO, int64 indices
    adj = torch.sparse_coo_tensor(np.vstack([rows, cols]), vals, (nr, nc)).to(dev)
    x = torch.from_numpy(gin.dense((nc, fin), seed + 1)).to(dev).requires_grad_(True)
    g = torch.from_numpy(gin.dense((nr, fout), seed + 2)).to(dev)
    layer = GraphConvolution(fin, fout, bias=bias).to(dev)
    with torch.no_grad():
        layer.weight.copy_(torch.from_numpy(g4[name + "/weight"]))
        if bias:
            layer.bias.copy_(torch.from_numpy(g4[name + "/bias"]))
    y = layer(x, adj)
    y.backward(g)
    assert_normwise(y.detach().cpu(), g4[name + "/y"], TOL, "y")
    assert_normwise(x.grad.cpu(), g4[name + "/grad_x"], TOL, "grad_x")
    assert_normwise(layer.weight.grad.cpu(), g4[name + "/grad_weight"], TOL, "grad_w")
    if bias:
        assert_normwise(layer.bias.grad.cpu(), g4[name + "/grad_bias"], TOL, "grad_b")
    # the bare products a4 / a6
    support = (x.detach() @ layer.weight.detach())
    assert_normwise(spmm(adj, support).cpu(), g4[name + "/spmm"], TOL, "spmm")
    gs = torch.zeros(nc, fout, device=dev, requires_grad=True)
    spmm(adj, gs).backward(g)
    assert_normwise(gs.grad.cpu(), g4[name + "/spmm_t"], TOL, "spmm_t")


def test_cora_step_matches_reference_golden(oracle, dev):
    """Config C2: Cora 2-layer GCN step on one MI355X vs the reference's CPU step (G2)."""
    from pygcn_amd import GCN
    from pygcn_amd.utils import load_data
    g1, g2 = load_golden("g1_init.npz"), load_golden("g2_cora_step.npz")
    adj, _, _, idx_train, _, _ = load_data()
    adj = adj.to(dev)
    x = torch.from_numpy(gin.cora_features()).to(dev).requires_grad_(True)
    labels = torch.from_numpy(gin.cora_labels()).to(dev)
    torch.manual_seed(42)
    model = GCN(1433, 16, 7, dropout=0.0).to(dev)
    for k in ("gc1", "gc2"):   # seeded init is bit-identical to the reference's (CPU test)
        np.testing.assert_array_equal(getattr(model, k).weight.detach().cpu().numpy(),
                                      g1[k + "_weight"])
    model.train()
    logp = model(x, adj)
    loss = torch.nn.functional.nll_loss(logp[idx_train.to(dev)], labels[idx_train.to(dev)])
    loss.backward()
    assert abs(loss.item() - float(g2["loss"])) <= TOL * abs(float(g2["loss"]))
    assert_normwise(logp.detach().cpu(), g2["logp"], TOL, "logp")
    assert_normwise(x.grad[:64].cpu(), g2["grad_x_head"], TOL, "grad_x")
    for k in ("gc1", "gc2"):
        assert_normwise(getattr(model, k).weight.grad.cpu(), g2[k + "_weight_grad"], TOL, k + ".w")
        assert_normwise(getattr(model, k).bias.grad.cpu(), g2[k + "_bias_grad"], TOL, k + ".b")


def test_generator_gcn_stack_matches_reference_golden(dev):
    from pygcn_amd.models import GCNStack
    g3 = load_golden("g3_generator_gcn.npz")
    n = 64
    rows, cols, vals = gin.random_coo(n, n, 400, seed=300)
    adj = torch.sparse_coo_tensor(np.vstack([rows, cols]), vals, (n, n)).to(dev)
    x = torch.from_numpy(gin.dense((n, 8), 301)).to(dev).requires_grad_(True)
    m = GCNStack(8, 32, 32, nlayers=3).to(dev)
    with torch.no_grad():
        for name, p in m.named_parameters():
            p.copy_(torch.from_numpy(g3["param_" + name]))
    y = m(x, adj)
    y.backward(torch.from_numpy(gin.dense((n, 32), 302)).to(dev))
    assert_normwise(y.detach().cpu(), g3["y"], TOL, "y")
    assert_normwise(x.grad.cpu(), g3["grad_x"], TOL, "grad_x")
    for name, p in m.named_parameters():
        assert_normwise(p.grad.cpu(), g3["grad_" + name], TOL, name)


def test_training_trajectory_matches_reference_golden(dev):
    """200 Adam epochs (dropout 0) vs the trajectory of the imported reference layer (G5).
    Gate 1e-5 on the first steps, 1e-3 on the whole curve (chained fp32 steps drift; the CPU
    oracle shows the same drift against the same fixture, tests/test_oracle_golden.py)."""
    from pygcn_amd import GCN
    from pygcn_amd.utils import accuracy, load_data
    g5 = load_golden("g5_trajectory.npz")
    adj, _, _, idx_train, _, _ = load_data()
    adj, idx_train = adj.to(dev), idx_train.to(dev)
    x = torch.from_numpy(gin.cora_features()).to(dev)
    labels = torch.from_numpy(gin.cora_labels()).to(dev)
    torch.manual_seed(42)
    model = GCN(1433, 16, 7, dropout=0.0).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    losses, accs = [], []
    for _ in range(200):
        model.train()
        opt.zero_grad()
        out = model(x, adj)
        loss = torch.nn.functional.nll_loss(out[idx_train], labels[idx_train])
        accs.append(accuracy(out[idx_train], labels[idx_train]).item())
        loss.backward()
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses[:5], g5["loss_train"][:5], rtol=1e-5)
    np.testing.assert_allclose(losses, g5["loss_train"], rtol=1e-3)
    assert np.abs(np.array(accs) - g5["acc_train"]).max() <= 1.5 / 140


# ------------------------------------------------------------------ oracle property tests
def _skewed_csr(oracle, n_rows, n_cols, avg_deg, seed, hubs=(), empties=0):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg_deg, size=n_rows)
    if empties:
        deg[rng.integers(0, n_rows, size=empties)] = 0
    for r, d in hubs:
        deg[r] = d
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    nnz = int(rowptr[-1])
    col = rng.integers(0, n_cols, size=nnz).astype(np.int32)
    val = (1.0 - rng.random(nnz)).astype(np.float32)
    return oracle.CSR(rowptr, col, val, (n_rows, n_cols))


@pytest.mark.parametrize("F", [1, 3, 7, 16, 20, 64, 100, 128, 200, 256, 260, 300, 512, 1000])
def test_spmm_matches_oracle_all_widths(oracle, dev, F):
    """Forward (a4) and transpose (a6) products on a skewed graph: empty rows, rows of length
    255/256/257 around the chunk threshold, hubs spanning many chunks."""
    from pygcn_amd import spmm_csr
    a = _skewed_csr(oracle, 3000, 2500, 6, seed=F, empties=300,
                    hubs=((5, 255), (6, 256), (7, 257), (100, 5000), (2999, 1025), (0, 700)))
    B = gin.dense((2500, F), 1000 + F)
    G = gin.dense((3000, F), 2000 + F)
    for lt, n_long, n_chunks in ((0, 4, 2 + 20 + 5 + 3), (1024, 2, 5 + 2)):   # 0 = default (256)
        g = _graph(a, dev, long_thresh=lt)
        out = spmm_csr(g, torch.from_numpy(B).to(dev))
        assert_normwise(out.cpu(), a.matmul(B), TOL, f"A@B F={F} L={lt}")
        out_t = spmm_csr(g.t(), torch.from_numpy(G).to(dev))
        assert_normwise(out_t.cpu(), a.t_matmul(G), TOL, f"A^T@G F={F} L={lt}")
        stats = g.schedule_stats()
        assert stats["n_long"] == n_long and stats["n_chunks"] == n_chunks


def test_spmm_int64_rowptr_and_strided_operands(oracle, dev):
    from pygcn_amd import CSRGraph, spmm_csr
    a = _skewed_csr(oracle, 700, 900, 9, seed=5, hubs=((3, 2000),), empties=40)
    g = CSRGraph(torch.from_numpy(a.rowptr).to(dev), torch.from_numpy(a.col).to(dev),
                 torch.from_numpy(a.val).to(dev), a.shape)
    assert g.rowptr.dtype == torch.int64
    big = torch.from_numpy(gin.dense((900, 320), 77)).to(dev)
    for F, view in ((256, big[:, :256]), (64, big[:, 64:128]), (63, big[:, 1:64])):
        assert not view.is_contiguous()
        out = spmm_csr(g, view)                       # ldb = 320 != F, offsets break alignment
        assert_normwise(out.cpu(), a.matmul(view.cpu().numpy()), TOL, f"strided F={F}")
    cm = torch.from_numpy(gin.dense((7, 900), 78)).to(dev).t()     # column-major [900, 7]
    assert cm.stride(1) != 1
    assert_normwise(spmm_csr(g, cm).cpu(), a.matmul(cm.cpu().numpy()), TOL, "non-unit stride")


def test_fused_bias_relu_epilogue(oracle, dev):
    from pygcn_amd import spmm_csr
    a = _skewed_csr(oracle, 1200, 1200, 5, seed=9, hubs=((11, 900),), empties=100)
    g = _graph(a, dev)
    for F in (256, 16, 7):
        B, b = gin.dense((1200, F), 31 + F), gin.dense((F,), 32 + F)
        ref = a.matmul(B) + b
        out = spmm_csr(g, torch.from_numpy(B).to(dev), bias=torch.from_numpy(b).to(dev))
        assert_normwise(out.cpu(), ref, TOL, f"bias F={F}")
        out = spmm_csr(g, torch.from_numpy(B).to(dev), bias=torch.from_numpy(b).to(dev), relu=True)
        assert_normwise(out.cpu(), np.maximum(ref, 0), TOL, f"bias+relu F={F}")
        # rows without stored entries must come out as bias exactly
        empty = np.nonzero(np.diff(a.rowptr) == 0)[0]
        np.testing.assert_array_equal(
            spmm_csr(g, torch.from_numpy(B).to(dev), bias=torch.from_numpy(b).to(dev))
            .cpu().numpy()[empty], np.broadcast_to(b, (len(empty), F)))


def test_inputs_not_mutated_and_deterministic(oracle, dev):
    from pygcn_amd import spmm_csr
    a = _skewed_csr(oracle, 2000, 2000, 8, seed=13, hubs=((1, 4000), (2, 300)))
    g = _graph(a, dev)
    B = torch.from_numpy(gin.dense((2000, 256), 14)).to(dev)
    B0, v0 = B.clone(), g.val.clone()
    o1, o2 = spmm_csr(g, B), spmm_csr(g, B)
    assert torch.equal(B, B0) and torch.equal(g.val, v0)
    assert torch.equal(o1, o2)    # chunk-ordered long-row sums: bitwise reproducible, no atomics


def test_non_finite_values_propagate_like_the_reference(oracle, dev):
    """inf in a gathered row reaches exactly the rows that reference it (0*inf hazards in
    masked lanes / padded slots would show up as NaN elsewhere)."""
    from pygcn_amd import spmm_csr
    a = _skewed_csr(oracle, 500, 400, 5, seed=21, hubs=((9, 600),))
    g = _graph(a, dev)
    B = gin.dense((400, 256), 22)
    B[17, 3] = np.inf
    ref = a.matmul(B)
    out = spmm_csr(g, torch.from_numpy(B).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(np.isfinite(out), np.isfinite(ref))
    fin = np.isfinite(ref)
    assert_normwise(np.where(fin, out, 0), np.where(fin, ref, 0), TOL, "finite part")


def test_bf16_storage_fp32_accumulate(oracle, dev):
    """Config C5 numerics: bf16 B/C, fp32 values and accumulation; compared with the fp32 oracle
    on bf16-rounded inputs; tolerance 2^-8 relative (one final rounding to bf16)."""
    from pygcn_amd import spmm_csr
    a = _skewed_csr(oracle, 1500, 1500, 7, seed=3, hubs=((4, 3000),), empties=50)
    g = _graph(a, dev)
    for F in (128, 512, 24, 5):
        Bb = torch.from_numpy(gin.dense((1500, F), 40 + F)).to(torch.bfloat16)
        ref = a.matmul(Bb.to(torch.float32).numpy())
        out = spmm_csr(g, Bb.to(dev))
        assert out.dtype == torch.bfloat16
        assert_normwise(out.float().cpu(), ref, 2.0 ** -8, f"bf16 F={F}")


def test_rmat_graph_forward_backward_vs_oracle(oracle, dev):
    """A scaled-down C3 (R-MAT, permuted, self-loops, row-normalized, F=256): layer forward and
    backward through autograd against the oracle."""
    from pygcn_amd import CSRGraph, GraphConvolution
    from pygcn_amd.utils import rmat_graph
    n = 30000
    rowptr, col, val = rmat_graph(n, 300000, seed=42, device="cpu")
    a = oracle.CSR(rowptr.numpy(), col.numpy(), val.numpy(), (n, n))
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    torch.manual_seed(0)
    layer = GraphConvolution(64, 256).to(dev)
    x = gin.dense((n, 64), 90)
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    y = layer(xg, g)
    go = gin.dense((n, 256), 91)
    y.backward(torch.from_numpy(go).to(dev))
    w, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    y_ref, _ = oracle.gc_forward(x, w, b, a)
    gx, gw, gb, _ = oracle.gc_backward(x, w, True, a, go)
    _, gw64, gb64 = oracle.gc_backward_f64(x, w, True, a, go, need_grad_x=False)
    assert_normwise(y.detach().cpu(), y_ref, TOL, "y")
    assert_normwise(xg.grad.cpu(), gx, TOL, "grad_x")
    assert_parity(layer.weight.grad.cpu(), gw, gw64, "grad_w")   # 30k-term fp32 reductions: float64 arbiter
    assert_parity(layer.bias.grad.cpu(), gb, gb64, "grad_b")


def test_errors_are_runtime_errors(dev):
    from pygcn_amd import CSRGraph, spmm_csr
    g = CSRGraph(torch.tensor([0, 1, 2], dtype=torch.int32, device=dev),
                 torch.tensor([0, 1], dtype=torch.int32, device=dev),
                 torch.ones(2, device=dev), (2, 2))
    with pytest.raises(RuntimeError, match="size mismatch"):
        spmm_csr(g, torch.ones(3, 4, device=dev))
    with pytest.raises(RuntimeError, match="float32 and bfloat16"):
        spmm_csr(g, torch.ones(2, 4, device=dev, dtype=torch.float64))
    with pytest.raises(RuntimeError, match="HIP device"):
        spmm_csr(g, torch.ones(2, 4))
    assert spmm_csr(g, torch.ones(2, 0, device=dev)).shape == (2, 0)
    # malformed CSR is rejected at construction (an out-of-range column would be an OOB gather)
    i32 = dict(dtype=torch.int32, device=dev)
    for rp, c in (([0, 1, 2], [0, 2]), ([0, 2, 1], [0, 1]), ([0, 1, 3], [0, 1]), ([1, 1, 2], [0, 1]),
                  ([0, 1, 2], [-1, 1])):
        with pytest.raises(RuntimeError, match="invalid CSR"):
            CSRGraph(torch.tensor(rp, **i32), torch.tensor(c, **i32), torch.ones(2, device=dev),
                     (2, 2))


def test_batched_samples_match_per_sample_loop(oracle, dev):
    """Row f3: k samples through one SpMM == the fork's per-sample loop (models.py:343-349),
    forward and backward, checked against the oracle run sample by sample."""
    from pygcn_amd import GraphConvolution
    k, n, fin, fout = 5, 800, 12, 48
    a = _skewed_csr(oracle, n, n, 6, seed=77, hubs=((3, 600),), empties=30)
    g = _graph(a, dev)
    torch.manual_seed(1)
    layer = GraphConvolution(fin, fout).to(dev)
    x = gin.dense((k, n, fin), 78)
    go = gin.dense((k, n, fout), 79)
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    y = layer(xg, g)
    assert y.shape == (k, n, fout)
    y.backward(torch.from_numpy(go).to(dev))
    w, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    gw_ref = np.zeros_like(w)
    gb_ref = np.zeros_like(b)
    gw64, gb64 = np.zeros(w.shape), np.zeros(b.shape)
    for i in range(k):
        yi, _ = oracle.gc_forward(x[i], w, b, a)
        gx, gw, gb, _ = oracle.gc_backward(x[i], w, True, a, go[i])
        assert_normwise(y[i].detach().cpu(), yi, TOL, f"y[{i}]")
        assert_normwise(xg.grad[i].cpu(), gx, TOL, f"grad_x[{i}]")
        gw_ref += gw
        gb_ref += gb
        _, w64, b64 = oracle.gc_backward_f64(x[i], w, True, a, go[i], need_grad_x=False)
        gw64 += w64
        gb64 += b64
    assert_parity(layer.weight.grad.cpu(), gw_ref, gw64, "grad_w")
    assert_parity(layer.bias.grad.cpu(), gb_ref, gb64, "grad_b")
    # the loop itself on the GPU path gives the same numbers
    loop = torch.stack([layer(torch.from_numpy(x[i]).to(dev), g) for i in range(k)])
    assert_normwise(loop.detach().cpu(), y.detach().cpu().numpy(), TOL, "loop vs batched")


@pytest.mark.parametrize("p", [0.5, 0.3])
@pytest.mark.parametrize("F,dtype", [(256, torch.float32), (16, torch.float32), (7, torch.float32),
                                     (300, torch.float32), (1024, torch.float32), (128, torch.bfloat16)])
def test_fused_relu_dropout_epilogue_matches_philox_restatement(oracle, dev, F, dtype, p):
    """Row f1: bias + ReLU + inverted dropout inside the store.  The mask is a pure function of
    (seed, row, f): restated in numpy (oracle.dropout_keep) and compared exactly, for every kernel
    variant (wide, narrow vector, narrow scalar, bf16, long rows) and both forms of the keep
    function (p = 1/2: 128 one-bit fields per Philox call; any other p: eight 16-bit fields)."""
    from pygcn_amd import spmm_csr
    n = 1500
    a = _skewed_csr(oracle, n, n, 6, seed=5, hubs=((2, 900), (700, 300)), empties=60)
    g = _graph(a, dev)
    B = torch.from_numpy(gin.dense((n, F), 11)).to(dtype)
    b = gin.dense((F,), 12)
    seed = 0x1234567890ABCDEF
    plain = spmm_csr(g, B.to(dev), bias=torch.from_numpy(b).to(dev), relu=True).float().cpu().numpy()
    out = spmm_csr(g, B.to(dev), bias=torch.from_numpy(b).to(dev), relu=True, dropout_p=p,
                   seed=seed).float().cpu().numpy()
    keep = oracle.dropout_keep(seed, np.arange(n), F, p)
    assert abs(keep.mean() - (1 - p)) < 0.03
    np.testing.assert_array_equal(out[~keep], 0.0)
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-6
    np.testing.assert_allclose(out[keep], plain[keep] * oracle.dropout_scale(p), rtol=tol, atol=1e-30)
    out2 = spmm_csr(g, B.to(dev), bias=torch.from_numpy(b).to(dev), relu=True, dropout_p=p,
                    seed=seed + 1).float().cpu().numpy()
    assert (out2 == 0).mean() != (out == 0).mean() or not np.array_equal(out2 == 0, out == 0)


def test_fused_dropout_autograd_and_model(oracle, dev):
    """Backward through the fused epilogue uses out > 0 as the combined ReLU/dropout mask."""
    from pygcn_amd import GCN, GraphConvolution
    from pygcn_amd.spmm import relu_dropout_backward
    n, fin, fout, p = 900, 10, 64, 0.3
    a = _skewed_csr(oracle, n, n, 5, seed=31, hubs=((1, 500),), empties=20)
    g = _graph(a, dev)
    torch.manual_seed(3)
    layer = GraphConvolution(fin, fout).to(dev)
    x = gin.dense((n, fin), 32)
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    y = layer(xg, g, relu=True, dropout=p)
    go = gin.dense((n, fout), 33)
    y.backward(torch.from_numpy(go).to(dev))
    yn = y.detach().cpu().numpy()
    w, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    z, _ = oracle.gc_forward(x, w, b, a)
    kept = yn != 0
    assert abs(kept[z > 0].mean() - (1 - p)) < 0.02 and not kept[z <= 0].any()
    np.testing.assert_allclose(yn[kept], (z / (1 - p))[kept], rtol=2e-5, atol=1e-6)
    g_pre = np.where(kept, go * oracle.dropout_scale(p), 0).astype(np.float32)
    gx, gw, gb, _ = oracle.gc_backward(x, w, True, a, g_pre)
    _, gw64, gb64 = oracle.gc_backward_f64(x, w, True, a, g_pre, need_grad_x=False)
    assert_normwise(xg.grad.cpu(), gx, TOL, "grad_x")
    assert_parity(layer.weight.grad.cpu(), gw, gw64, "grad_w")
    assert_parity(layer.bias.grad.cpu(), gb, gb64, "grad_b")
    # the standalone backward kernel, odd sizes / bf16 / aliasing-free
    for dt, nel in ((torch.float32, 1000003), (torch.bfloat16, 4096), (torch.bfloat16, 77)):
        go_t = torch.randn(nel, device=dev).to(dt)
        o_t = torch.randn(nel, device=dev).to(dt)
        got = relu_dropout_backward(go_t, o_t, 1.25).float()
        ref = torch.where(o_t.float() > 0, go_t.float() * 1.25, torch.zeros_like(got))
        assert torch.allclose(got, ref.to(dt).float(), rtol=2.0 ** -7 if dt == torch.bfloat16 else 0)
    # model: eval mode is deterministic, train mode with dropout differs between calls
    torch.manual_seed(0)
    m = GCN(fin, 32, 8, dropout=0.5).to(dev)
    m.eval()
    e1, e2 = m(xg.detach(), g), m(xg.detach(), g)
    assert torch.equal(e1, e2)
    m.train()
    t1, t2 = m(xg.detach(), g), m(xg.detach(), g)
    assert not torch.equal(t1, t2)
    torch.manual_seed(9)
    r1 = m(xg.detach(), g)
    torch.manual_seed(9)
    r2 = m(xg.detach(), g)
    assert torch.equal(r1, r2)     # reproducible under torch.manual_seed


def test_randomized_shapes_and_schedules(oracle, dev):
    """40 seeded random problems over shape, skew, feature width, dtype of rowptr, epilogue and
    the schedule knobs of the ABI (item_cost / long_thresh down to values that make almost every
    row a chunked long row or a single-row item)."""
    from pygcn_amd import CSRGraph, spmm_csr
    rng = np.random.default_rng(2024)
    for case in range(40):
        n_rows = int(rng.integers(1, 700))
        n_cols = int(rng.integers(1, 700))
        F = int(rng.choice([1, 2, 5, 8, 31, 32, 48, 64, 96, 128, 132, 256, 384, 515]))
        deg = rng.poisson(rng.choice([0.3, 2, 9]), size=n_rows)
        for _ in range(int(rng.integers(0, 4))):
            deg[rng.integers(0, n_rows)] = int(rng.integers(50, 1500))
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        nnz = int(rowptr[-1])
        col = rng.integers(0, n_cols, size=nnz).astype(np.int32)
        val = rng.standard_normal(nnz).astype(np.float32)
        a = oracle.CSR(rowptr, col, val, (n_rows, n_cols))
        kw = dict(item_cost=int(rng.choice([0, 1, 8, 33, 200])),
                  long_thresh=int(rng.choice([0, 1, 4, 37, 600])))
        rp_t = torch.from_numpy(rowptr if case % 3 == 0 else rowptr.astype(np.int32))
        g = CSRGraph(rp_t.to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev),
                     (n_rows, n_cols), **kw)
        B = gin.dense((n_cols, F), 5000 + case)
        bias = gin.dense((F,), 6000 + case) if case % 2 else None
        relu = bool(case % 4 == 1)
        ref = a.matmul(B)
        if bias is not None:
            ref = ref + bias
        if relu:
            ref = np.maximum(ref, 0)
        out = spmm_csr(g, torch.from_numpy(B).to(dev),
                       bias=None if bias is None else torch.from_numpy(bias).to(dev), relu=relu)
        scale = max(np.abs(a.matmul(B)).max(), 1e-30) if nnz else 1.0
        err = np.abs(out.cpu().numpy().astype(np.float64) - ref).max() if ref.size else 0.0
        bound = 1e-5 * max(scale, np.abs(ref).max() if ref.size else 0.0)
        assert err <= bound, f"case {case}: n={n_rows}x{n_cols} F={F} {kw} err {err:.3e}"
        G = gin.dense((n_rows, F), 7000 + case)
        out_t = spmm_csr(g.t(), torch.from_numpy(G).to(dev))
        assert_normwise(out_t.cpu(), a.t_matmul(G), TOL, f"case {case} transpose")


def test_bf16_model_end_to_end(oracle, dev):
    """Config C5 numerics through the whole layer stack against the ORACLE: the bf16 model
    (bf16 parameters / activations, fp32 adjacency values and accumulation) is compared with the
    oracle's fp32 forward / loss / backward (`gcn2_loss_backward`) run on the SAME bf16-rounded
    parameters and inputs.  What differs is only the storage rounding of the intermediate
    activations and gradients (support, hidden layer, log-probabilities: one rounding each), so the
    gates are a few bf16 ulps: 2^-6 normwise on the log-probabilities after two layers, 2^-4 on
    the parameter gradients (bf16 is outside the 1e-5 contract)."""
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.utils import rmat_graph
    n, F = 20000, 128
    rowptr, col, val = rmat_graph(n, 200000, seed=3, device="cpu")
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    x16 = torch.from_numpy(gin.dense((n, F), 1)).to(torch.bfloat16)
    y = np.random.default_rng(2).integers(0, F, n)
    idx = np.arange(n // 10)
    torch.manual_seed(5)
    m16 = GCN(F, F, F, dropout=0.0).to(torch.bfloat16).to(dev)
    out16 = m16(x16.to(dev), g)
    assert out16.dtype == torch.bfloat16
    idx_t = torch.from_numpy(idx).to(dev)
    loss = torch.nn.functional.nll_loss(out16[idx_t].float(), torch.from_numpy(y).to(dev)[idx_t])
    loss.backward()
    a = oracle.CSR(rowptr.numpy().astype(np.int64), col.numpy(), val.numpy(), (n, n))
    p = {k: v.detach().float().cpu().numpy() for k, v in m16.state_dict().items()}   # bf16-rounded
    ref_loss, fw, grads, _ = oracle.gcn2_loss_backward(x16.float().numpy(), a, p, y, idx)
    assert_normwise(out16.float().detach().cpu(), fw["logp"], 2.0 ** -6, "logp")
    assert abs(loss.item() - ref_loss) <= 2.0 ** -6 * abs(ref_loss)
    for k, v in grads.items():
        mod, name = k.split(".")
        got = getattr(getattr(m16, mod), name).grad
        assert got.dtype == torch.bfloat16
        assert_normwise(got.float().cpu(), v, 2.0 ** -4, k + ".grad")


@pytest.mark.parametrize("idx64", [False, True])
def test_native_transpose_matches_oracle_bitwise(oracle, dev, idx64):
    """Row f4: CSR(A^T) built on the device equals the oracle's stable counting-sort transpose
    entry for entry (same order inside every row), for int32 and int64 row pointers."""
    from pygcn_amd import CSRGraph
    a = _skewed_csr(oracle, 4000, 2500, 7, seed=12, empties=400,
                    hubs=((5, 3000), (3999, 1), (0, 600)))
    rp = a.rowptr if idx64 else a.rowptr.astype(np.int32)
    g = CSRGraph(torch.from_numpy(rp).to(dev), torch.from_numpy(a.col).to(dev),
                 torch.from_numpy(a.val).to(dev), a.shape)
    t = g.t()
    rp_t, col_t, val_t = oracle.csr_transpose(a.rowptr, a.col, a.val, 2500)
    assert t.shape == (2500, 4000) and t.rowptr.dtype == g.rowptr.dtype
    np.testing.assert_array_equal(t.rowptr.cpu().numpy(), rp_t)
    np.testing.assert_array_equal(t.col.cpu().numpy(), col_t)
    np.testing.assert_array_equal(t.val.cpu().numpy(), val_t)
    assert t.t() is g
    # degenerate: no stored entries
    e = CSRGraph(torch.zeros(6, dtype=torch.int32, device=dev),
                 torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, device=dev), (5, 9))
    assert e.t().rowptr.cpu().tolist() == [0] * 10 and e.t().nnz == 0


def test_native_row_normalize_matches_reference_semantics(dev):
    """`normalize(mx)` of the reference (utils.py:390-397) on the device, in place."""
    import scipy.sparse as sp
    from pygcn_amd import CSRGraph
    from pygcn_amd.utils import normalize
    rng = np.random.default_rng(4)
    m = sp.random(3000, 3000, density=0.003, random_state=4, format="csr", dtype=np.float32)
    m.data = rng.random(m.nnz).astype(np.float32)
    m = m.tolil()
    m[7, :] = 0
    m[11, :200] = rng.random(200)       # a long row
    m = m.tocsr()
    m.sort_indices()
    g = CSRGraph.from_scipy(m, device=dev)
    g.val[g.rowptr[20].item():g.rowptr[21].item()] = 0.0      # stored zeros: row sums to 0
    m.data[m.indptr[20]:m.indptr[21]] = 0.0
    ref = sp.csr_matrix(normalize(m))
    g.row_normalize_()
    got = sp.csr_matrix((g.val.cpu().numpy(), g.col.cpu().numpy(), g.rowptr.cpu().numpy()),
                        shape=(3000, 3000))
    # (scipy's product drops the explicit zeros, so compare as dense matrices)
    np.testing.assert_allclose(got.toarray(), ref.toarray(), rtol=2e-6, atol=0)
    assert not np.isnan(g.val.cpu().numpy()).any()
    np.testing.assert_array_equal(got.toarray()[[7, 20]], 0.0)


def test_c_abi_from_plain_cpp_without_torch(dev, tmp_path):
    """The boundary is a C-ABI: a stand-alone C++ program (hipMalloc + the entry points of
    include/gcn_spmm.h, no PyTorch) runs forward, device transpose and backward and checks them."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.join(ROOT, "pygcn_amd", "csrc")
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "c_abi_smoke.cpp"), "-L", libdir,
                           "-lgcn_spmm", f"-Wl,-rpath,{libdir}", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C_ABI_SMOKE OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("F", [4, 16, 128, 256, 512, 1024])
def test_backward_with_colsum_matches_torch(dev, F):
    """One-pass grad_pre + bias gradient (column sums) vs torch, masked and unmasked, odd row
    counts; deterministic across launches."""
    from pygcn_amd.spmm import backward_with_colsum
    for n in (1, 63, 1000, 70001):
        gen = torch.Generator(device=dev).manual_seed(n + F)
        go = torch.randn(n, F, generator=gen, device=dev)
        out = torch.randn(n, F, generator=gen, device=dev)
        go[::3] = 0                               # a third of the rows entirely zero
        gp, cs, hint = backward_with_colsum(go, out, 1.5)
        ref = torch.where(out > 0, go * 1.5, torch.zeros_like(go))
        assert torch.equal(gp, ref)
        if F <= 256:
            from pygcn_amd.spmm import row_bitmap
            bits, cnt = hint
            rb, rc = row_bitmap(ref)
            assert torch.equal(bits, rb) and int(cnt) == int(rc) == int((ref != 0).any(1).sum())
        else:
            assert hint is None
        ref_cs = ref.double().sum(0)
        assert float((cs.double() - ref_cs).abs().max()) <= 1e-5 * float(ref.abs().sum(0).max()) + 1e-6
        gp2, cs2, _ = backward_with_colsum(go, out, 1.5)
        assert torch.equal(cs, cs2)
        gq, cq, hq = backward_with_colsum(go, None, 1.0)
        assert gq is go
        if F <= 256:
            from pygcn_amd.spmm import row_bitmap
            assert torch.equal(hq[0], row_bitmap(go)[0])
        assert float((cq.double() - go.double().sum(0)).abs().max()) <= \
            1e-5 * float(go.abs().sum(0).max()) + 1e-6
    assert backward_with_colsum(torch.randn(10, 7, device=dev)) is None
    assert backward_with_colsum(torch.randn(10, 12, device=dev).bfloat16()) is None   # 12 % 8 != 0


@pytest.mark.parametrize("F", [8, 128, 512])
def test_backward_with_colsum_bf16(dev, F):
    """bf16 storage (config C5): same pass on 8-element lanes; sums and flags follow the stored
    (rounded) values."""
    from pygcn_amd.spmm import backward_with_colsum, row_bitmap
    for n in (5, 1000, 40001):
        gen = torch.Generator(device=dev).manual_seed(n + F)
        go = torch.randn(n, F, generator=gen, device=dev).bfloat16()
        out = torch.randn(n, F, generator=gen, device=dev).bfloat16()
        go[1::4] = 0
        gp, cs, hint = backward_with_colsum(go, out, 2.0)
        ref = torch.where(out.float() > 0, go.float() * 2.0, torch.zeros(n, F, device=dev)).bfloat16()
        assert gp.dtype == torch.bfloat16 and torch.equal(gp, ref)
        ref_cs = ref.double().sum(0)
        tol = 2.0 ** -7 * float(ref.float().abs().sum(0).max()) + 1e-3
        assert float((cs.double() - ref_cs).abs().max()) <= tol
        bits, cnt = hint
        rb, rc = row_bitmap(ref)
        assert torch.equal(bits, rb) and int(cnt) == int(rc)


@pytest.mark.parametrize("F,dtype", [(256, torch.float32), (512, torch.float32), (64, torch.float32),
                                     (16, torch.float32), (7, torch.float32), (128, torch.bfloat16)])
@pytest.mark.parametrize("density", [0.02, 0.5, 0.9])
def test_row_sparse_operand_hint_changes_nothing_but_traffic(oracle, dev, F, dtype, density):
    """Rows of B flagged all-zero are not gathered: the product must be identical with and
    without the hint, on every kernel path, whether the device-side count enables the flags
    (density < 3/4) or not; stale flags on non-zero rows are the caller's bug and not tested."""
    from pygcn_amd import spmm_csr
    a = _skewed_csr(oracle, 2500, 2200, 6, seed=int(F + 100 * density), empties=100,
                    hubs=((3, 1200), (900, 300), (901, 40)))
    g = _graph(a, dev)
    gen = torch.Generator(device=dev).manual_seed(F)
    B = torch.randn(2200, F, generator=gen, device=dev)
    keep = torch.rand(2200, generator=gen, device=dev) < density
    B = (B * keep[:, None]).to(dtype)
    from pygcn_amd.spmm import row_bitmap
    flags, cnt = row_bitmap(B)
    plain = spmm_csr(g, B)
    hinted = spmm_csr(g, B, b_hint=(flags, cnt))
    if F >= 256 and dtype == torch.float32:
        assert torch.equal(plain, hinted)     # wide kernel: same summation order, bitwise equal
    else:                                     # narrow kernel: survivors are re-dealt to the lane
        assert_normwise(hinted.float().cpu(), plain.float().cpu().numpy(),   # groups: rounding order
                        2.0 ** -8 if dtype == torch.bfloat16 else 1e-6, "hinted vs plain")
    ref = a.matmul(B.float().cpu().numpy())
    assert_normwise(hinted.float().cpu(), ref, 2.0 ** -8 if dtype == torch.bfloat16 else TOL, "hinted")
    # non-finite values in a flagged-NONZERO row still propagate
    if dtype == torch.float32 and density < 0.75:
        B2 = B.clone()
        r = int(torch.nonzero(keep)[0])
        B2[r, 0] = float("inf")
        assert torch.equal(spmm_csr(g, B2).isfinite(), spmm_csr(g, B2, b_hint=(flags, cnt)).isfinite())


@pytest.mark.parametrize("F,dtype", [(256, torch.float32), (64, torch.float32), (7, torch.float32),
                                     (128, torch.bfloat16)])
def test_two_block_dense_operand(oracle, dev, F, dtype):
    """[B ; B2] stacked by rows without materialising it (the sharded path's own rows | halo
    rows): identical to the product with the concatenated operand, on every kernel path."""
    from pygcn_amd import spmm_csr
    a = _skewed_csr(oracle, 1800, 2000, 6, seed=F + 3, empties=50, hubs=((5, 900), (6, 30)))
    g = _graph(a, dev)
    gen = torch.Generator(device=dev).manual_seed(F)
    B = torch.randn(2000, F, generator=gen, device=dev).to(dtype)
    for split in (0, 1, 777, 1999, 2000):
        own, halo = B[:split], B[split:].clone()
        got = spmm_csr(g, own, B2=halo)
        assert torch.equal(got, spmm_csr(g, B)), split
    wide = torch.randn(2000, 2 * F + 8, generator=gen, device=dev).to(dtype)   # strided blocks
    got = spmm_csr(g, wide[:1000, :F], B2=wide[1000:, F:2 * F])
    ref = spmm_csr(g, torch.cat([wide[:1000, :F], wide[1000:, F:2 * F]]).contiguous())
    assert torch.equal(got, ref)
    with pytest.raises(RuntimeError, match="size mismatch"):
        spmm_csr(g, B[:100], B2=B[:100])


@pytest.mark.parametrize("F,dtype", [(256, torch.float32), (600, torch.float32), (64, torch.float32),
                                     (16, torch.float32), (7, torch.float32), (128, torch.bfloat16),
                                     (512, torch.bfloat16)])
def test_output_row_flags(oracle, dev, F, dtype):
    """gcn_epilogue.c_row_nonzero: byte r is set exactly where the stored row r has a non-zero
    element — on the wide, narrow (short / medium rows) and long-row (partial slab + reduce)
    paths, with and without the operand hint, and through a ReLU that zeroes whole rows."""
    from pygcn_amd import spmm_csr
    from pygcn_amd.spmm import row_bitmap
    a = _skewed_csr(oracle, 2500, 2200, 6, seed=F + 11, empties=100,
                    hubs=((3, 1200), (900, 300), (901, 40)))
    g = _graph(a, dev)
    gen = torch.Generator(device=dev).manual_seed(F)
    B = torch.randn(2200, F, generator=gen, device=dev)
    keep = torch.rand(2200, generator=gen, device=dev) < 0.03
    keep[5] = True
    B = (B * keep[:, None]).to(dtype)
    for hint in (None, row_bitmap(B)):
        flags = torch.zeros(2500, dtype=torch.uint8, device=dev)
        out = spmm_csr(g, B, b_hint=hint, c_flags=flags)
        expect = (out != 0).any(1)
        assert torch.equal(flags.bool(), expect)
        assert 0 < int(expect.sum()) < 2500                   # the case exercises both outcomes
        assert torch.equal(out, spmm_csr(g, B, b_hint=hint))  # the flags change nothing else
        # skip mode: all-zero rows are not written at all (long rows excepted: always stored)
        flags2 = torch.zeros(2500, dtype=torch.uint8, device=dev)
        out2 = torch.full((2500, F), float("nan"), device=dev).to(dtype)
        spmm_csr(g, B, b_hint=hint, c_flags=flags2, skip_zero_rows=True, out=out2)
        assert torch.equal(flags2, flags)
        assert torch.equal(out2[expect], out[expect])
        rest = out2[~expect]
        assert bool((rest.isnan().all(1) | (rest == 0).all(1)).all())
        if F <= 256 or (dtype == torch.bfloat16 and F <= 512):
            assert int(rest.isnan().all(1).sum()) >= rest.shape[0] - 3    # all but the long rows
    bias = -torch.rand(F, device=dev) * 0.05                  # negative bias + ReLU: whole rows -> 0
    flags = torch.zeros(2500, dtype=torch.uint8, device=dev)
    out = spmm_csr(g, B, bias=bias, relu=True, c_flags=flags)
    assert torch.equal(flags.bool(), (out != 0).any(1))
    with pytest.raises(RuntimeError, match="c_flags"):
        spmm_csr(g, B, c_flags=torch.zeros(2499, dtype=torch.uint8, device=dev))


def test_layer_backward_row_compaction(oracle, dev):
    """GraphConvFunction runs the weight / input gradient GEMMs on the non-zero rows of
    Aᵀ·grad_pre only when the loss touches few vertices.  Same gradients as the uncompacted
    backward (up to the summation order of grad_W) and as the two-node composition
    DenseMMFunction + SpMMFunction."""
    import importlib
    from pygcn_amd import CSRGraph, GraphConvolution
    from pygcn_amd.utils import rmat_graph
    S = importlib.import_module("pygcn_amd.spmm")   # `pygcn_amd.spmm` the attribute is the function
    n = 1 << 18
    rowptr, col, val = rmat_graph(n, 4 * n, seed=5, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    assert n >= S.MIN_ROWS
    torch.manual_seed(0)
    l1, l2 = GraphConvolution(48, 256).to(dev), GraphConvolution(256, 256).to(dev)
    x = torch.randn(n, 48, device=dev)
    idx = torch.randperm(n, device=dev)[: n // 400]
    tgt = torch.randn(idx.numel(), 256, device=dev)

    def grads(compaction, fused=True):
        S.set_row_compaction(compaction)
        S._poison_unwritten = True      # rows left unwritten on purpose hold NaN: nobody may read them
        try:
            for p in list(l1.parameters()) + list(l2.parameters()):
                p.grad = None
            torch.manual_seed(1)
            if fused:
                h = l2(l1(x, g, relu=True, dropout=0.25), g)
            else:
                seed = S.next_dropout_seed()
                h = S.SpMMFunction.apply(g, S.DenseMMFunction.apply(x, l1.weight), l1.bias, True, 0.25, seed)
                h = S.SpMMFunction.apply(g, S.DenseMMFunction.apply(h, l2.weight), l2.bias)
            ((h[idx] - tgt) ** 2).sum().backward()
            return [p.grad.clone() for p in list(l1.parameters()) + list(l2.parameters())]
        finally:
            S.set_row_compaction(True)
            S._poison_unwritten = False

    calls = []
    orig = S._dense_grads
    S._dense_grads = lambda *a, **k: (calls.append(a[5] if len(a) > 5 else k.get("rows")), orig(*a, **k))[1]
    try:
        compact = grads(True)
    finally:
        S._dense_grads = orig
    assert any(r is not None and 0 < r.numel() < n // 3 for r in calls), "compaction did not engage"
    dense = grads(False)
    two_node = grads(False, fused=False)
    for c, d, t in zip(compact, dense, two_node):
        assert torch.equal(d, t)
        assert_normwise(c.cpu(), d.cpu().numpy(), TOL, "compacted vs dense backward")


@pytest.mark.parametrize("F,dtype", [(256, torch.float32), (200, torch.float32), (64, torch.float32),
                                     (16, torch.float32), (7, torch.float32), (33, torch.float32),
                                     (128, torch.bfloat16), (512, torch.bfloat16)])
def test_fused_log_softmax_epilogue(oracle, dev, F, dtype):
    """gcn_epilogue.log_softmax: the stored rows are log_softmax(A·B + bias) — wide, narrow
    (row-per-lane-group, medium rows, VEC = 1 fallback) and long-row reduce paths; empty rows
    give log_softmax(bias)."""
    from pygcn_amd import spmm_csr
    a = _skewed_csr(oracle, 2500, 2200, 6, seed=F + 21, empties=100,
                    hubs=((3, 1200), (900, 300), (901, 40)))
    g = _graph(a, dev)
    gen = torch.Generator(device=dev).manual_seed(F)
    B = (3.0 * torch.randn(2200, F, generator=gen, device=dev)).to(dtype)
    bias = torch.randn(F, generator=gen, device=dev)
    got = spmm_csr(g, B, bias=bias, log_softmax=True)
    z = a.matmul(B.float().cpu().numpy()).astype(np.float64) + bias.cpu().numpy().astype(np.float64)
    ref = z - z.max(1, keepdims=True)
    ref = ref - np.log(np.exp(ref).sum(1, keepdims=True))
    tol = 2.0 ** -8 if dtype == torch.bfloat16 else TOL
    assert_normwise(got.float().cpu(), ref.astype(np.float32), tol, "fused log_softmax")
    if dtype == torch.float32:   # rows are normalised: logsumexp(row) = 0
        assert float(torch.logsumexp(got.double(), 1).abs().max()) < 1e-5
    with pytest.raises(RuntimeError, match="log_softmax"):
        spmm_csr(g, B, bias=bias, relu=True, log_softmax=True)
    if dtype == torch.float32:
        wide = torch.randn(2200, 300, device=dev)
        with pytest.raises(RuntimeError, match="one wavefront"):
            spmm_csr(g, wide, log_softmax=True)


@pytest.mark.parametrize("F,dtype", [(256, torch.float32), (64, torch.float32), (16, torch.float32),
                                     (128, torch.bfloat16)])
def test_log_softmax_backward_with_colsum(dev, F, dtype):
    """gcn_log_softmax_backward_colsum against torch's log_softmax autograd, with mostly-zero
    gradient rows (the idx_train pattern) and the by-products (column sums, row bitmap, count)."""
    from pygcn_amd.spmm import backward_with_colsum, row_bitmap
    n = 5000
    gen = torch.Generator(device=dev).manual_seed(F)
    z = (2.0 * torch.randn(n, F, generator=gen, device=dev)).requires_grad_(True)
    logp = torch.log_softmax(z, 1)
    g = torch.randn(n, F, generator=gen, device=dev)
    g = g * (torch.rand(n, 1, generator=gen, device=dev) < 0.05)
    logp.backward(g)
    ref = z.grad
    gp, cs, hint = backward_with_colsum(g.to(dtype).contiguous(), logp.detach().to(dtype).contiguous(),
                                        log_softmax=True)
    if dtype == torch.float32:
        assert_normwise(gp.cpu(), ref.cpu().numpy(), TOL, "log_softmax backward")
        assert_normwise(cs.cpu(), ref.sum(0).cpu().numpy(), 1e-4, "bias gradient")
    else:
        assert_normwise(gp.float().cpu(), ref.cpu().numpy(), 2.0 ** -6, "log_softmax backward bf16")
    bits, cnt = hint
    rb, rc = row_bitmap(gp)
    assert torch.equal(bits, rb) and int(cnt) == int(rc)
    assert int(cnt) == int((g != 0).any(1).sum())       # zero gradient rows stay exactly zero


def test_model_uses_fused_log_softmax_and_matches_torch(oracle, dev):
    """GCN.forward ends in the fused epilogue; output and every parameter gradient equal the
    unfused composition F.log_softmax(gc2(...)) (7 classes: torch backward fallback; 64 classes:
    HIP backward)."""
    import importlib
    from pygcn_amd import GCN
    S = importlib.import_module("pygcn_amd.spmm")
    a = _skewed_csr(oracle, 3000, 3000, 5, seed=77, empties=20, hubs=((9, 700),))
    g = _graph(a, dev)
    for nclass in (7, 64):
        torch.manual_seed(3)
        model = GCN(40, 32, nclass, 0.0).to(dev)
        x = torch.randn(3000, 40, device=dev)
        idx = torch.arange(0, 3000, 20, device=dev)
        lab = torch.randint(0, nclass, (idx.numel(),), device=dev)
        seen = []
        orig = S.spmm_csr
        Fz = importlib.import_module("pygcn_amd.fused")        # (the model's one node calls it from there)
        S.spmm_csr = Fz.spmm_csr = lambda *a_, **k: (seen.append(k.get("log_softmax", False)), orig(*a_, **k))[1]
        try:
            out = model(x, g)
        finally:
            S.spmm_csr = Fz.spmm_csr = orig
        assert any(seen), "the fused log_softmax epilogue was not used"
        torch.nn.functional.nll_loss(out[idx], lab).backward()
        got = [p.grad.clone() for p in model.parameters()]
        model.zero_grad()
        h = model.gc1(x, g, relu=True)
        ref_out = torch.log_softmax(model.gc2(h, g), 1)
        torch.nn.functional.nll_loss(ref_out[idx], lab).backward()
        assert_normwise(out.detach().cpu(), ref_out.detach().cpu().numpy(), TOL, "fused output")
        for a_, p in zip(got, model.parameters()):
            assert_normwise(a_.cpu(), p.grad.cpu().numpy(), TOL, "fused gradients")


def test_dropout_under_hipgraph_replay_draws_fresh_masks(oracle, dev):
    """A host seed would be frozen into a captured launch; the device-resident seed
    (gcn_epilogue.seed_dev) is advanced by an op recorded in the same capture, so each replay
    masks differently — and a launch with a tensor seed equals the launch with that value as a
    host seed."""
    from pygcn_amd import GraphConvolution, spmm_csr
    a = _skewed_csr(oracle, 1500, 1500, 6, seed=91, hubs=((4, 600),))
    g = _graph(a, dev)
    g.plan()
    B = torch.randn(1500, 64, device=dev)
    bias = torch.rand(64, device=dev)
    seed_t = torch.tensor([123456789012345], dtype=torch.int64, device=dev)
    assert torch.equal(spmm_csr(g, B, bias=bias, relu=True, dropout_p=0.5, seed=seed_t),
                       spmm_csr(g, B, bias=bias, relu=True, dropout_p=0.5, seed=123456789012345))
    with pytest.raises(RuntimeError, match="tensor seed"):
        spmm_csr(g, B, relu=True, dropout_p=0.5, seed=seed_t.int())

    torch.manual_seed(0)
    layer = GraphConvolution(64, 64).to(dev)
    x = torch.randn(1500, 64, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(2):
            layer(x, g, relu=True, dropout=0.5)      # eager warm-up creates the device seed
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), torch.no_grad():
        out = layer(x, g, relu=True, dropout=0.5)
    graph.replay()
    first = out.clone()
    graph.replay()
    second = out.clone()
    dense = layer(x, g, relu=True).detach()
    for o in (first, second):                         # each replay is a valid inverted dropout
        kept = o != 0
        assert torch.allclose(o[kept], 2.0 * dense[kept], rtol=1e-6, atol=0)
        frac = float(kept.sum()) / float((dense != 0).sum())
        assert 0.45 < frac < 0.55
    assert not torch.equal(first != 0, second != 0), "replays reused the same dropout mask"


def test_randomized_option_combinations(oracle, dev):
    """60 seeded random problems over the OPTIONS of gcn_spmm_csr_ep — operand hint, output row
    flags, fused log_softmax, two-block operand, bf16, int64 row pointers — crossed with random
    shapes and schedule knobs, each checked against the oracle product in fp64."""
    from pygcn_amd import CSRGraph, spmm_csr
    from pygcn_amd.spmm import log_softmax_fusable, row_bitmap
    rng = np.random.default_rng(77)
    for case in range(60):
        n_rows, n_cols = int(rng.integers(1, 600)), int(rng.integers(2, 600))
        bf16 = bool(rng.integers(0, 4) == 0)
        F = int(rng.choice([8, 16, 64, 128, 256, 512] if bf16 else [1, 4, 7, 33, 64, 100, 128, 256, 300]))
        deg = rng.poisson(rng.choice([0.5, 3, 10]), size=n_rows)
        for _ in range(int(rng.integers(0, 3))):
            deg[rng.integers(0, n_rows)] = int(rng.integers(50, 1200))
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        nnz = int(rowptr[-1])
        col = rng.integers(0, n_cols, size=nnz).astype(np.int32)
        val = (1.0 - rng.random(nnz)).astype(np.float32)
        a = oracle.CSR(rowptr, col, val, (n_rows, n_cols))
        kw = dict(item_cost=int(rng.choice([0, 8, 200])), long_thresh=int(rng.choice([0, 4, 37])))
        rp_t = torch.from_numpy(rowptr if case % 2 else rowptr.astype(np.int32))
        g = CSRGraph(rp_t.to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev),
                     (n_rows, n_cols), **kw)
        dt = torch.bfloat16 if bf16 else torch.float32
        B = torch.from_numpy(gin.dense((n_cols, F), 9000 + case)).to(dev)
        B = (B * (torch.rand(n_cols, 1, device=dev) < rng.choice([0.05, 0.5, 1.0]))).to(dt)
        bias = torch.from_numpy(gin.dense((F,), 9500 + case)).to(dev) if case % 3 else None
        use_ls = bool(rng.integers(0, 2)) and log_softmax_fusable(F, dt)
        use_hint, use_flags, use_b2 = (bool(rng.integers(0, 2)) for _ in range(3))
        opts = {}
        if use_hint:
            opts["b_hint"] = row_bitmap(B)
        flags = torch.zeros(n_rows, dtype=torch.uint8, device=dev) if use_flags else None
        lhs = B
        if use_b2:
            split = int(rng.integers(0, n_cols + 1))
            lhs, opts["B2"] = B[:split], B[split:].clone()
        out = spmm_csr(g, lhs, bias=bias, c_flags=flags, log_softmax=use_ls, **opts)
        ref = a.matmul(B.float().cpu().numpy()).astype(np.float64)
        if bias is not None:
            ref = ref + bias.cpu().numpy().astype(np.float64)
        scale = max(float(np.abs(ref).max()) if ref.size else 0.0, 1e-30)
        if use_ls:
            ref = ref - ref.max(1, keepdims=True)
            ref = ref - np.log(np.exp(ref).sum(1, keepdims=True))
            scale = max(scale, float(np.abs(ref).max()))
        what = f"case {case}: {n_rows}x{n_cols} F={F} {dt} {kw} ls={use_ls} hint={use_hint} " \
               f"flags={use_flags} b2={use_b2}"
        err = float(np.abs(out.float().cpu().numpy().astype(np.float64) - ref).max()) if ref.size else 0.0
        assert err <= (2.0 ** -7 if bf16 else 1e-5) * scale, f"{what}: err {err:.3e} scale {scale:.3e}"
        if use_flags:
            assert torch.equal(flags.bool(), (out != 0).any(1)), what


@pytest.mark.parametrize("F,dtype,idx64", [(256, torch.float32, False), (600, torch.float32, True),
                                           (512, torch.bfloat16, False), (64, torch.float32, False),
                                           (16, torch.float32, True), (7, torch.float32, False),
                                           (128, torch.bfloat16, False)])
@pytest.mark.parametrize("density", [0.0, 0.1, 0.6, 1.0])
def test_output_row_selection(oracle, dev, F, dtype, idx64, density):
    """gcn_epilogue.c_row_select: rows whose bit is set equal the unrestricted product (bitwise
    in the wide kernel: same entries, same order); the other rows may hold anything and are not
    looked at.  Long rows (chunk slab + reduce), empty rows, rows around the tile boundary and
    the combination with the operand hint and the output flags are all in the case; F = 64 / 16 / 7
    take the narrow kernel (row-per-lane-group, medium-row and one-row-at-a-time paths)."""
    from pygcn_amd import spmm_csr
    from pygcn_amd.spmm import pack_row_flags, row_bitmap
    a = _skewed_csr(oracle, 2500, 2200, 6, seed=F + 31, empties=100,
                    hubs=((3, 1200), (900, 300), (901, 40), (1700, 64), (1701, 65), (1702, 63)))
    rp = a.rowptr.astype(np.int64 if idx64 else np.int32)
    from pygcn_amd import CSRGraph
    g = CSRGraph(torch.from_numpy(rp).to(dev), torch.from_numpy(a.col).to(dev),
                 torch.from_numpy(a.val).to(dev), a.shape)
    gen = torch.Generator(device=dev).manual_seed(F)
    B = torch.randn(2200, F, generator=gen, device=dev).to(dtype)
    want = torch.rand(2500, generator=gen, device=dev) < density
    if 0.0 < density < 1.0:
        want[3], want[900], want[1701] = True, False, True       # one long row in, one out
    bits, _ = pack_row_flags(want)
    full = spmm_csr(g, B)
    out = torch.full((2500, F), float("nan"), device=dev).to(dtype)
    spmm_csr(g, B, out=out, c_select=bits)
    assert torch.equal(out[want], full[want])
    if density < 1.0:                  # unwanted rows were really skipped (both kernels)
        assert bool(out[~want].isnan().all())
    # together with a row-sparse operand and the output flags
    Bs = B * (torch.rand(2200, 1, generator=gen, device=dev) < 0.1)
    flags = torch.zeros(2500, dtype=torch.uint8, device=dev)
    out2 = torch.zeros((2500, F), device=dev).to(dtype)
    spmm_csr(g, Bs, out=out2, c_select=bits, b_hint=row_bitmap(Bs), c_flags=flags)
    ref2 = spmm_csr(g, Bs)
    assert_normwise(out2[want].float().cpu(), ref2[want].float().cpu().numpy(), 1e-6, "select + hint")
    assert torch.equal(flags.bool()[want], (ref2 != 0).any(1)[want])
    with pytest.raises(RuntimeError, match="c_select"):
        spmm_csr(g, B, c_select=bits[:-1])


def test_layer_output_carries_its_maximum_to_the_next_layer(dev, h2_scheme):
    """The GEMM that ends a reassociated layer leaves max|out| for the tensor object it returns;
    the next layer's scaled GEMM picks it up instead of reducing over [N, 256] again.  The record is
    bound to the object AND its version."""
    import importlib
    from pygcn_amd import CSRGraph, GraphConvolution
    from pygcn_amd.utils import rmat_graph
    S = importlib.import_module("pygcn_amd.spmm")
    n = 20000
    rowptr, col, val = rmat_graph(n, 8 * n, seed=3, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    torch.manual_seed(0)
    layer = GraphConvolution(256, 256).to(dev)
    x = torch.randn(n, 256, device=dev)
    h = layer(x, g, relu=True)
    known = S.known_absmax(h)
    assert known is not None and known.item() == h.detach().abs().max().item()
    assert S.known_absmax(h.detach().clone()) is None
    with torch.no_grad():
        h.add_(1.0)
    assert S.known_absmax(h) is None


@pytest.mark.parametrize("width,how", [(256, "reassociated"), (128, "row-selected")])
def test_first_layer_weight_gradient_without_a_transpose_product(oracle, dev, width, how, gemm_scheme):
    """A first layer (its input needs no gradient) under a row-sparse grad_pre forms
    grad_W = (A·X)ᵀ·grad_pre: at 256 -> 256 fp32 the layer is evaluated as (A·X)·W and keeps A·X
    from its forward pass (no sparse product in backward at all); other widths run a forward
    product restricted to the rows grad_pre is non-zero on.  Same gradients as the
    transpose-product path (taken with row compaction off and the "exact" GEMM scheme — hipBLASLt,
    the reference's order Â·(X·W) — which switches the reassociation off)."""
    import importlib
    from pygcn_amd import CSRGraph, GraphConvolution
    from pygcn_amd.utils import rmat_graph
    S = importlib.import_module("pygcn_amd.spmm")
    n = 1 << 18
    rowptr, col, val = rmat_graph(n, 4 * n, seed=6, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    torch.manual_seed(0)
    layer = GraphConvolution(width, width).to(dev)
    x = torch.randn(n, width, device=dev)
    idx = torch.randperm(n, device=dev)[: n // 20]
    tgt = torch.randn(idx.numel(), width, device=dev)

    def grads(compaction, scheme):
        S.set_row_compaction(compaction)
        S.set_gemm_scheme(scheme)
        S._poison_unwritten = True
        try:
            layer.zero_grad()
            seen = []
            orig = S.spmm_csr
            S.spmm_csr = lambda *a_, **k: (seen.append((k.get("tag", "fwd"), k.get("c_select") is not None)),
                                           orig(*a_, **k))[1]
            try:
                torch.manual_seed(1)
                ((layer(x, g, relu=True)[idx] - tgt) ** 2).sum().backward()
            finally:
                S.spmm_csr = orig
            return [p.grad.clone() for p in layer.parameters()], seen
        finally:
            S.set_row_compaction(False)
            S.set_gemm_scheme(gemm_scheme)
            S._poison_unwritten = False

    new, seen = grads(True, gemm_scheme)
    ref, seen_ref = grads(False, "exact")
    backward = [sel for tag, sel in seen if tag == "bwd"]
    assert backward == ([] if how == "reassociated" else [True]), seen
    assert [sel for tag, sel in seen_ref if tag == "bwd"] == [False], seen_ref
    for a_, b_ in zip(new, ref):
        assert_normwise(a_.cpu(), b_.cpu().numpy(), 2e-5, how + " grad")     # (two float32 routes, each within 1e-5 of exact)


@pytest.mark.parametrize("F,dtype", [(256, torch.float32), (64, torch.float32), (128, torch.bfloat16)])
def test_backward_passes_skip_zero_rows(dev, F, dtype):
    """skip_zero_rows of the fused backward passes: the rows flagged in the bitmap equal the
    full result, every other row is left untouched (NaN pre-fill survives), column sums, bitmap
    and count are unchanged."""
    import importlib
    S = importlib.import_module("pygcn_amd.spmm")
    n = 6000
    gen = torch.Generator(device=dev).manual_seed(F + 5)
    g = (torch.randn(n, F, generator=gen, device=dev)
         * (torch.rand(n, 1, generator=gen, device=dev) < 0.07)).to(dtype)
    out_relu = torch.relu(torch.randn(n, F, generator=gen, device=dev)).to(dtype)
    logp = torch.log_softmax(torch.randn(n, F, generator=gen, device=dev), 1).to(dtype)
    for out, kw in ((out_relu, dict(scale=2.0)), (logp, dict(log_softmax=True))):
        full, cs, hint = S.backward_with_colsum(g, out, **kw)
        S._poison_unwritten = True
        try:
            part, cs2, hint2 = S.backward_with_colsum(g, out, skip_zero_rows=True, **kw)
        finally:
            S._poison_unwritten = False
        keep = S.unpack_row_flags(hint[0], n)
        assert torch.equal(hint2[0], hint[0]) and int(hint2[1]) == int(hint[1]) == int(keep.sum())
        assert torch.equal(cs2, cs)
        assert torch.equal(part[keep], full[keep])
        assert bool(part[~keep].isnan().all()) and 0 < int(keep.sum()) < n
        assert bool((full[~keep] == 0).all())


@pytest.mark.parametrize("fout,share", [(256, 0.5), (256, 0.9), (64, 0.05), (64, 0.3), (64, 0.9)])
def test_unwritten_rows_are_never_read(dev, fout, share):
    """The backward pass leaves all-zero gradient rows unwritten only as long as the product that
    follows skips them (device-side rule: < 3/4 non-zero rows in the wide kernel, < 1/8 in the
    narrow one — mirrored on the host by _hint_will_be_used); above the threshold they must get
    real zeros first.  With the NaN pre-fill every mistake shows as a NaN gradient."""
    import importlib
    from pygcn_amd import CSRGraph, GraphConvolution
    from pygcn_amd.utils import rmat_graph
    S = importlib.import_module("pygcn_amd.spmm")
    n = 1 << 17
    rowptr, col, val = rmat_graph(n, 3 * n, seed=11, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    torch.manual_seed(0)
    layer = GraphConvolution(32, fout).to(dev)
    x = torch.randn(n, 32, device=dev, requires_grad=True)      # general path (grad_input needed)
    rows = torch.nonzero(torch.rand(n, device=dev) < share).squeeze(1)
    w = torch.randn(rows.numel(), fout, device=dev)

    def run(poison, compaction):
        S._poison_unwritten, _ = poison, S.set_row_compaction(compaction)
        try:
            layer.zero_grad()
            x.grad = None
            (layer(x, g, relu=True)[rows] * w).sum().backward()
            return [x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
        finally:
            S._poison_unwritten = False
            S.set_row_compaction(True)

    got, ref = run(True, True), run(False, False)
    for a_, b_ in zip(got, ref):
        assert bool(a_.isfinite().all())
        assert_normwise(a_.cpu(), b_.cpu().numpy(), TOL, "skip-write path vs plain path")


def test_colsum_pass_refuses_row_bitmap_for_rows_wider_than_a_wavefront(dev):
    """with F / lane width > 64 a row spans several wavefronts, so the one-pass backward
    cannot produce the row bitmap — the C-ABI must say so (GCN_E_BADARG), never return success with
    the bitmap and the count left unwritten."""
    import ctypes
    from pygcn_amd import _native
    L = _native.lib()
    n, F = 300, 512                                   # fp32: F / 4 = 128 lanes per row
    g = torch.randn(n, F, device=dev)
    o = torch.randn(n, F, device=dev)
    res = torch.empty_like(g)
    colsum = torch.empty(F, device=dev)
    bits = torch.full(((n + 31) // 32,), -1, dtype=torch.int32, device=dev)
    cnt = torch.full((1,), -7, dtype=torch.int32, device=dev)
    ws_bytes = L.gcn_bwd_colsum_workspace_bytes(n, F, _native.GCN_DTYPE_F32)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    rc = L.gcn_relu_dropout_backward_colsum(_native.GCN_DTYPE_F32, g.data_ptr(), o.data_ptr(),
                                            res.data_ptr(), colsum.data_ptr(), n, F, 1.0,
                                            bits.data_ptr(), cnt.data_ptr(), 0, ws.data_ptr(),
                                            ws_bytes, stream)
    assert rc == -1 and b"lane width <= 64" in L.gcn_last_error()
    torch.cuda.synchronize()
    assert int(cnt) == -7 and bool((bits == -1).all())          # untouched, and the caller was told
    # without the optional outputs the same shape is fine
    rc = L.gcn_relu_dropout_backward_colsum(_native.GCN_DTYPE_F32, g.data_ptr(), o.data_ptr(),
                                            res.data_ptr(), colsum.data_ptr(), n, F, 1.0, None, None,
                                            0, ws.data_ptr(), ws_bytes, stream)
    assert rc == 0
    want = torch.where(o > 0, g, torch.zeros_like(g))
    assert torch.equal(res, want) and torch.allclose(colsum, want.sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("F,dtype,p", [(256, torch.float32, 0.5), (256, torch.float32, 0.25), (64, torch.float32, 0.1),
                                       (128, torch.bfloat16, 0.3), (128, torch.bfloat16, 0.5)])
def test_dropout_row_base_gives_a_shard_the_masks_of_the_whole(oracle, dev, F, dtype, p):
    """ABI 22: `drop_row_base` is added to the row index in the dropout counter — a row-block shard
    that passes its first global row draws exactly the masks the single-GPU run draws for those
    rows (SpMM epilogue and both GEMM epilogues; oracle.dropout_keep restates the function)."""
    from pygcn_amd import CSRGraph, spmm_csr
    from pygcn_amd.spmm import gemm_bf16, gemm_xw256
    n, base, seed = 700, 123456789012, 0xC0FFEE1234
    eye = CSRGraph(torch.arange(n + 1, dtype=torch.int32, device=dev), torch.arange(n, dtype=torch.int32, device=dev),
                   torch.ones(n, device=dev), (n, n))
    ones = torch.ones(n, F, device=dev).to(dtype)
    got = spmm_csr(eye, ones, relu=True, dropout_p=p, seed=seed, row_base=base).float().cpu().numpy() != 0
    want = oracle.dropout_keep(seed, np.arange(n), F, p, row_base=base)
    np.testing.assert_array_equal(got, want)
    assert abs(want.mean() - (1 - p)) < 0.02
    assert not np.array_equal(want, oracle.dropout_keep(seed, np.arange(n), F, p))
    if dtype == torch.float32 and F == 256:
        W = torch.eye(256, device=dev)
        y = gemm_xw256(ones, W, x_bound=torch.ones(1, device=dev), relu=True, dropout_p=p, seed=seed, row_base=base)
        np.testing.assert_array_equal(y.cpu().numpy() != 0, want)
    if dtype == torch.bfloat16:
        W = torch.eye(128, device=dev).bfloat16()
        y = gemm_bf16(ones, W, relu=True, dropout_p=p, seed=seed, row_base=base)
        np.testing.assert_array_equal(y.float().cpu().numpy() != 0, want)


@pytest.mark.parametrize("F,dtype", [(256, torch.float32), (64, torch.float32), (7, torch.float32),
                                     (128, torch.bfloat16)])
def test_product_reports_its_own_absmax(oracle, dev, F, dtype):
    """gcn_epilogue.c_absmax: the launch reports max|stored value| (every kernel family: wide, narrow
    vector / scalar, bf16, long rows incl. their reduce kernel, with and without a fused epilogue)
    — exactly the maximum of the tensor it wrote; inf / NaN in the result show up as non-finite."""
    from pygcn_amd import spmm_csr
    n = 3000
    a = _skewed_csr(oracle, n, n, 6, seed=15, hubs=((3, 1500), (900, 700)), empties=40)
    g = _graph(a, dev)
    B = (torch.from_numpy(gin.dense((n, F), 21)) * 3.0).to(dtype).to(dev)
    bias = torch.from_numpy(gin.dense((F,), 22)).to(dev)
    for kw in ({}, {"bias": bias}, {"bias": bias, "relu": True}, {"bias": bias, "log_softmax": True}):
        if kw.get("log_softmax") and F == 7 and False:
            continue
        m = torch.zeros(1, device=dev)
        out = spmm_csr(g, B, c_absmax=m, **kw)
        assert m.item() == out.float().abs().max().item(), (kw.keys(), m.item(), out.float().abs().max().item())
        assert torch.equal(out, spmm_csr(g, B, **kw))             # the side output changes nothing
    if dtype == torch.float32:
        Bn = B.clone()
        Bn[5, 0] = float("inf")
        m = torch.zeros(1, device=dev)
        spmm_csr(g, Bn, c_absmax=m)
        assert not torch.isfinite(m).all()


@pytest.mark.parametrize("F,dtype", [(256, torch.float32), (64, torch.float32), (7, torch.float32),
                                     (300, torch.float32), (128, torch.bfloat16)])
def test_sddmm_on_the_adjacency_pattern(oracle, dev, F, dtype):
    """the gradient of the adjacency VALUES, grad_val[e] = <G[row(e)],
    B[col[e]]> on A's pattern (gcn_sddmm_csr) — PyTorch's `mm` derivative for a sparse first
    operand.  Against float64 numpy on the same (bf16-rounded) operands: rows of 1 ... 1500 entries
    incl. the chunked long-row work units, empty rows, vector and scalar element paths."""
    from pygcn_amd.spmm import sddmm_csr
    n, m = 2500, 1800
    a = _skewed_csr(oracle, n, m, 6, seed=25, hubs=((3, 1500), (900, 700)), empties=40)
    g = _graph(a, dev)
    G = torch.from_numpy(gin.dense((n, F), 31)).to(dtype)
    B = torch.from_numpy(gin.dense((m, F), 32)).to(dtype)
    got = sddmm_csr(g, G.to(dev), B.to(dev))
    assert got.dtype == torch.float32 and got.shape == (a.nnz,)
    row = np.repeat(np.arange(n), np.diff(a.rowptr))
    ref = (G.double().numpy()[row] * B.double().numpy()[a.col]).sum(1)
    assert_normwise(got.cpu(), ref, 1e-5, f"sddmm F={F} {dtype}")
    # the same through unaligned operands (scalar element path): a column slice with an odd pitch
    if dtype == torch.float32 and F > 8:
        Gs, Bs = G.to(dev)[:, 1:F - 2], B.to(dev)[:, 1:F - 2]
        ref2 = (G.double().numpy()[row][:, 1:F - 2] * B.double().numpy()[a.col][:, 1:F - 2]).sum(1)
        assert_normwise(sddmm_csr(g, Gs, Bs).cpu(), ref2, 1e-5, "sddmm, unaligned slices")


@pytest.mark.parametrize("F,dtype,m", [(256, torch.float32, 5000), (32, torch.float32, 777), (288, torch.float32, 1001),
                                       (1024, torch.float32, 300), (128, torch.bfloat16, 4099),
                                       (544, torch.bfloat16, 65), (256, torch.float32, 0)])
def test_rows_pack_unpack_round_trip(dev, F, dtype, m):
    """Wire format of the compressed halo exchange (gcn_rows_pack_count / _values, gcn_rows_unpack,
    gcn_bits_row_counts): bit-exact against the torch-op statement of the same format
    (pygcn_amd.sharded.pack_bits) and a bit-exact round trip, with a row list, a padded leading
    dimension, -0.0 (travels as zero), NaN / inf (travel as values) and all-zero / full rows."""
    from pygcn_amd.sharded import pack_bits
    from pygcn_amd.spmm import rows_pack, rows_unpack
    g = torch.Generator(device="cpu").manual_seed(F + m)
    n = 2 * m + 3
    wide = torch.zeros((n, F + 8), dtype=dtype, device=dev)
    src = wide[:, :F]                                            # leading dimension F + 8
    vals = torch.randn((n, F), generator=g) * (torch.rand((n, F), generator=g) < 0.3)
    vals[0].zero_()
    vals[1] = 1.0
    vals[2, ::3] = -0.0
    vals[2, 1] = float("nan")
    vals[2, 4] = float("inf")
    src.copy_(vals.to(dtype))
    for rows in (None, torch.randperm(n, generator=g)[:m].to(dev)):
        want = src if rows is None else src[rows]
        bits, offsets, packed = rows_pack(src, rows)
        mask = want != 0
        assert torch.equal(bits, pack_bits(mask))
        assert torch.equal(offsets[1:], mask.sum(1).cumsum(0)) and int(offsets[0]) == 0
        ref_vals = want[mask]
        assert packed.dtype == dtype and packed.shape == ref_vals.shape
        assert torch.equal(packed.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                           ref_vals.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))
        back = rows_unpack(bits, packed, F)
        expect = torch.where(mask, want, torch.zeros_like(want))  # (-0.0 arrives as +0.0)
        it = torch.int16 if dtype == torch.bfloat16 else torch.int32
        assert back.shape == expect.shape and torch.equal(back.contiguous().view(it), expect.contiguous().view(it))


def test_rows_pack_rejects_unsupported_shapes(dev):
    from pygcn_amd.spmm import rows_pack, rows_unpack
    with pytest.raises(RuntimeError):
        rows_pack(torch.zeros((4, 48), device=dev))              # width not a multiple of 32
    with pytest.raises(RuntimeError):
        rows_pack(torch.zeros((4, 64), device=dev, dtype=torch.float16))
    with pytest.raises(RuntimeError):
        rows_pack(torch.zeros((4, 64)))                          # host tensor: no CPU path
    with pytest.raises(RuntimeError):
        rows_unpack(torch.zeros((4, 2), dtype=torch.int32, device=dev), torch.zeros(0, device=dev), 96)


def test_long_row_chunk_length_follows_the_storage_type(oracle, dev):
    """A graph built without an explicit `long_thresh` chunks long rows at the C-ABI default (256:
    the sequential fp32 chain of a chunk bounds the rounding error of the 1e-5 contract) for fp32
    operands and at tuning.LONG_THRESH_BF16 (1 024) for bf16 storage — a second cached schedule;
    an explicit `long_thresh` is honoured for every type.  Same bits as a graph built with that
    chunk length explicitly; both chunk lengths agree to bf16 resolution."""
    from pygcn_amd import spmm_csr, tuning, _native
    n, F = 4000, 128
    a = _skewed_csr(oracle, n, n, 6, seed=41, hubs=((5, 3000), (1700, 1500), (2500, 700)), empties=30)
    g = _graph(a, dev)
    assert g.plan().long_thresh == _native.GCN_DEFAULT_LONG_THRESH == 256
    assert g.plan(torch.float32).long_thresh == 256
    pb = g.plan(torch.bfloat16)
    assert pb.long_thresh == tuning.LONG_THRESH_BF16 == 1024 and pb is g.plan(torch.bfloat16)
    assert 0 < pb.n_long < g.plan().n_long and pb.n_chunks < g.plan().n_chunks
    B = torch.from_numpy(gin.dense((n, F), 7)).to(dev)
    bias = torch.from_numpy(gin.dense((F,), 8)).to(dev)
    got = spmm_csr(g, B.bfloat16(), bias=bias, log_softmax=True)
    g1024, g256 = _graph(a, dev, long_thresh=1024), _graph(a, dev, long_thresh=256)
    assert g256.plan(torch.bfloat16).long_thresh == 256
    assert torch.equal(got, spmm_csr(g1024, B.bfloat16(), bias=bias, log_softmax=True))
    other = spmm_csr(g256, B.bfloat16(), bias=bias, log_softmax=True)
    assert (got.float() - other.float()).abs().max().item() <= 2.0 ** -7 * other.float().abs().max().item()
    ref, _ = oracle.gc_forward(B.bfloat16().float().cpu().numpy(), np.eye(F, dtype=np.float32), bias.cpu().numpy(), a)
    ref = ref - np.log(np.exp(ref - ref.max(1, keepdims=True)).sum(1, keepdims=True)) - ref.max(1, keepdims=True)
    assert_normwise(got.float().cpu(), ref, 2.0 ** -7, "bf16 product with 1 024-entry chunks")
    # fp32 keeps the default schedule
    assert torch.equal(spmm_csr(g, B), spmm_csr(g256, B))
