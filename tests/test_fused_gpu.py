"""`model(x, adj, rows=idx)` — the 2-layer training step as one autograd node (pygcn_amd/fused.py)
— against the oracle's forward / loss / backward (reference semantics: pygcn/train.py:153-157 with
the upstream model of models.py:23,48,50,68).  Tensors whose rows are left unwritten on purpose
are pre-filled with NaN in these tests, so reading a row that must not be read poisons the result."""
import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_normwise, assert_parity, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, params=["bf16x3", "h2"])
def scheme(request):
    """Every test of this module runs under both decompositions of the 256-wide fp32 GEMMs: the
    default fp32-equivalent three-part bf16 scheme and the opt-in scaled two-part fp16 scheme."""
    from pygcn_amd import spmm as S
    before = S.gemm_scheme()
    S.set_gemm_scheme(request.param)
    yield request.param
    S.set_gemm_scheme(before)


# Route-vs-route comparisons (two float32 evaluations of the same gradient in different summation
# orders, measured <= 3.7e-6 in round 4: profiles/r04_parity_ledger.md): the contract itself.
ROUTES = 1e-5


@pytest.fixture()
def poison():
    from pygcn_amd import spmm as S
    S._poison_unwritten = True
    yield
    S._poison_unwritten = False


def _check(model, x, adj_graph, a, labels, idx, oracle, need_x=False):
    dev = x.device
    idx_t = torch.from_numpy(np.asarray(idx)).to(dev)
    model.train()
    model.zero_grad()
    out_rows, full = model(x, adj_graph, rows=idx_t, keep_full=True)
    loss = torch.nn.functional.nll_loss(out_rows, torch.from_numpy(labels).to(dev)[idx_t])
    loss.backward()
    p = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ref_loss, fw, grads, extra = oracle.gcn2_loss_backward(x.detach().cpu().numpy(), a, p, labels,
                                                           np.asarray(idx), need_grad_x=need_x)
    assert not full.requires_grad
    assert_normwise(full.cpu(), fw["logp"], TOL, "full log-probabilities")
    assert_normwise(out_rows.detach().cpu(), fw["logp"][np.asarray(idx)], TOL, "selected rows")
    assert abs(loss.item() - ref_loss) <= TOL * abs(ref_loss)
    # parameter gradients are float32 reductions over the graph's vertices: float64 arbiter
    # (conftest.assert_parity), the float64 step given the float32 oracle's ReLU pattern
    _, _, grads64 = oracle.gcn2_loss_backward_f64(x.detach().cpu().numpy(), a, p, labels, np.asarray(idx),
                                                  relu_mask=fw["h1"] > 0)
    for k, v in grads.items():
        mod, name = k.split(".")
        got = getattr(getattr(model, mod), name).grad
        assert got is not None and torch.isfinite(got).all(), k
        assert_parity(got.cpu(), v, grads64[k], k + ".grad")
    if need_x:
        assert_normwise(x.grad.cpu(), extra["grad_x"], TOL, "grad_x")
    kept = {k: q.grad.detach().clone() for k, q in model.named_parameters()}
    # and it is the same function as the plain call + indexing
    model.zero_grad()
    plain = model(x.detach(), adj_graph)
    loss2 = torch.nn.functional.nll_loss(plain[idx_t], torch.from_numpy(labels).to(dev)[idx_t])
    assert abs(loss2.item() - loss.item()) <= 1e-6 * abs(loss.item())
    return kept


def test_cora_step_through_the_one_node_path(oracle, dev, poison):
    """Config C2 shapes (1433 -> 16 -> 7): wide first layer, so layer 1 takes the transpose-product
    branch; adjacency handed over as the torch sparse COO tensor the reference builds."""
    from pygcn_amd import GCN
    from pygcn_amd.utils import load_data
    adj, _, _, idx_train, _, _ = load_data()
    g = np.load(gin.__file__.replace("inputs.py", "cora_graph.npz"))
    a = oracle.cora_adjacency(g["edges"], int(g["n"]))
    x = torch.from_numpy(gin.cora_features()).to(dev)
    torch.manual_seed(42)
    model = GCN(1433, 16, 7, dropout=0.0).to(dev)
    got = _check(model, x, adj.to(dev), a, gin.cora_labels(), idx_train.numpy(), oracle)
    # ... and against G2, the same step captured from the imported reference layer (seed-42
    # parameters = G1, same features / labels / idx_train): all four parameter gradients
    g2 = load_golden("g2_cora_step.npz")
    for mod, name in (("gc1", "weight"), ("gc1", "bias"), ("gc2", "weight"), ("gc2", "bias")):
        assert_normwise(got[f"{mod}.{name}"].cpu(), g2[f"{mod}_{name}_grad"], TOL, f"G2 {mod}_{name}_grad")


@pytest.mark.parametrize("fin,hid,ncls,share", [(256, 256, 256, 0.05), (48, 64, 16, 0.3), (256, 256, 64, 0.9),
                                                (700, 32, 8, 0.02)])
def test_rmat_step_matches_oracle(oracle, dev, poison, fin, hid, ncls, share):
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.utils import rmat_graph
    n = 30000
    rowptr, col, val = rmat_graph(n, 300000, seed=21, device="cpu")
    a = oracle.CSR(rowptr.numpy().astype(np.int64), col.numpy(), val.numpy(), (n, n))
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    rng = np.random.default_rng(fin + ncls)
    x = torch.from_numpy(gin.dense((n, fin), 5)).to(dev)
    labels = rng.integers(0, ncls, n)
    idx = rng.permutation(n)[: int(n * share)]            # unsorted on purpose
    torch.manual_seed(1)
    model = GCN(fin, hid, ncls, dropout=0.0).to(dev)
    _check(model, x, g, a, labels, idx, oracle)


@pytest.mark.parametrize("start,count", [(0, 1500), (1234, 4000), (29000, 1000)])
def test_loss_rows_that_are_a_range(oracle, dev, poison, start, count):
    """upstream's idx_train is `range(140)` (utils.py:370): a range at the start, in the middle and
    at the end of the vertex list (the block of Âᵀ the backward product runs on is cut per row set)."""
    from pygcn_amd import GCN, CSRGraph, fused
    from pygcn_amd.utils import rmat_graph
    n = 30000
    rowptr, col, val = rmat_graph(n, 300000, seed=22, device="cpu")
    a = oracle.CSR(rowptr.numpy().astype(np.int64), col.numpy(), val.numpy(), (n, n))
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    x = torch.from_numpy(gin.dense((n, 256), 6)).to(dev)
    labels = np.random.default_rng(start).integers(0, 256, n)
    idx = np.arange(start, start + count)
    rs = fused.row_sets(g, torch.from_numpy(idx).to(dev))
    assert rs.sorted_unique and rs.at_block.shape == (rs.n2, count)
    torch.manual_seed(2)
    model = GCN(256, 256, 256, dropout=0.0).to(dev)
    _check(model, x, g, a, labels, idx, oracle)


def test_duplicate_rows_input_gradient_and_dropout(oracle, dev, poison):
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd import spmm as S
    from pygcn_amd.utils import rmat_graph
    n, F = 20000, 64
    rowptr, col, val = rmat_graph(n, 150000, seed=22, device="cpu")
    a = oracle.CSR(rowptr.numpy().astype(np.int64), col.numpy(), val.numpy(), (n, n))
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    rng = np.random.default_rng(3)
    labels = rng.integers(0, F, n)
    idx = np.concatenate([rng.integers(0, n, 900), [7, 7, 7, n - 1]])      # duplicates
    torch.manual_seed(2)
    model = GCN(F, F, F, dropout=0.0).to(dev)
    x = torch.from_numpy(gin.dense((n, F), 6)).to(dev).requires_grad_(True)    # input gradient too
    _check(model, x, g, a, labels, idx, oracle, need_x=True)
    # training-mode dropout: the one-node path and the layer-by-layer path draw the same mask from
    # the same generator state, so their gradients agree (layer path = the tested reference here)
    model.dropout = 0.4
    idx_t = torch.from_numpy(idx).to(dev)
    y = torch.from_numpy(labels).to(dev)[idx_t]
    grads = []
    for fused in (True, False):
        model.zero_grad()
        torch.manual_seed(11)
        out = model(x.detach(), g, rows=idx_t) if fused else model(x.detach(), g)[idx_t]
        torch.nn.functional.nll_loss(out, y).backward()
        grads.append([p.grad.clone() for p in model.parameters()])
    for p, q in zip(*grads):
        assert_normwise(p.cpu(), q.cpu().numpy(), ROUTES, "dropout: one node vs layers")


def test_bf16_one_node_path(oracle, dev):
    """C5 numerics (bf16 storage) through the one-node path, against the oracle on the bf16-rounded
    parameters / inputs (same gates as test_bf16_model_end_to_end)."""
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.utils import rmat_graph
    n, F = 20000, 128
    rowptr, col, val = rmat_graph(n, 200000, seed=3, device="cpu")
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    a = oracle.CSR(rowptr.numpy().astype(np.int64), col.numpy(), val.numpy(), (n, n))
    x16 = torch.from_numpy(gin.dense((n, F), 1)).to(torch.bfloat16)
    y = np.random.default_rng(2).integers(0, F, n)
    idx = np.arange(n // 10)
    torch.manual_seed(5)
    m16 = GCN(F, F, F, dropout=0.0).to(torch.bfloat16).to(dev)
    idx_t = torch.from_numpy(idx).to(dev)
    out = m16(x16.to(dev), g, rows=idx_t)
    assert out.dtype == torch.bfloat16 and out.shape == (len(idx), F)
    loss = torch.nn.functional.nll_loss(out.float(), torch.from_numpy(y).to(dev)[idx_t])
    loss.backward()
    p = {k: v.detach().float().cpu().numpy() for k, v in m16.state_dict().items()}
    ref_loss, fw, grads, _ = oracle.gcn2_loss_backward(x16.float().numpy(), a, p, y, idx)
    assert_normwise(out.float().detach().cpu(), fw["logp"][idx], 2.0 ** -6, "logp rows")
    assert abs(loss.item() - ref_loss) <= 2.0 ** -6 * abs(ref_loss)
    for k, v in grads.items():
        mod, name = k.split(".")
        assert_normwise(getattr(getattr(m16, mod), name).grad.float().cpu(), v, 2.0 ** -4, k + ".grad")


def test_backward_pass_has_no_host_synchronisation(dev, monkeypatch):
    """After the row sets of (graph, rows) exist, a training step through the one-node path reads
    nothing back to the host — so it can be captured into a hipGraph."""
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.utils import rmat_graph
    n, F = 200000, 256                      # >= MIN_ROWS: the layer-by-layer path would synchronise
    rowptr, col, val = rmat_graph(n, 2000000, seed=4, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    g.plan(), g.t().plan()
    x = torch.randn(n, F, device=dev)
    y = torch.randint(0, F, (n,), device=dev)
    idx = torch.arange(n // 20, device=dev)
    y_idx = y[idx]
    model = GCN(F, F, F, dropout=0.5).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.nll_loss(model(x, g, rows=idx), y_idx)
        loss.backward()
        opt.step()
        return loss
    step()                                   # builds the row sets (one-off host reads)
    torch.cuda.synchronize()
    reads = []
    for name in ("item", "tolist", "cpu"):
        real = getattr(torch.Tensor, name)
        monkeypatch.setattr(torch.Tensor, name,
                            (lambda r, nm: lambda t, *a, **k: (reads.append(nm) if t.is_cuda else None,
                                                               r(t, *a, **k))[1])(real, name))
    real_nonzero = torch.nonzero
    monkeypatch.setattr(torch, "nonzero", lambda *a, **k: (reads.append("nonzero"), real_nonzero(*a, **k))[1])
    l1 = step()
    monkeypatch.undo()
    assert reads == [], reads
    assert torch.isfinite(l1).item()


def test_upstream_lines_take_the_row_sparse_route(oracle, dev, poison):
    """`output = model(x, adj); loss = nll_loss(output[idx], labels[idx])` — upstream's lines,
    no `rows=`: the selection hands the layers a RowGrad (pygcn_amd/rowgrad.py) and the backward
    pass runs on the transpose block and compact rows.  Same gradients as the oracle, as the
    `rows=` path, and as the dense route (taken when the index is not a tensor)."""
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.rowgrad import RowGrad, RowSelectable
    from pygcn_amd.utils import rmat_graph
    import pygcn_amd.spmm as S
    n = 30000
    rowptr, col, val = rmat_graph(n, 300000, seed=23, device="cpu")
    a = oracle.CSR(rowptr.numpy().astype(np.int64), col.numpy(), val.numpy(), (n, n))
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    x = torch.from_numpy(gin.dense((n, 256), 8)).to(dev)
    labels_np = np.random.default_rng(3).integers(0, 256, n)
    labels = torch.from_numpy(labels_np).to(dev)
    idx_np = np.random.default_rng(4).permutation(n)[: n // 12]               # unsorted
    idx = torch.from_numpy(idx_np).to(dev)
    torch.manual_seed(5)
    model = GCN(256, 256, 256, dropout=0.0).to(dev)
    model.train()
    import pygcn_amd.fused as Fz
    seen = []
    orig = Fz._gcn2_backward_rows
    Fz._gcn2_backward_rows = lambda ctx, *a, **k: (seen.append("rows"), orig(ctx, *a, **k))[1]
    try:
        out = model(x, g)
        assert isinstance(out, RowSelectable) and not isinstance(out[idx], RowSelectable)
        assert not isinstance(out + 1, RowSelectable) and torch.equal(out[idx], out.as_subclass(torch.Tensor)[idx])
        loss = torch.nn.functional.nll_loss(out[idx], labels[idx])
        loss.backward()
    finally:
        Fz._gcn2_backward_rows = orig
    assert seen == ["rows"]                              # the model's one node took the row route
    got = {k: p.grad.clone() for k, p in model.named_parameters()}
    p = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ref_loss, _, grads, _ = oracle.gcn2_loss_backward(x.cpu().numpy(), a, p, labels_np, idx_np)
    assert abs(loss.item() - ref_loss) <= TOL * abs(ref_loss)
    _, _, grads64 = oracle.gcn2_loss_backward_f64(x.cpu().numpy(), a, p, labels_np, idx_np)
    for k, v in grads.items():
        assert_parity(got[k].cpu(), v, grads64[k], "upstream lines: " + k + ".grad")
    # the dense route (a list index is not intercepted) and the rows= route agree with it
    for route in ("list", "rows"):
        model.zero_grad(set_to_none=True)
        if route == "list":
            sel = model(x, g)[idx.tolist()]
        else:
            sel = model(x, g, rows=idx)
        torch.nn.functional.nll_loss(sel, labels[idx]).backward()
        for k, q in model.named_parameters():
            assert_normwise(q.grad.cpu(), got[k].cpu().numpy(), ROUTES, route + " route: " + k)
    # a second consumer of the output: the RowGrad meets a dense gradient and is materialised
    model.zero_grad(set_to_none=True)
    out = model(x, g)
    (torch.nn.functional.nll_loss(out[idx], labels[idx]) + 1e-3 * out.mean()).backward()
    both = {k: q.grad.clone() for k, q in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    out = model(x, g).as_subclass(torch.Tensor)
    (torch.nn.functional.nll_loss(out[idx], labels[idx]) + 1e-3 * out.mean()).backward()
    for k, q in model.named_parameters():
        assert_normwise(both[k].cpu(), q.grad.cpu().numpy(), ROUTES, "two consumers: " + k)
    rg = RowGrad(torch.tensor([2, 0, 2], device=dev), torch.ones(3, 4, device=dev), 5)
    assert torch.equal(rg.dense(), torch.tensor([[1.] * 4, [0.] * 4, [2.] * 4, [0.] * 4, [0.] * 4], device=dev))
    # bare layers composed by hand (no model-level node): each layer's own node takes the RowGrad
    seen = []
    orig = S.GraphConvFunction._backward_rows
    S.GraphConvFunction._backward_rows = staticmethod(
        lambda ctx, grad: (seen.append(type(grad).__name__), orig(ctx, grad))[1])
    try:
        model.zero_grad(set_to_none=True)
        h = model.gc1(x, g, relu=True)
        out = model.gc2(h, g, log_softmax=True).as_subclass(RowSelectable)
        torch.nn.functional.nll_loss(out[idx], labels[idx]).backward()
    finally:
        S.GraphConvFunction._backward_rows = orig
    assert seen == ["RowGrad", "RowGrad"]                                     # both layers took it
    for k, q in model.named_parameters():
        assert_normwise(q.grad.cpu(), got[k].cpu().numpy(), ROUTES, "layer-by-layer rows route: " + k)


def _oracle_step(oracle, model, x, a, labels_np, idx_np, relu_mask=None):
    p = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    return oracle.gcn2_loss_backward(x.detach().float().cpu().numpy(), a, p, labels_np, idx_np,
                                     relu_mask=relu_mask)


@pytest.mark.parametrize("fin,hid,ncls", [(256, 256, 256), (200, 256, 40), (128, 128, 128), (48, 64, 16),
                                          (256, 256, 64)])
def test_dense_loss_step_matches_oracle(oracle, dev, poison, fin, hid, ncls):
    """A loss over ALL vertices (the fork's live loss reduces over every node, pygcn/train.py:151-155)
    through the model's one node, `_gcn2_backward_dense`: the structural mean-NLL gradient
    (pygcn_amd.functional.nll_loss -> NLLGrad -> gcn_nll_log_softmax_backward_colsum), torch's
    F.nll_loss (a dense gradient) and the layer-by-layer path — all against the oracle, at the
    bench's shape and at shapes that take every fallback of the dispatch (a non-reassociable first
    layer 200 -> 256, a class count 40 / 16 inside one wavefront, fp32 128 -> 128 without a
    hand-written GEMM, 48 -> 64)."""
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.functional import NLLGrad, nll_loss
    from pygcn_amd.utils import rmat_graph
    import pygcn_amd.fused as Fz
    n = 30000
    rowptr, col, val = rmat_graph(n, 300000, seed=31, device="cpu")
    a = oracle.CSR(rowptr.numpy().astype(np.int64), col.numpy(), val.numpy(), (n, n))
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    x = torch.from_numpy(gin.dense((n, fin), 9)).to(dev)
    labels_np = np.random.default_rng(fin).integers(0, ncls, n)
    labels = torch.from_numpy(labels_np).to(dev)
    torch.manual_seed(3)
    model = GCN(fin, hid, ncls, dropout=0.0).to(dev)
    model.train()
    # every hidden unit takes part in a loss over all vertices: the ReLU derivative at units within
    # rounding of zero is a convention — the oracle is given the device's (see device_relu_mask)
    from _sampling import device_relu_mask
    mask, _ = device_relu_mask(oracle, model, x, g, a)
    ref_loss, fw, grads, _ = _oracle_step(oracle, model, x, a, labels_np, np.arange(n), relu_mask=mask)
    _, _, grads64 = oracle.gcn2_loss_backward_f64(
        x.cpu().numpy(), a, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()},
        labels_np, np.arange(n), relu_mask=mask)
    seen = []
    orig = Fz._gcn2_backward_dense
    Fz._gcn2_backward_dense = lambda ctx, x_, w1, w2, h1, logp, grad, needs: (
        seen.append(type(grad).__name__), orig(ctx, x_, w1, w2, h1, logp, grad, needs))[1]
    try:
        for route in ("structural", "dense"):
            model.zero_grad(set_to_none=True)
            out = model(x, g)
            loss = nll_loss(out, labels) if route == "structural" else torch.nn.functional.nll_loss(out, labels)
            loss.backward()
            assert abs(loss.item() - ref_loss) <= TOL * abs(ref_loss)
            assert_normwise(out.detach().cpu(), fw["logp"], TOL, route + ": log-probabilities")
            for k, v in grads.items():
                mod, name = k.split(".")
                got = getattr(getattr(model, mod), name).grad
                assert torch.isfinite(got).all(), k
                assert_parity(got.cpu(), v, grads64[k], f"{route} route: {k}.grad")
    finally:
        Fz._gcn2_backward_dense = orig
    assert seen == ["NLLGrad", "Tensor"]
    # NLLGrad is an ordinary gradient for everybody else
    lp = torch.randn(50, 8, device=dev).log_softmax(1).requires_grad_(True)
    t = torch.randint(0, 8, (50,), device=dev)
    nll_loss(lp * 1.0, t).backward()
    g1 = lp.grad.clone()
    lp.grad = None
    torch.nn.functional.nll_loss(lp * 1.0, t).backward()
    assert torch.allclose(g1, lp.grad, rtol=1e-6, atol=0)
    ng = NLLGrad(t, torch.full((1,), -0.02, device=dev), (50, 8), torch.float32)
    assert torch.equal(ng.dense().nonzero()[:, 1], t) and abs(ng.dense().sum().item() + 1.0) < 1e-6


def test_mean_over_nodes_loss_through_the_one_node_path(oracle, dev):
    """The fork's live loss (pygcn/train.py:151-155): `compressed = torch.mean(gcn_output, axis=0)`,
    a small head, an MSE — every row of the gradient is non-zero and identical.  One node vs the
    layer-by-layer path vs torch's CPU autograd on the same dense arithmetic in float64."""
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.utils import rmat_graph
    n, F = 20000, 256
    rowptr, col, val = rmat_graph(n, 200000, seed=41, device="cpu")
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    x = torch.from_numpy(gin.dense((n, F), 10)).to(dev)
    torch.manual_seed(11)
    model = GCN(F, F, F, dropout=0.0).to(dev)
    head = torch.nn.Linear(F, 1).to(dev)
    model.train()

    def loss_of(out):
        return torch.nn.functional.mse_loss(head(out.mean(0)).squeeze(), torch.tensor(0.25, device=out.device))
    loss_of(model(x, g)).backward()
    one = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    h = model.gc1(x, g, relu=True)
    loss_of(model.gc2(h, g, log_softmax=True)).backward()
    for k, p in model.named_parameters():
        assert_normwise(one[k].cpu(), p.grad.cpu().numpy(), ROUTES, "one node vs layers: " + k)
    # float64 CPU autograd of the same function (torch.spmm = the reference's call), with the
    # device's ReLU derivative at the units within rounding of zero (tests/_sampling.py)
    A = torch.sparse_csr_tensor(rowptr.long(), col.long(), val.double(), (n, n))
    P = {k: v.detach().double().cpu().requires_grad_(True) for k, v in model.state_dict().items()}
    xd = x.double().cpu()
    model.eval()
    with torch.no_grad():
        mask = (model.gc1(x, g, relu=True) > 0).cpu()
    pre = torch.sparse.mm(A, xd @ P["gc1.weight"]) + P["gc1.bias"]
    flips = mask != (pre.detach() > 0)
    assert float(pre.detach()[flips].abs().max()) <= 1e-5 * float(pre.detach().abs().max()) if flips.any() else True
    h1 = pre * mask
    lp = torch.log_softmax(torch.sparse.mm(A, h1 @ P["gc2.weight"]) + P["gc2.bias"], 1)
    hw, hb = head.weight.detach().double().cpu(), head.bias.detach().double().cpu()
    ref = torch.nn.functional.mse_loss((lp.mean(0) @ hw.t() + hb).squeeze(), torch.tensor(0.25, dtype=torch.float64))
    ref.backward()
    for k in one:
        assert_normwise(one[k].cpu(), P[k].grad.numpy(), TOL, "vs float64 autograd: " + k)


def test_row_sets_of_aliasing_index_views_do_not_collide(oracle, dev):
    """idx[:100] and idx[0:200:2] share storage pointer, length and version counter —
    the row-set cache must tell them apart (stride / offset are part of the key)."""
    from pygcn_amd import GCN, CSRGraph, fused
    from pygcn_amd.utils import rmat_graph
    n = 20000
    rowptr, col, val = rmat_graph(n, 200000, seed=51, device="cpu")
    a = oracle.CSR(rowptr.numpy().astype(np.int64), col.numpy(), val.numpy(), (n, n))
    g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
    x = torch.from_numpy(gin.dense((n, 256), 12)).to(dev)
    labels_np = np.random.default_rng(8).integers(0, 256, n)
    labels = torch.from_numpy(labels_np).to(dev)
    base = torch.from_numpy(np.random.default_rng(9).permutation(n)[:3000]).to(dev)
    v1, v2 = base[:1500], base[0:3000:2]
    assert v1.data_ptr() == v2.data_ptr() and v1.numel() == v2.numel() and v1._version == v2._version
    assert fused.rows_key(v1) != fused.rows_key(v2)
    torch.manual_seed(6)
    model = GCN(256, 256, 256, dropout=0.0).to(dev)
    model.train()
    for view in (v1, v2, v1):
        model.zero_grad(set_to_none=True)
        out = model(x, g)
        torch.nn.functional.nll_loss(out[view], labels[view]).backward()
        _, _, grads, _ = _oracle_step(oracle, model, x, a, labels_np, view.cpu().numpy())
        _, _, grads64 = oracle.gcn2_loss_backward_f64(
            x.cpu().numpy(), a, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()},
            labels_np, view.cpu().numpy())
        for k, v in grads.items():
            mod, name = k.split(".")
            assert_parity(getattr(getattr(model, mod), name).grad.cpu(), v, grads64[k], f"aliasing views: {k}")


def test_upstream_loss_lines_with_ignore_index_and_bf16(oracle, dev):
    """`F.nll_loss(output[idx_train], labels[idx_train])` on the model's output runs as the gather /
    scatter pair (rowgrad.LossRows) and the all-vertices loss as the structural NLLGrad — both must
    honour torch's `ignore_index` (labels of -100 add nothing and do not count), at fp32 against
    torch's own kernels on a plain tensor."""
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.functional import nll_loss
    from pygcn_amd.rowgrad import LossRows
    from pygcn_amd.utils import rmat_graph
    n, F_ = 20000, 64
    rowptr, col, val = rmat_graph(n, 200000, seed=61, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    x = torch.randn(n, F_, device=dev)
    labels = torch.randint(0, F_, (n,), device=dev)
    labels[::7] = -100
    idx = torch.randperm(n, device=dev)[: n // 4]
    torch.manual_seed(2)
    model = GCN(F_, F_, F_, dropout=0.0).to(dev)
    model.train()
    results = []
    for route in ("intercepted", "plain"):
        model.zero_grad(set_to_none=True)
        out = model(x, g)
        if route == "plain":
            out = out.as_subclass(torch.Tensor)
        sel = out[idx]
        assert isinstance(sel, LossRows) == (route == "intercepted")
        loss = torch.nn.functional.nll_loss(sel, labels[idx])
        loss.backward()
        results.append((loss.item(), [p.grad.clone() for p in model.parameters()]))
    assert abs(results[0][0] - results[1][0]) <= 1e-6 * abs(results[1][0])
    for a, b in zip(results[0][1], results[1][1]):
        assert_normwise(a.cpu(), b.cpu().numpy(), ROUTES, "LossRows route vs torch's nll_loss")
    # all vertices, structural gradient with ignored rows vs torch's dense gradient
    results = []
    for fn in (nll_loss, torch.nn.functional.nll_loss):
        model.zero_grad(set_to_none=True)
        loss = fn(model(x, g), labels)
        loss.backward()
        results.append((loss.item(), [p.grad.clone() for p in model.parameters()]))
    assert abs(results[0][0] - results[1][0]) <= 1e-6 * abs(results[1][0])
    for a, b in zip(results[0][1], results[1][1]):
        assert_normwise(a.cpu(), b.cpu().numpy(), ROUTES, "NLLGrad with ignored rows vs torch")


def test_hooks_and_a_second_backward_see_ordinary_gradients(oracle, dev):
    """the model's output is a Tensor subclass in training mode and its
    gradient may travel as a storage-less wrapper (RowGrad / NLLGrad).  Whoever else touches that
    gradient must see the ordinary dense tensor: a hook registered on the output (it can compute
    with it and even replace it), `retain_graph=True` followed by a second backward, and
    `torch.autograd.grad` with respect to the output."""
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.functional import nll_loss
    from pygcn_amd.utils import rmat_graph
    n, F_ = 20000, 64
    rowptr, col, val = rmat_graph(n, 200000, seed=71, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    x = torch.randn(n, F_, device=dev)
    labels = torch.randint(0, F_, (n,), device=dev)
    idx = torch.randperm(n, device=dev)[: n // 10]
    torch.manual_seed(4)
    model = GCN(F_, F_, F_, dropout=0.0).to(dev)
    model.train()

    def grads(loss_of, hook=None, twice=False):
        model.zero_grad(set_to_none=True)
        out = model(x, g)
        if hook is not None:
            out.register_hook(hook)
        loss = loss_of(out)
        loss.backward(retain_graph=twice)
        if twice:
            loss.backward()
        return [p.grad.clone() for p in model.parameters()]
    for loss_of in (lambda o: torch.nn.functional.nll_loss(o[idx], labels[idx]), lambda o: nll_loss(o, labels)):
        base = grads(loss_of)
        seen = []

        def hook(gr):
            seen.append((tuple(gr.shape), float(gr.abs().sum())))       # computing with it materialises it
            return gr * 2.0                                              # ... and so does replacing it
        doubled = grads(loss_of, hook=hook)
        assert seen and seen[0][0] == (n, F_) and seen[0][1] > 0
        for a, b in zip(doubled, base):
            assert_normwise(a.cpu(), 2.0 * b.cpu().numpy(), ROUTES, "hook that doubles the gradient")
        two = grads(loss_of, twice=True)                                 # gradients accumulate over two passes
        for a, b in zip(two, base):
            assert_normwise(a.cpu(), 2.0 * b.cpu().numpy(), ROUTES, "retain_graph + second backward")
        model.zero_grad(set_to_none=True)
        out = model(x, g)
        (g_out,) = torch.autograd.grad(loss_of(out), out, retain_graph=True)
        dense = g_out + 0                                                # any operator sees the dense tensor
        assert type(dense) is torch.Tensor and dense.shape == (n, F_) and torch.isfinite(dense).all()
        assert int((dense != 0).any(1).sum()) in (idx.numel(), n)


def test_input_product_cache_is_bitwise_neutral_and_notices_changes(dev):
    """fused.set_input_product_cache(True): z = Â·X of the reassociated first layer is computed once
    per (graph, X, versions); results are bitwise those of the uncached run, an in-place edit of X
    or of the adjacency values invalidates it, and one forward product per epoch disappears."""
    from pygcn_amd import GCN, CSRGraph, fused
    from pygcn_amd import spmm as S
    from pygcn_amd.utils import rmat_graph
    n, F_ = 20000, 256
    rowptr, col, val = rmat_graph(n, 200000, seed=81, device=dev)
    g = CSRGraph(rowptr, col, val, (n, n))
    x = torch.randn(n, F_, device=dev)
    labels = torch.randint(0, F_, (n,), device=dev)
    idx = torch.arange(n // 10, device=dev)
    torch.manual_seed(8)
    model = GCN(F_, F_, F_, dropout=0.0).to(dev)
    model.train()

    def step():
        model.zero_grad(set_to_none=True)
        out = model(x, g, rows=idx)
        torch.nn.functional.nll_loss(out, labels[idx]).backward()
        return out.detach().clone(), [p.grad.clone() for p in model.parameters()]
    base = step()
    launches = []
    S.set_timing_records(launches)
    try:
        fused.set_input_product_cache(True)
        first, second = step(), step()
        n_fwd = [sum(1 for r in launches if r[0] == "fwd")]
        x.mul_(1.0)                                           # version bump: must recompute
        third = step()
        n_fwd.append(sum(1 for r in launches if r[0] == "fwd"))
    finally:
        fused.set_input_product_cache(False)
        S.set_timing_records(None)
    for got in (first, second, third):
        assert torch.equal(got[0], base[0])
        for a, b in zip(got[1], base[1]):
            assert torch.equal(a, b)
    assert n_fwd[0] == 2 + 1 and n_fwd[1] == n_fwd[0] + 2     # 2 products, then 1 (cached), then 2 again
