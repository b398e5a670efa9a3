"""Pins the CPU oracle (oracle/) against the golden vectors captured from the IMPORTED reference
(tests/golden/make_golden.py).  CPU only; no product code involved."""
import numpy as np
import pytest

import inputs as gin
from conftest import assert_normwise, load_golden
from make_golden_cases import EDGE_CASES

TOL = 1e-5   # BASELINE.json north_star: within 1e-5 relative fp32 (normwise, see conftest)


@pytest.fixture(scope="module")
def cora(oracle):
    g = load_golden("cora_graph.npz")
    adj = oracle.cora_adjacency(g["edges"], int(g["n"]))
    return g, adj


def test_cora_recipe_matches_reference_adjacency(cora):
    g, adj = cora
    assert adj.shape == (2708, 2708) and adj.nnz == int(g["nnz"]) == 13264
    np.testing.assert_array_equal(adj.rowptr, g["csr_rowptr"])
    np.testing.assert_array_equal(adj.col, g["csr_col"])
    np.testing.assert_allclose(adj.val, g["csr_val"], rtol=1e-7, atol=0)
    rowsum = np.add.reduceat(adj.val.astype(np.float64), adj.rowptr[:-1])
    np.testing.assert_allclose(rowsum, 1.0, atol=1e-6)


def test_init_bounds_match_reference_init(oracle):
    g1 = load_golden("g1_init.npz")
    for name, (fin, fout) in {"gc1": (1433, 16), "gc2": (16, 7)}.items():
        wb, bb = oracle.init_bounds(fin, fout)
        w, b = g1[name + "_weight"], g1[name + "_bias"]
        assert w.shape == (fin, fout) and b.shape == (fout,)
        assert np.abs(w).max() <= wb and np.abs(b).max() <= bb
        if w.size > 1000:   # the uniform fills the interval
            assert np.abs(w).max() > 0.99 * wb
    assert str(g1["repr"]) == "GraphConvolution (1433 -> 16)"


def test_cora_step_matches_reference(oracle, cora):
    _, adj = cora
    g1, g2 = load_golden("g1_init.npz"), load_golden("g2_cora_step.npz")
    x, labels, idx = gin.cora_features(), gin.cora_labels(), gin.cora_splits()[0]
    assert abs(np.abs(x).sum(dtype=np.float64) - float(g2["features_abs_sum"])) < 1e-6
    p = {"gc1.weight": g1["gc1_weight"], "gc1.bias": g1["gc1_bias"],
         "gc2.weight": g1["gc2_weight"], "gc2.bias": g1["gc2_bias"]}
    loss, fw, grads, mid = oracle.gcn2_loss_backward(x, adj, p, labels, idx, need_grad_x=True)
    assert abs(loss - float(g2["loss"])) <= TOL * abs(float(g2["loss"]))
    for k in ("h1", "h2", "logp"):
        assert_normwise(fw[k], g2[k], TOL, k)
    for k in ("grad_h2", "grad_a1", "grad_h1"):
        assert_normwise(mid[k], g2[k], TOL, k)
    assert_normwise(mid["grad_x"][:64], g2["grad_x_head"], TOL, "grad_x")
    for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias"):
        assert_normwise(grads[k], g2[k.replace(".", "_") + "_grad"], TOL, k)


def test_generator_gcn_stack_matches_reference(oracle):
    """Three GraphConvolution+ReLU layers stacked by the reference's own model code
    (models.py:74-124)."""
    g3 = load_golden("g3_generator_gcn.npz")
    n = 64
    rows, cols, vals = gin.random_coo(n, n, 400, seed=300)
    adj = oracle.CSR.from_coo(rows, cols, vals, (n, n))
    x = gin.dense((n, 8), 301)
    acts, pre = [x], []
    for i in (1, 2, 3):
        h, _ = oracle.gc_forward(acts[-1], g3[f"param_gc{i}.weight"], g3[f"param_gc{i}.bias"], adj)
        pre.append(h)
        acts.append(np.maximum(h, 0))
    assert_normwise(acts[-1], g3["y"], TOL, "y")
    g = gin.dense((n, 32), 302)
    for i in (3, 2, 1):
        g = g * (pre[i - 1] > 0)
        g, gw, gb, _ = oracle.gc_backward(acts[i - 1], g3[f"param_gc{i}.weight"], True, adj, g)
        assert_normwise(gw, g3[f"grad_gc{i}.weight"], TOL, f"gw{i}")
        assert_normwise(gb, g3[f"grad_gc{i}.bias"], TOL, f"gb{i}")
    assert_normwise(g, g3["grad_x"], TOL, "grad_x")


@pytest.mark.parametrize("case", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_cases_match_reference(oracle, case):
    g4 = load_golden("g4_edge_cases.npz")
    k = EDGE_CASES.index(case)
    name, nr, nc, nnz, fin, fout, bias, kw = case
    seed = 400 + 10 * k
    rows, cols, vals = gin.random_coo(nr, nc, nnz, seed=seed, **kw)
    x, g = gin.dense((nc, fin), seed + 1), gin.dense((nr, fout), seed + 2)
    w = g4[name + "/weight"]
    b = g4[name + "/bias"] if bias else None
    adj = oracle.CSR.from_coo(rows, cols, vals, (nr, nc))
    y, support = oracle.gc_forward(x, w, b, adj)
    assert_normwise(y, g4[name + "/y"], TOL, "y")
    assert_normwise(adj.matmul(support), g4[name + "/spmm"], TOL, "spmm csr")
    assert_normwise(oracle.spmm_coo(rows, cols, vals, support, nr), g4[name + "/spmm"], TOL,
                    "spmm coo")
    assert_normwise(adj.t_matmul(g), g4[name + "/spmm_t"], TOL, "spmm_t scatter")
    rp_t, col_t, val_t = oracle.csr_transpose(adj.rowptr, adj.col, adj.val, nc)
    assert_normwise(oracle.spmm_csr(rp_t, col_t, val_t, g), g4[name + "/spmm_t"], TOL,
                    "spmm on CSR(A^T)")
    gx, gw, gb, _ = oracle.gc_backward(x, w, bias, adj, g)
    assert_normwise(gx, g4[name + "/grad_x"], TOL, "grad_x")
    assert_normwise(gw, g4[name + "/grad_weight"], TOL, "grad_w")
    if bias:
        assert_normwise(gb, g4[name + "/grad_bias"], TOL, "grad_b")
    if name + "/spmm_f64" in g4.files:
        ref64 = g4[name + "/spmm_f64"]
        assert_normwise(oracle.spmm_csr_f64acc(adj.rowptr, adj.col, adj.val, support), ref64,
                        1e-6, "f64acc")


def test_training_trajectory_matches_reference(oracle, cora):
    """200 Adam epochs of the upstream-semantics 2-layer GCN (dropout 0).  The trajectory is a
    long chain of fp32 steps, so the gate here is looser than the single-step gate: 1e-3 on the
    loss curve, documented in DESIGN.md (single-step parity is test_cora_step_*, at 1e-5)."""
    _, adj = cora
    g1, g5 = load_golden("g1_init.npz"), load_golden("g5_trajectory.npz")
    p = {"gc1.weight": g1["gc1_weight"], "gc1.bias": g1["gc1_bias"],
         "gc2.weight": g1["gc2_weight"], "gc2.bias": g1["gc2_bias"]}
    x, labels, idx = gin.cora_features(), gin.cora_labels(), gin.cora_splits()[0]
    losses, accs, final = oracle.train_trajectory(x, adj, p, labels, idx, epochs=200)
    ref = g5["loss_train"]
    assert ref[-1] < ref[0]
    np.testing.assert_allclose(losses[:5], ref[:5], rtol=1e-5)
    np.testing.assert_allclose(losses, ref, rtol=1e-3)
    assert np.abs(accs - g5["acc_train"]).max() <= 1.5 / len(idx)
    # parameters after 200 chained Adam steps drift further than the loss does (measured 4e-3)
    assert_normwise(final["gc2.weight"], g5["final_gc2_weight"], 2e-2, "final gc2.weight")
