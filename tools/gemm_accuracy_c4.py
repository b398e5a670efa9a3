"""Accuracy of the three GEMM schemes on the ACTUAL dense products of the graded C4 epoch (VERDICT r03
next #1: "on the actual C4 activations and gradients, not randn").

One training step of the bench's model (C4 graph, seeds of bench.py, dropout 0.5, loss on the
idx_train share) is run with every call of the 256-wide GEMM entry points RECORDED — operands, row
lists, epilogue — and each recorded product is then re-evaluated under
    bf16x3  three bf16 parts, six MFMAs (the default; fp32-equivalent)
    h2      two scaled fp16 parts, three MFMAs (opt-in)
    exact   hipBLASLt fp32 (torch.mm; what `torch.mm(input, self.weight)`, pygcn/layers.py:33, runs on a GPU)
and compared with a float64 product: row GEMMs on SAMPLE_ROWS sampled output rows, weight gradients
(reductions over the listed vertices) in full.  Printed per product: max|err| / max|ref| (the
parity metric) and the rms error relative to the rms of the result.

    python tools/gemm_accuracy_c4.py [config]        (config: c4 default, c3, tiny)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import GCN, CSRGraph, fused, spmm as S   # noqa: E402
from pygcn_amd.functional import nll_loss   # noqa: E402
from pygcn_amd.utils import rmat_graph   # noqa: E402

SAMPLE_ROWS = 200_000
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
n, e = {"c4": (10_000_000, 100_000_000), "c3": (1_000_000, 10_000_000), "tiny": (50_000, 500_000)}[cfg]
dev = torch.device("cuda:0")
F = 256
rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
g = CSRGraph(rowptr, col, val, (n, n))
x = torch.randn(n, F, generator=torch.Generator(device=dev).manual_seed(44), device=dev)
labels = torch.randint(0, F, (n,), device=dev, generator=torch.Generator(device=dev).manual_seed(45))
idx = torch.arange(max(1, int(n * 140 / 2708)), device=dev)
torch.manual_seed(42)
model = GCN(F, F, F, dropout=0.5).to(dev)
model.train()

calls = []
real_xw, real_wg = S.gemm_xw256, S.weight_grad_rows


def rec_xw(X, W, x_bound=None, y_absmax=None, rows=None, mask_src=None, mask_scale=1.0, bias=None,
           relu=False, dropout_p=0.0, seed=0, mask_rows=None, row_base=0, **more):
    out = real_xw(X, W, x_bound, y_absmax, rows=rows, mask_src=mask_src, mask_scale=mask_scale, bias=bias,
                  relu=relu, dropout_p=dropout_p, seed=seed, mask_rows=mask_rows, row_base=row_base, **more)
    if out is not None:
        calls.append(("xw", dict(X=X.detach(), W=W.detach().clone(), rows=rows,
                                 what=("forward + layer epilogue" if (bias is not None or relu) else
                                       "grad_input (masked store)" if mask_src is not None else "forward"))))
    return out


def rec_wg(A, G, rows_a=None, rows_g=None, a_bound=None, g_bound=None, n_list=None, **more):
    out = real_wg(A, G, rows_a, rows_g, a_bound, g_bound, n_list, **more)
    if out is not None and A.dtype == torch.float32:
        calls.append(("wg", dict(A=A.detach(), G=G.detach().clone(), rows_a=rows_a, rows_g=rows_g, n_list=n_list)))
    return out


S.gemm_xw256 = fused.gemm_xw256 = rec_xw
S.weight_grad_rows = rec_wg
try:
    torch.manual_seed(7)
    loss = nll_loss(model(x, g, rows=idx).float(), labels[idx])
    loss.backward()
finally:
    S.gemm_xw256 = fused.gemm_xw256 = real_xw
    S.weight_grad_rows = real_wg
torch.cuda.synchronize()
print(f"{cfg}: one training step, loss {loss.item():.6f}; {len(calls)} recorded 256-wide GEMM calls", flush=True)


def errs(got, ref):
    d = (got.double() - ref)
    return float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


def bound(t):
    return t.abs().max().reshape(1).float()


results = []
gen = torch.Generator(device=dev).manual_seed(99)
for kind, c in calls:
    if kind == "xw":
        X, W, rows = c["X"], c["W"], c["rows"]
        m_out = rows.numel() if rows is not None else X.shape[0]
        pick = torch.randperm(m_out, generator=gen, device=dev)[:SAMPLE_ROWS]
        src = rows[pick].long() if rows is not None else pick
        Xs = X.index_select(0, src).contiguous()
        ref = Xs.double() @ W.double()
        row = {"product": f"X[{m_out}, 256] x W — {c['what']}", "max|X|": float(X.abs().max()),
               "share of zeros in X": round(float((Xs == 0).float().mean()), 3)}
        for scheme in ("bf16x3", "h2", "exact"):
            S.set_gemm_scheme(scheme)
            got = real_xw(Xs, W, bound(X)) if scheme != "exact" else torch.mm(Xs, W)
            row[scheme] = dict(zip(("normwise", "rms_rel"), errs(got, ref)))
    else:
        A, G, ra, rg, nl = c["A"], c["G"], c["rows_a"], c["rows_g"], c["n_list"]
        nl = nl if nl is not None else (ra.numel() if ra is not None else A.shape[0])
        Aa = A.index_select(0, ra[:nl].long()) if ra is not None else A[:nl]
        Gg = G.index_select(0, rg[:nl].long()) if rg is not None else G[:nl]
        ref = torch.zeros(256, 256, dtype=torch.float64, device=dev)
        for s0 in range(0, nl, 1 << 20):                           # float64 in slabs (memory)
            ref += Aa[s0:s0 + (1 << 20)].double().t() @ Gg[s0:s0 + (1 << 20)].double()
        row = {"product": f"A[{nl} listed rows]^T x G — weight gradient", "max|A|": float(Aa.abs().max()),
               "max|G|": float(Gg.abs().max())}
        for scheme in ("bf16x3", "h2", "exact"):
            S.set_gemm_scheme(scheme)
            got = real_wg(A, G, ra, rg, bound(Aa), bound(Gg), nl) if scheme != "exact" else torch.mm(Aa.t(), Gg)
            row[scheme] = dict(zip(("normwise", "rms_rel"), errs(got, ref)))
    S.set_gemm_scheme("bf16x3")
    results.append(row)
    print(json.dumps(row), flush=True)
print("ACCURACY_JSON " + json.dumps({"config": cfg, "sample_rows": SAMPLE_ROWS, "products": results}))
