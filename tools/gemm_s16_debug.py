"""Where does gcn_gemm_xw256_f32_b3 (contiguous rows) differ from an fp64 product?  (debug aid)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import spmm as S   # noqa: E402

dev = torch.device("cuda:0")
S.set_gemm_scheme("bf16x3")
for M in (128, 256, 300, 4099):
    g = torch.Generator(device=dev).manual_seed(M)
    X = torch.randn(M, 256, generator=g, device=dev)
    W = torch.randn(256, 256, generator=g, device=dev)
    Y = S.gemm_xw256(X, W)
    ref = X.double() @ W.double()
    bad = ((Y.double() - ref).abs() > 1e-4 * ref.abs().max()) | ~torch.isfinite(Y)
    print(f"M={M}: wrong entries {int(bad.sum())} of {bad.numel()}; nonfinite {int((~torch.isfinite(Y)).sum())}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("  rows wrong:", rows[:40].tolist(), "... n =", rows.numel())
        print("  cols wrong:", cols[:40].tolist(), "... n =", cols.numel())
        r = int(rows[0])
        print("  row", r, "got", Y[r, :8].tolist(), "\n        ref", ref[r, :8].float().tolist())
    # unit tests of the layout: X = e_k rows  ->  Y[r] = W[k]
    Xe = torch.zeros(M, 256, device=dev)
    k = torch.arange(M, device=dev) % 256
    Xe[torch.arange(M, device=dev), k] = 1.0
    Ye = S.gemm_xw256(Xe, W)
    d = (Ye - W[k]).abs()
    print("  unit rows: max err", float(d.max()), "wrong rows", int((d.amax(1) > 1e-5).sum()))
    if d.max() > 1e-5:
        r = int((d.amax(1) > 1e-5).nonzero()[0])
        # which W row / column did we get instead?
        got = Ye[r]
        match = ((W - got[None, :]).abs().amax(1) < 1e-5).nonzero().flatten().tolist()
        print(f"  row {r} (k={int(k[r])}): matches W rows {match}; got[:6]={got[:6].tolist()} want {W[k[r], :6].tolist()}")
        cm = [(int(c), ((W[k[r]] - got[c]).abs() < 1e-6).nonzero().flatten().tolist()[:4]) for c in range(8)]
        print("   per output column c: W[k] columns equal to got[c]:", cm)
