"""Times of the 256-wide GEMM kernels of the library named by GCN_SPMM_LIB (an experiment build,
tools/build_gemm_variant.sh) at M = GEMM_M (default 10^7): plain / layer epilogue / backward mask /
weight gradient, both schemes, 3 interleaved rounds."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import spmm as S   # noqa: E402

dev = torch.device("cuda:0")
M = int(os.environ.get("GEMM_M", 10_000_000))
X = torch.randn(M, 256, device=dev)
W = torch.randn(256, 256, device=dev) * 0.06
bias = torch.randn(256, device=dev)
Gd = torch.randn(M, 256, device=dev)
xb, gb = X.abs().max().reshape(1), Gd.abs().max().reshape(1)


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cases = [("plain", lambda: S.gemm_xw256(X, W, x_bound=xb)),
         ("epi", lambda: S.gemm_xw256(X, W, x_bound=xb, bias=bias, relu=True, dropout_p=0.5, seed=5)),
         ("masked", lambda: S.gemm_xw256(X, W, x_bound=xb, mask_src=Gd, mask_scale=2.0)),
         ("atg", lambda: S.weight_grad_rows(X, Gd, a_bound=xb, g_bound=gb))]
ref = {}
for rnd in range(3):
    out = [os.path.basename(os.environ.get("GCN_SPMM_LIB", "product")), f"round {rnd}:"]
    for scheme in ("bf16x3", "h2"):
        S.set_gemm_scheme(scheme)
        out.append(scheme + " " + " ".join(f"{name} {t(fn):.2f}" for name, fn in cases) + " |")
    print(" ".join(out), flush=True)
S.set_gemm_scheme("bf16x3")
