"""Accuracy and time of the two hand-written X[M,256]·W[256,256] kernels against hipBLASLt (torch.mm)
in one process (A/B on one box): gcn_gemm_xw256_f32 (3 x bf16) and gcn_gemm_xw256_f32_h2 (2 x fp16)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd.spmm import gemm_xw256
dev = torch.device("cuda:0")
for M in (1000, 100003):
    X = torch.randn(M, 256, device=dev) * torch.rand(M, 1, device=dev) * 10
    W = torch.randn(256, 256, device=dev)
    ref64 = X.double() @ W.double()
    s = ref64.abs().max().item()
    for name, Y in (("bf16x3", gemm_xw256(X, W)), ("h2", gemm_xw256(X, W, x_bound=X.abs().max().reshape(1))),
                    ("torch fp32", X @ W)):
        print(f"M={M:7d}  {name:10s} normwise err {((Y.double()-ref64).abs().max().item()/s):.3e}", flush=True)
M = int(os.environ.get("GEMM_M", 10_000_000))
X = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev); Y = torch.empty(M, 256, device=dev)
b = X.abs().max().reshape(1)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(3):      # interleaved rounds
    print("round %d: bf16x3 %.2f ms   h2 %.2f ms   torch.mm %.2f ms" % (
        rnd, t(lambda: gemm_xw256(X, W)), t(lambda: gemm_xw256(X, W, x_bound=b)), t(lambda: torch.mm(X, W, out=Y))), flush=True)
