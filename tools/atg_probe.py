"""Times of the gather-fused weight-gradient kernels (grad_W = Aᵀ · G over a row list) at full
height — fp32 256 x 256 at M = 10^7 (gcn_gemm_atg256_f32) and bf16 128 x 128 at M = 5·10^7
(gcn_gemm_atg_bf16) — with the fraction of the HBM roofline (both operands read once)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd.spmm import weight_grad_rows
dev = torch.device("cuda:0")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dtype, M, F in ((torch.float32, 10_000_000, 256), (torch.bfloat16, 50_000_000, 128)):
    A = torch.randn(M, F, device=dev, dtype=dtype); G = torch.randn(M, F, device=dev, dtype=dtype) * 0.01
    kw = {} if dtype == torch.bfloat16 else {"a_bound": A.abs().max().reshape(1), "g_bound": G.abs().max().reshape(1)}
    out = weight_grad_rows(A, G, **kw)
    n = 200_000
    ref = A[:n].double().t() @ G[:n].double()
    got = weight_grad_rows(A[:n], G[:n], **kw)
    print(f"{dtype}: normwise err on {n} rows %.3e" % ((got.double() - ref).abs().max() / ref.abs().max()).item())
    gb = 2 * M * F * A.element_size() / 1e9
    for rnd in range(2):
        ms = t(lambda: weight_grad_rows(A, G, **kw))
        print(f"  round {rnd}: {ms:.2f} ms  ({gb / ms / 8:.2f} of 8 TB/s)", flush=True)
    del A, G
