"""Times of the bf16 streaming GEMM (gcn_gemm_xw_bf16, 128 -> 128) at config C5's height, every
store variant: plain, forward epilogue (bias + ReLU + dropout), backward mask at the output row,
backward mask through a row list — with the fraction of the HBM roofline each reaches (bytes: X and
Y once, the mask once where there is one)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd.spmm import gemm_bf16
dev = torch.device("cuda:0")
M, F = int(os.environ.get("GEMM_M", 50_000_000)), 128
X = torch.randn(M, F, device=dev, dtype=torch.bfloat16)
W = (torch.randn(F, F, device=dev) * 0.09).bfloat16()
bias = torch.randn(F, device=dev) * 0.1
H = torch.relu(torch.randn(M, F, device=dev, dtype=torch.bfloat16))
rows = torch.arange(M, device=dev, dtype=torch.int32)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n = 4096
ref = X[:n].double() @ W.double()
Y = gemm_bf16(X, W)
print("plain  normwise err %.3e" % ((Y[:n].double() - ref).abs().max() / ref.abs().max()).item())
Ym = gemm_bf16(X, W, mask_src=H, mask_scale=2.0)
refm = torch.where(H[:n] > 0, ref * 2.0, torch.zeros((), dtype=torch.float64, device=dev))
print("masked normwise err %.3e" % ((Ym[:n].double() - refm).abs().max() / refm.abs().max()).item())
assert torch.equal(Ym, gemm_bf16(X, W, mask_src=H, mask_rows=rows, mask_scale=2.0))
gb = M * F * 2 / 1e9
for rnd in range(2):
    a = t(lambda: gemm_bf16(X, W))
    b = t(lambda: gemm_bf16(X, W, bias=bias, relu=True, dropout_p=0.5, seed=1234))
    c = t(lambda: gemm_bf16(X, W, mask_src=H, mask_scale=2.0))
    d = t(lambda: gemm_bf16(X, W, mask_src=H, mask_rows=rows, mask_scale=2.0))
    print("round %d  plain %.2f ms (%.2f)  forward %.2f ms (%.2f)  mask %.2f ms (%.2f)  mask+rows %.2f ms (%.2f)   [fraction of 8 TB/s]"
          % (rnd, a, 2 * gb / a / 8, b, 2 * gb / b / 8, c, 3 * gb / c / 8, d, 3 * gb / d / 8), flush=True)
