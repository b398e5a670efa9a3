"""Times of the fp32 256 -> 256 GEMM (gcn_gemm_xw256_f32_h2) at M = 10^7: plain, with the layer's
forward epilogue (bias + ReLU + dropout) and with the ReLU / dropout backward mask in its store."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd.spmm import gemm_xw256
dev = torch.device("cuda:0")
M = int(os.environ.get("GEMM_M", 10_000_000))
X = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev) * 0.06
H = torch.relu(torch.randn(M, 256, device=dev))
b = X.abs().max().reshape(1)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
Y = gemm_xw256(X, W, x_bound=b, mask_src=H, mask_scale=2.0)
ref = torch.where(H[:4096] > 0, (X[:4096].double() @ W.double()) * 2.0, torch.zeros((), dtype=torch.float64, device=dev))
print("masked normwise err %.3e" % ((Y[:4096].double() - ref).abs().max() / ref.abs().max()).item())
bias = torch.randn(256, device=dev) * 0.1
gb = M * 256 * 4 / 1e9
for rnd in range(3):
    a = t(lambda: gemm_xw256(X, W, x_bound=b))
    f = t(lambda: gemm_xw256(X, W, x_bound=b, bias=bias, relu=True, dropout_p=0.5, seed=1234))
    m = t(lambda: gemm_xw256(X, W, x_bound=b, mask_src=H, mask_scale=2.0))
    print("round %d  plain %.2f ms (%.2f)   forward %.2f ms (%.2f)   masked %.2f ms (%.2f)   [fraction of 8 TB/s]"
          % (rnd, a, 2 * gb / a / 8, f, 2 * gb / f / 8, m, 3 * gb / m / 8), flush=True)
