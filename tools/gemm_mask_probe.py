"""Time of the grad_input GEMM with the ReLU / dropout backward mask in its store (M = 10^7)."""
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
for rnd in range(3):
    print("round %d  plain %.2f ms   masked %.2f ms" % (rnd, t(lambda: gemm_xw256(X, W, x_bound=b)),
                                                        t(lambda: gemm_xw256(X, W, x_bound=b, mask_src=H, mask_scale=2.0))), flush=True)
