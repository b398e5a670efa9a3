import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd import _native
L = _native.lib()
dev = torch.device("cuda:0")
def run(X, W, Y, ws):
    rc = L.gcn_gemm_xw256_f32(X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), Y.data_ptr(), Y.stride(0),
                              X.shape[0], ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "gemm")
ws = torch.empty(L.gcn_gemm_xw256_workspace_bytes(), dtype=torch.uint8, device=dev)
for M in (1, 31, 128, 1000, 100003):
    X = torch.randn(M, 256, device=dev) * torch.rand(M, 1, device=dev) * 10
    W = torch.randn(256, 256, device=dev)
    Y = torch.empty(M, 256, device=dev)
    run(X, W, Y, ws); torch.cuda.synchronize()
    ref64 = (X.double() @ W.double())
    ref32 = X @ W
    s = ref64.abs().max().item()
    print(f"M={M:7d}  custom err {((Y.double()-ref64).abs().max().item()/s):.3e}   torch fp32 err {((ref32.double()-ref64).abs().max().item()/s):.3e}", flush=True)
M = 10_000_000
X = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev); Y = torch.empty(M, 256, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("custom  %.2f ms" % t(lambda: run(X, W, Y, ws)))
print("torch   %.2f ms" % t(lambda: torch.mm(X, W, out=Y)))
ref = torch.mm(X[:100000], W); run(X, W, Y, ws); torch.cuda.synchronize()
print("big-M check rel err %.3e" % ((Y[:100000] - ref).abs().max().item() / ref.abs().max().item()))
