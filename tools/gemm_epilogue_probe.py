"""gcn_gemm_xw256_f32_h2 with the forward epilogue (bias + ReLU + Philox dropout) across build variants
(one process, interleaved): python tools/gemm_epilogue_probe.py build/variants/*.so"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd._native import GcnGemmEpilogue
dev = torch.device("cuda:0")
M = 10_000_000
X = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev) * 0.1; Y = torch.empty(M, 256, device=dev)
bias = torch.randn(256, device=dev); b = X.abs().max().reshape(1)
ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
libs = {}
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.gcn_gemm_xw256_f32_h2.restype = ctypes.c_int
    L.gcn_gemm_xw256_f32_h2.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.POINTER(GcnGemmEpilogue), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    libs[os.path.basename(path).replace("libgcn_", "").replace(".so", "")] = L
def run(L, ep):
    rc = L.gcn_gemm_xw256_f32_h2(X.data_ptr(), 256, None, W.data_ptr(), 256, Y.data_ptr(), 256, M, b.data_ptr(), None, ep,
                                 ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
eps = {"plain": None,
       "bias+relu": GcnGemmEpilogue(bias.data_ptr(), 1, 0.0, 0, None, None, 0, 1.0),
       "bias+relu+dropout": GcnGemmEpilogue(bias.data_ptr(), 1, 0.5, 12345, None, None, 0, 1.0)}
ref = {}
for name, L in libs.items():
    for en, ep in eps.items():
        run(L, ep); torch.cuda.synchronize()
        if en not in ref: ref[en] = Y[:40000].clone()
        else: assert torch.equal(Y[:40000], ref[en]), (name, en)
for rnd in range(3):
    print("round %d  " % rnd + "  ".join("%s/%s %.2f" % (n, en, t(lambda: run(L, ep))) for n, L in libs.items() for en, ep in eps.items()), flush=True)
