"""Build-variant sweep of gcn_gemm_xw256_f32_h2 (one process, one box, interleaved rounds): every
variant is a separate libgcn_*.so built with different -D knobs (tools/build_gemm_variants.sh).
Times the plain product and the forward-epilogue instantiation (bias + ReLU + dropout) and checks
that all variants store the same bits.
Usage: python tools/gemm_variant_sweep.py build/variants/*.so"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd._native import GcnGemmEpilogue
dev = torch.device("cuda:0")
M = int(os.environ.get("GEMM_M", 10_000_000))
X = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev) * 0.06; Y = torch.empty(M, 256, device=dev)
bias = torch.randn(256, device=dev) * 0.1
b = X.abs().max().reshape(1)
libs = {}
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.gcn_gemm_xw256_h2_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_gemm_xw256_f32_h2.restype = ctypes.c_int
    L.gcn_gemm_xw256_f32_h2.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.POINTER(GcnGemmEpilogue), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_last_error.restype = ctypes.c_char_p
    libs[os.path.basename(path).replace("libgcn_", "").replace(".so", "")] = L
ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
EP = GcnGemmEpilogue(bias.data_ptr(), 1, 0.5, 0x1234, None, None, 0, 1.0, None, 0)
def run(L, ep=None, rows=M):
    rc = L.gcn_gemm_xw256_f32_h2(X.data_ptr(), 256, None, W.data_ptr(), 256, Y.data_ptr(), 256, rows, b.data_ptr(), None,
                                 ctypes.byref(ep) if ep is not None else None, ws.data_ptr(), ws.numel(),
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0, (rc, L.gcn_last_error())
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for ep in (None, EP):
    ref = None
    for rows in (M, 1000003, 777):              # full, ragged tail, a single partial tile
        for name, L in libs.items():
            Y.fill_(float("nan"))
            run(L, ep, rows); torch.cuda.synchronize()
            got = Y[:rows][-60000:].clone(), Y[:50000].clone()
            assert torch.isfinite(got[0]).all() and torch.isfinite(got[1][:min(rows, 50000)]).all(), name
            if name == list(libs)[0]: ref = got
            else: assert torch.equal(got[0], ref[0]) and torch.equal(got[1][:min(rows, 50000)], ref[1][:min(rows, 50000)]), (name, rows)
    print("epilogue" if ep else "plain", "bits equal across variants", flush=True)
ref64 = (X[:4096].double() @ W.double())
run(list(libs.values())[-1]); torch.cuda.synchronize()
print("normwise error vs fp64: %.3e" % ((Y[:4096].double() - ref64).abs().max() / ref64.abs().max()).item(), flush=True)
for rnd in range(3):
    print("round %d  plain: " % rnd + "  ".join("%s %.2f" % (n, t(lambda: run(L))) for n, L in libs.items())
          + "   epilogue: " + "  ".join("%s %.2f" % (n, t(lambda: run(L, EP))) for n, L in libs.items()), flush=True)
