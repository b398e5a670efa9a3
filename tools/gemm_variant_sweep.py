"""Build-variant sweep of gcn_gemm_xw256_f32_h2 (one process, one box, interleaved rounds): every
variant is a separate libgcn_*.so built with different -D knobs (tools/build_gemm_variants.sh).
Usage: python tools/gemm_variant_sweep.py build/variants/*.so"""
import ctypes, os, sys, torch
dev = torch.device("cuda:0")
M = int(os.environ.get("GEMM_M", 10_000_000))
X = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev); Y = torch.empty(M, 256, device=dev)
b = X.abs().max().reshape(1)
libs = {}
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.gcn_gemm_xw256_h2_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_gemm_xw256_f32_h2.restype = ctypes.c_int
    L.gcn_gemm_xw256_f32_h2.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    libs[os.path.basename(path)] = L
ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
def run(L):
    rc = L.gcn_gemm_xw256_f32_h2(X.data_ptr(), 256, None, W.data_ptr(), 256, Y.data_ptr(), 256, M, b.data_ptr(), None,
                                 None, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = None
for name, L in libs.items():
    run(L); torch.cuda.synchronize()
    if ref is None: ref = Y[:50000].clone()
    elif "nostore" not in name and "diag1" not in name: assert torch.equal(Y[:50000], ref), name
for rnd in range(3):
    print("round %d  " % rnd + "  ".join("%s %.2f" % (n.replace("libgcn_", "").replace(".so", ""), t(lambda: run(L))) for n, L in libs.items()), flush=True)
