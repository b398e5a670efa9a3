"""Ablation sweep of gcn_gemm_atg256_f32 (weight gradient Aᵀ·G, fp32 256 x 256, M = 10^7): every
variant is a libgcn_*.so built with a different -DATG_ABLATE (tools/build_gemm_variants.sh atg) —
what the kernel costs without the loads of its step loop (the arithmetic alone).  (Variants that
drop the MFMAs or the fp16 split tell nothing: hipcc then removes the whole loop as dead code.)
Usage: python tools/atg_variant_sweep.py build/variants/*.so"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")
M = int(os.environ.get("GEMM_M", 10_000_000))
A = torch.randn(M, 256, device=dev); G = torch.randn(M, 256, device=dev) * 0.01
rows = torch.arange((M + 15) // 16 * 16, device=dev, dtype=torch.int32).clamp_(max=M - 1)
ab, gb = A.abs().max().reshape(1), G.abs().max().reshape(1)
out = torch.empty(256, 256, device=dev)
vp, i64 = ctypes.c_void_p, ctypes.c_int64
libs = {}
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.gcn_gemm_atg256_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_gemm_atg256_workspace_bytes.argtypes = [i64]
    L.gcn_gemm_atg256_f32.restype = ctypes.c_int
    L.gcn_gemm_atg256_f32.argtypes = [vp, i64, vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, ctypes.c_size_t, vp]
    libs[os.path.basename(path).replace("libgcn_", "").replace(".so", "")] = L
ws = torch.empty(next(iter(libs.values())).gcn_gemm_atg256_workspace_bytes(M), dtype=torch.uint8, device=dev)
def run(L):
    rc = L.gcn_gemm_atg256_f32(A.data_ptr(), 256, rows.data_ptr(), G.data_ptr(), 256, rows.data_ptr(), M, ab.data_ptr(),
                               gb.data_ptr(), out.data_ptr(), 256, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(3):
    print("round %d  " % rnd + "  ".join("%s %.2f" % (n, t(lambda: run(L))) for n, L in libs.items()), flush=True)
