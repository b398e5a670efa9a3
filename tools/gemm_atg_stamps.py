"""gemm_atg256_h2_kernel under -DGEMM_PROFILE_STAMPS: wave lifetime in shader cycles vs wall time = the clock the
chip holds, and the cycles a 32-row super-step costs a wave (GCN_SPMM_LIB selects the experiment build)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import _native, spmm as S   # noqa: E402

dev = torch.device("cuda:0")
L = _native.lib()
L.gcn_debug_gemm_stamps.restype = ctypes.c_int
M = int(os.environ.get("GEMM_M", 10_000_000))
X = torch.randn(M, 256, device=dev)
Gd = torch.randn(M, 256, device=dev)
xb, gb = X.abs().max().reshape(1), Gd.abs().max().reshape(1)


def stamps(reset=True):
    out = (ctypes.c_ulonglong * 8)()
    assert L.gcn_debug_gemm_stamps(out, int(reset)) == 0
    return list(out)


for scheme in ("bf16x3", "h2"):
    S.set_gemm_scheme(scheme)
    for _ in range(2):
        S.weight_grad_rows(X, Gd, a_bound=xb, g_bound=gb)
    torch.cuda.synchronize()
    stamps()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    reps = 5
    for _ in range(reps):
        S.weight_grad_rows(X, Gd, a_bound=xb, g_bound=gb)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    life, waves = stamps()[3:5]
    supers = M / 32 / (waves / reps / 8)
    print(f"{os.path.basename(os.environ.get('GCN_SPMM_LIB', 'product')):18s} {scheme:7s} {ms:6.2f} ms | wave lifetime {life / waves:10.0f} cyc "
          f"(of a launch incl. its reduce kernel: {life / waves / (ms * 1e-3) / 1e9:5.2f} GHz-equivalent) | workgroups {waves / reps / 8:.0f} | "
          f"cycles per 32 rows {life / waves / supers:7.0f} (matrix pipe: {6144 if scheme == 'bf16x3' else 3072})", flush=True)
S.set_gemm_scheme("bf16x3")
