"""Where a tile's cycles go in gemm_xw256_h2_kernel (experiment build with -DGEMM_PROFILE_STAMPS:
tools/build_gemm_variant.sh stamps -DGEMM_PROFILE_STAMPS; run with GCN_SPMM_LIB=build/variants/libgcn_stamps.so).
Per wave and tile: s_memtime cycles from the top of the tile loop to the end of the K loop, and from
there to the end of the store section; the whole wave lifetime; wall time of the launch."""
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
W = torch.randn(256, 256, device=dev) * 0.06
bias = torch.randn(256, device=dev)
xb = X.abs().max().reshape(1)


def step_stamps(reset=True):
    out = (ctypes.c_ulonglong * 16)()
    assert L.gcn_debug_gemm_step_stamps(out, int(reset)) == 0
    return list(out)


def stamps(reset=True):
    out = (ctypes.c_ulonglong * 8)()
    assert L.gcn_debug_gemm_stamps(out, int(reset)) == 0
    return list(out)


for scheme in ("bf16x3", "h2"):
    for name, kw in (("plain", {}), ("bias+relu+drop.5", dict(bias=bias, relu=True, dropout_p=0.5, seed=3))):
        S.set_gemm_scheme(scheme)
        for _ in range(2):
            S.gemm_xw256(X, W, x_bound=xb, **kw)
        torch.cuda.synchronize()
        stamps()
        step_stamps()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        reps = 5
        for _ in range(reps):
            S.gemm_xw256(X, W, x_bound=xb, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        k, s, n, life, waves, f2, f8 = stamps()[:7]
        print(f"{scheme:7s} {name:18s} {ms:6.2f} ms | per wave-tile: K loop {k / n:8.0f} cyc, store section {s / n:8.0f} cyc "
              f"| wave lifetime {life / waves:10.0f} cyc = {life / waves / (ms * 1e-3 * reps) / 1e9 * reps:5.2f} GHz-equivalent "
              f"| tiles per wave {n / waves:6.1f} | K share {k / life:.3f} store share {s / life:.3f} "
              f"| K steps 0-1: {f2 / n / 2:6.0f} cyc/step, steps 2-7: {(f8 - f2) / n / 6:6.0f}, steps 8-15: {(k - f8) / n / 8:6.0f}", flush=True)
        ph = step_stamps()
        names = ["-", "barrier wait", "DMA issue", "frag reads + MFMA group 0", "groups 1-3", "groups 4-7", "vmcnt wait", "X read + split"]
        for base, label in ((0, "step 4 (even)"), (8, "step 5 (odd) ")):
            print("      " + label + ": " + ", ".join(f"{names[i]} {ph[base + i] / n:6.0f}" for i in range(1, 8))
                  + f" | sum {sum(ph[base + 1:base + 8]) / n:6.0f}", flush=True)
S.set_gemm_scheme("bf16x3")
