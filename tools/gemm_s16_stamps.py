"""gemm_xw256_s16_kernel under -DGEMM_PROFILE_STAMPS (tools/build_gemm_variant.sh st16 -DGEMM_PROFILE_STAMPS; run with
GCN_SPMM_LIB=build/variants/libgcn_st16.so): the clock the chip holds (wave lifetime in shader cycles / wall time of
the launch) and the cycles a wave spends waiting at its stage barriers."""
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
S.set_gemm_scheme("bf16x3")


def stamps(reset=True):
    out = (ctypes.c_ulonglong * 8)()
    assert L.gcn_debug_gemm_stamps(out, int(reset)) == 0
    return list(out)


for name, kw in (("plain", {}), ("bias+relu+drop.5", dict(bias=bias, relu=True, dropout_p=0.5, seed=3))):
    for _ in range(2):
        S.gemm_xw256(X, W, **kw)
    torch.cuda.synchronize()
    stamps()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    reps = 5
    for _ in range(reps):
        S.gemm_xw256(X, W, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    bw, flush, n, life, waves = stamps()[:5]
    print(f"{os.path.basename(os.environ.get('GCN_SPMM_LIB', 'product')):18s} {name:18s} {ms:6.2f} ms | wave lifetime {life / waves:10.0f} cyc = "
          f"{life / waves / (ms * 1e-3) / 1e9:5.2f} GHz | tiles per wave {n / waves:6.1f} | cycles per tile {life / n:7.0f} "
          f"(per stage {life / n / 8:5.0f}; matrix pipe 3072) | barrier wait per stage {bw / n / 8:5.0f} | final store section {flush / waves:6.0f}", flush=True)
