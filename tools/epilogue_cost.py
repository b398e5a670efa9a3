#!/usr/bin/env python3
"""Cost of the fused epilogues on the forward SpMM (C4 graph, one process = one box):
plain, + bias, + bias/ReLU/dropout, + bias/log_softmax, for fp32 F=256 (wide kernel) and
bf16 F=128 / fp32 F=64 (narrow kernel)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import CSRGraph, spmm_csr
from pygcn_amd.utils import rmat_graph

dev = torch.device("cuda:0")
n = 10_000_000
rowptr, col, val = rmat_graph(n, 100_000_000, device=dev)
g = CSRGraph(rowptr, col, val, (n, n))
g.plan()
nnz = col.numel()


def t_of(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[2]


for F, dt in ((256, torch.float32), (128, torch.bfloat16), (64, torch.float32)):
    B = torch.randn(n, F, device=dev).to(dt)
    bias = torch.randn(F, device=dev)
    out = torch.empty(n, F, device=dev, dtype=dt)
    for name, kw in (("plain", {}), ("bias", dict(bias=bias)),
                     ("bias+relu+dropout", dict(bias=bias, relu=True, dropout_p=0.5, seed=7)),
                     ("bias+log_softmax", dict(bias=bias, log_softmax=True)),
                     ("bias+absmax", dict(bias=bias, c_absmax=torch.zeros(1, device=dev))),
                     ("plain again", {})):
        t = t_of(lambda: spmm_csr(g, B, out=out, **kw))
        print(f"F {F:4d} {str(dt):15s} {name:20s} {t:8.3f} ms  {nnz / t / 1e6:7.3f} GEdge/s", flush=True)
    del B, out
