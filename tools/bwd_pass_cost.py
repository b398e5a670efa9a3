#!/usr/bin/env python3
"""Time of the fused backward passes (gcn_relu_dropout_backward_colsum /
gcn_log_softmax_backward_colsum) at C4 size for dense and row-sparse incoming gradients."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd.spmm import backward_with_colsum

dev = torch.device("cuda:0")
n, F = 10_000_000, 256
out = torch.relu(torch.randn(n, F, device=dev))
logp = torch.log_softmax(torch.randn(n, F, device=dev), 1)


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


for dens in (1.0, 0.16, 0.05):
    g = torch.randn(n, F, device=dev)
    if dens < 1.0:
        g *= (torch.rand(n, 1, device=dev) < dens)
    print(f"rows non-zero {dens:4.2f}: relu/dropout bwd {t_of(lambda: backward_with_colsum(g, out, 2.0)):6.3f} ms   "
          f"log_softmax bwd {t_of(lambda: backward_with_colsum(g, logp, log_softmax=True)):6.3f} ms   "
          f"plain colsum {t_of(lambda: backward_with_colsum(g)):6.3f} ms", flush=True)
    del g
