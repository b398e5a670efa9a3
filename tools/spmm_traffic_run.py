#!/usr/bin/env python3
"""Workload for the PMC traffic passes (run under `rocprofv3 --pmc FETCH_SIZE` and, separately,
`--pmc WRITE_SIZE`): launches the SpMM kernel on
  1. a CALIBRATION matrix with a known byte count in the kernel's own access pattern — a random
     permutation matrix P (one stored entry per row): every 1-KiB row of B is gathered exactly
     once, every row of C written once; B and C are 4 GiB each, far beyond the 256 MiB Infinity
     Cache, so the counters must read n*(F*4+8+4) fetched and n*F*4 written bytes;
  2. the C4 graph of bench.py (forward A·B and backward A^T·G).
Dispatches are told apart by grid size in the counter CSV; sizes are printed as JSON.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import CSRGraph, spmm_csr  # noqa: E402
from pygcn_amd.utils import rmat_graph  # noqa: E402

dev = torch.device("cuda:0")
F, reps = 256, 3
info = {}

n = 4 * 1024 * 1024
g = torch.Generator(device=dev)
g.manual_seed(1)
perm = torch.randperm(n, generator=g, device=dev).to(torch.int32)
P = CSRGraph(torch.arange(n + 1, device=dev, dtype=torch.int32), perm,
             torch.ones(n, device=dev), (n, n))
B = torch.randn(n, F, device=dev)
for _ in range(reps):
    spmm_csr(P, B)
torch.cuda.synchronize()
st = P.schedule_stats()
info["calib"] = {"n": n, "nnz": n, "grid_x": (st["n_items"] + st["n_chunks"] + 3) // 4,
                 "fetch_bytes_expected": n * (F * 4 + 8 + 4) + 8 * st["n_items"],
                 "write_bytes_expected": n * F * 4}
del P, B, perm

n, e = 10_000_000, 100_000_000
rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
A = CSRGraph(rowptr, col, val, (n, n))
B = torch.randn(n, F, device=dev)
for _ in range(reps):
    spmm_csr(A, B)
At = A.t()
for _ in range(reps):
    spmm_csr(At, B)
torch.cuda.synchronize()
for name, gr in (("c4_fwd", A), ("c4_bwd", At)):
    st = gr.schedule_stats()
    info[name] = {"n": n, "nnz": gr.nnz, "grid_x": (st["n_items"] + st["n_chunks"] + 3) // 4,
                  "algorithmic_bytes": gr.nnz * (F * 4 + 8) + n * (F * 4 + 4), **st}
print("TRAFFIC_INFO " + json.dumps(info))
