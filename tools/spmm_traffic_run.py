#!/usr/bin/env python3
"""Workload for the PMC traffic passes (run under `rocprofv3 --pmc FETCH_SIZE` and, separately,
`--pmc WRITE_SIZE`): launches the SpMM kernel on
  1. a CALIBRATION matrix with a known byte count in the kernel's own access pattern — a random
     permutation matrix P (one stored entry per row): every row of B is gathered exactly once,
     every row of C written once; B and C are 4 GiB each, far beyond the 256 MiB Infinity Cache,
     so the counters must read n*(F*s+8+p) fetched and n*F*s written bytes;
  2. the graphs of the configuration:
       c4 (default)  fp32 F = 256, spmm_wide_kernel:  the C4 R-MAT graph (forward and transpose) and
                     the UNIFORM degree-10 graph of `bench.py`'s roofline_uniform (forward);
       c5            bf16 F = 128, spmm_narrow_kernel (256-byte rows — their own calibration: the
                     FETCH_SIZE factor of 1-KiB rows does not carry over): the C5
                     R-MAT graph (forward).
Dispatches are told apart by grid size in the counter CSV; sizes are printed as JSON."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import CSRGraph, spmm_csr  # noqa: E402
from pygcn_amd.utils import rmat_graph, uniform_graph  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")
reps = 3
F, dt, s = (256, torch.float32, 4) if config == "c4" else (128, torch.bfloat16, 2)
info = {"config": config, "F": F, "dtype": "f32" if s == 4 else "bf16",
        "kernel": "spmm_wide_kernel" if config == "c4" else "spmm_narrow_kernel"}


def grid_x(g):
    st = g.schedule_stats(dt)
    return (st["n_items"] + st["n_chunks"] + 3) // 4, st


n = (4 if config == "c4" else 16) * 1024 * 1024          # 4 GiB of rows either way
gen = torch.Generator(device=dev)
gen.manual_seed(1)
perm = torch.randperm(n, generator=gen, device=dev).to(torch.int32)
P = CSRGraph(torch.arange(n + 1, device=dev, dtype=torch.int32), perm, torch.ones(n, device=dev), (n, n))
B = torch.randn(n, F, device=dev).to(dt)
for _ in range(reps):
    spmm_csr(P, B)
torch.cuda.synchronize()
gx, st = grid_x(P)
info["calib"] = {"n": n, "nnz": n, "grid_x": gx,
                 "fetch_bytes_expected": n * (F * s + 8 + 4) + 8 * st["n_items"],
                 "write_bytes_expected": n * F * s}
del P, B, perm
torch.cuda.empty_cache()

if config == "c4":
    n, e = 10_000_000, 100_000_000
    rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
    A = CSRGraph(rowptr, col, val, (n, n))
    B = torch.randn(n, F, device=dev)
    for _ in range(reps):
        spmm_csr(A, B)
    At = A.t()
    for _ in range(reps):
        spmm_csr(At, B)
    U = CSRGraph(*uniform_graph(n, e // n, seed=46, device=dev), (n, n))
    for _ in range(reps):
        spmm_csr(U, B)
    torch.cuda.synchronize()
    graphs = (("c4_fwd", A), ("c4_bwd", At), ("c4_uniform_fwd", U))
else:
    n, e = 50_000_000, 1_000_000_000
    rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
    A = CSRGraph(rowptr, col, val, (n, n))
    del rowptr, col, val
    B = torch.randn(n, F, device=dev).to(dt)
    for _ in range(reps):
        spmm_csr(A, B)
    torch.cuda.synchronize()
    graphs = (("c5_fwd", A),)
for name, gr in graphs:
    gx, st = grid_x(gr)
    p = 8 if gr.nnz >= 2 ** 31 - 1 else 4
    info[name] = {"n": n, "nnz": gr.nnz, "grid_x": gx,
                  "algorithmic_bytes": gr.nnz * (F * s + 8) + n * (F * s + p), **st}
print("TRAFFIC_INFO " + json.dumps(info))
