#!/usr/bin/env python3
"""3 forward SpMM launches on the C4 graph (F=256 fp32) — workload for rocprofv3 --pmc passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd import CSRGraph, spmm_csr
from pygcn_amd.utils import rmat_graph
dev = torch.device("cuda:0")
n, e, F = 10_000_000, 100_000_000, int(os.environ.get("FEAT", 256))
rowptr, col, val = rmat_graph(n, e, device=dev)
A = CSRGraph(rowptr, col, val, (n, n))
B = torch.randn(n, F, device=dev)
if os.environ.get("DTYPE") == "bf16": B = B.bfloat16()
for _ in range(3): spmm_csr(A, B)
torch.cuda.synchronize()
