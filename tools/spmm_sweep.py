#!/usr/bin/env python3
"""Sweep the runtime schedule parameters of the SpMM on the C4 (or C3) graph in one process."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import CSRGraph, spmm_csr
from pygcn_amd.utils import rmat_graph

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=10_000_000)
ap.add_argument("--edges", type=int, default=100_000_000)
ap.add_argument("--feat", type=int, default=256)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--grid", default="64:256,32:256,128:256,256:256,64:128,64:512,64:1024,128:512")
a = ap.parse_args()
dev = torch.device("cuda:0")
rowptr, col, val = rmat_graph(a.nodes, a.edges, device=dev)
B = torch.randn(a.nodes, a.feat, device=dev)
if a.dtype == "bf16":
    B = B.bfloat16()
nnz = col.numel()
s = 2 if a.dtype == "bf16" else 4
alg = nnz * (a.feat * s + 8) + a.nodes * (a.feat * s + 4)
for spec in a.grid.split(","):
    ic, lt = [int(v) for v in spec.split(":")]
    g = CSRGraph(rowptr, col, val, (a.nodes, a.nodes), item_cost=ic, long_thresh=lt)
    g.plan()
    for _ in range(2):
        spmm_csr(g, B)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(); spmm_csr(g, B); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    st = g.schedule_stats()
    print(f"item_cost {ic:4d} long_thresh {lt:5d}: {t:8.3f} ms  {nnz/t/1e6:7.3f} GEdge/s  "
          f"{alg/t/1e6:8.1f} GB/s alg  items {st['n_items']} chunks {st['n_chunks']}", flush=True)
