#!/usr/bin/env python3
"""Config C5 at full size on ONE MI355X: R-MAT scale 26 with rejection to 5*10^7 vertices, 10^9
sampled directed pairs, seeded permutation, dedupe, +I, row-normalized; bf16 storage F = 128, fp32
values and accumulation.  Everything (generation, CSR, transpose) is built on the device; only the
row pointer visits the host for the native planner."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import CSRGraph, spmm_csr
from pygcn_amd.utils import rmat_graph

n = int(os.environ.get("C5_NODES", 50_000_000)); e = int(os.environ.get("C5_EDGES", 1_000_000_000))
F = 128
dev = torch.device("cuda:0")
def sync(): torch.cuda.synchronize(); return time.perf_counter()
t0 = sync()
rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
t1 = sync(); print(f"generate+dedupe+normalize: {t1-t0:.1f} s, nnz {col.numel()}", flush=True)
torch.cuda.empty_cache()
g = CSRGraph(rowptr, col, val, (n, n)); g.plan()
t2 = sync(); print(f"plan: {t2-t1:.1f} s {g.schedule_stats()}", flush=True)
gt = g.t(); gt.plan()
t3 = sync(); print(f"transpose+plan: {t3-t2:.1f} s", flush=True)
torch.cuda.empty_cache()
nnz = g.nnz
res = {"n": n, "sampled_edges": e, "nnz": nnz, "F": F, "gen_s": round(t1-t0,1), "transpose_s": round(t3-t2,1)}
for dtype, s in ((torch.bfloat16, 2), (torch.float32, 4)):
    ones = torch.ones(n, F, device=dev, dtype=dtype)
    out = spmm_csr(g, ones)
    err = float((out.float() - 1).abs().max()); del out
    B = torch.randn(n, F, device=dev).to(dtype)
    for name, gr in (("fwd", g), ("bwd", gt)):
        for _ in range(2): spmm_csr(gr, B)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record(); spmm_csr(gr, B); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[2]
        alg = nnz * (F * s + 8) + n * (F * s + 4)
        res[f"{str(dtype).split('.')[-1]}_{name}"] = {"ms": round(t, 2), "gedges": round(nnz / t / 1e6, 3),
                                                     "alg_GBps": round(alg / t / 1e6, 1), "frac_8TBps": round(alg / t / 8e9, 3)}
    res[f"{str(dtype).split('.')[-1]}_rowsum_err"] = err
    del B, ones; torch.cuda.empty_cache()
res["peak_mem_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
print("C5_RESULT " + json.dumps(res), flush=True)
