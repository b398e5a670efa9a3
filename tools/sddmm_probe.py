"""Time of gcn_sddmm_csr (gradient of the adjacency values) on the C4 graph, F = 256 fp32, next to the
forward product on the same graph."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd import CSRGraph, spmm_csr
from pygcn_amd.spmm import sddmm_csr
from pygcn_amd.utils import rmat_graph
dev = torch.device("cuda:0")
n = int(os.environ.get("N", 10_000_000))
rowptr, col, val = rmat_graph(n, 10 * n, device=dev)
g = CSRGraph(rowptr, col, val, (n, n)); g.plan()
G = torch.randn(n, 256, device=dev); B = torch.randn(n, 256, device=dev)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ts, tp = t(lambda: sddmm_csr(g, G, B)), t(lambda: spmm_csr(g, B))
alg = g.nnz * (256 * 4 + 4 + 4) + n * (256 * 4 + 4)
print(f"nnz {g.nnz}: sddmm {ts:.2f} ms = {g.nnz / ts / 1e6:.2f} GEdge/s = {alg / ts / 1e6 / 8000:.3f} of 8 TB/s (gather model)   "
      f"spmm {tp:.2f} ms = {g.nnz / tp / 1e6:.2f} GEdge/s")
