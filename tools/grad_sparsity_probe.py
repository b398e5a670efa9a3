import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import importlib
from pygcn_amd import GCN, CSRGraph
spmm_mod = importlib.import_module("pygcn_amd.spmm")
from pygcn_amd.utils import rmat_graph
dev = torch.device("cuda:0")
n, e, feat = 10_000_000, 100_000_000, 256
rowptr, col, val = rmat_graph(n, e, device=dev)
g = CSRGraph(rowptr, col, val, (n, n))
x = torch.randn(n, feat, device=dev); labels = torch.randint(0, feat, (n,), device=dev)
model = GCN(feat, feat, feat, dropout=0.5).to(dev)
idx = torch.arange(int(n * 140 / 2708), device=dev)
orig = spmm_mod.spmm_csr
def spy(graph, B, **kw):
    if kw.get("tag") == "bwd":
        nz = (B != 0).any(1)
        print(f"bwd SpMM operand: {100*nz.float().mean().item():.2f} % of rows non-zero; "
              f"{100*(B != 0).float().mean().item():.2f} % of elements", flush=True)
        cols_hit = nz[graph.col.long()].float().mean().item()
        print(f"   -> {100*cols_hit:.2f} % of the stored entries reference a non-zero row", flush=True)
    return orig(graph, B, **kw)
spmm_mod.spmm_csr = spy
model.train()
out = model(x, g)
F.nll_loss(out[idx], labels[idx]).backward()
