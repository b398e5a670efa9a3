import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd.utils import rmat_graph
dev = "cuda:0"
n, e = 10_000_000, 100_000_000
rowptr, col, val = rmat_graph(n, e, device=dev)
indeg = torch.bincount(col.long(), minlength=n)
s, _ = torch.sort(indeg, descending=True)
cs = torch.cumsum(s, 0).double() / col.numel()
for k in (256, 1024, 4096, 16384, 32768, 65536, 262144, 1048576):
    print(f"top {k:8d} columns ({k/1024:7.1f} MiB of B rows): {100*cs[k-1].item():5.1f} % of stored entries; "
          f"min in-degree in set {int(s[k-1])}")
deg = (rowptr[1:] - rowptr[:-1])
print("rows with 1 entry: %.1f %%" % (100 * (deg == 1).double().mean().item()))
print("entries in rows > 256: %.1f %%" % (100 * deg[deg > 256].sum().item() / col.numel()))
