"""Cache-policy experiments of the SpMM gather on the C4 graph (10^7 vertices, F = 256 fp32): every
variant is a libgcn_*.so built by tools/build_spmm_variants.sh and runs in its own process
(GCN_SPMM_LIB).  Variants named hub* read bit 31 of a column index as "hub column — keep in L2":
this script sets it for the HUB_K columns with the most stored entries (default 2048 = 2 MiB of
rows) and every other row is gathered with the streaming (nt) policy.
Usage: python tools/spmm_policy_sweep.py build/variants/*.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for path in sys.argv[1:]:
        env = dict(os.environ, GCN_SPMM_LIB=os.path.abspath(path))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.basename(path)], env=env, check=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from pygcn_amd import CSRGraph
from pygcn_amd.spmm import spmm_csr
from pygcn_amd.utils import rmat_graph
name = sys.argv[2]
dev = torch.device("cuda:0")
n, e, F = int(os.environ.get("NODES", 10_000_000)), int(os.environ.get("EDGES", 100_000_000)), 256
rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
B = torch.randn(n, F, device=dev)
ref_rows = torch.randint(0, n, (2000,), device=dev)
def ref(rows):          # fp64 reference of a few rows
    out = torch.zeros(rows.numel(), F, dtype=torch.float64, device=dev)
    for i, r in enumerate(rows.tolist()):
        a, b = int(rowptr[r]), int(rowptr[r + 1])
        out[i] = (val[a:b, None].double() * B[col[a:b].long()].double()).sum(0)
    return out
want = ref(ref_rows[:200])
share = ""
if "hub" in name:
    K = int(os.environ.get("HUB_K", 2048))
    deg = torch.bincount(col.long(), minlength=n)
    hubs = torch.topk(deg, K).indices
    is_hub = torch.zeros(n, dtype=torch.bool, device=dev); is_hub[hubs] = True
    tag = is_hub[col.long()]
    share = f"  hub columns {K}: {tag.float().mean().item():.3f} of the entries"
g = CSRGraph(rowptr, col, val, (n, n))
if "hub" in name:       # (tagged AFTER the constructor's range check, in the array the plan points at)
    g.col.copy_(torch.where(tag, g.col | torch.tensor(-2**31, dtype=torch.int32, device=dev), g.col))
    col = g.col & 0x7fffffff
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
C = spmm_csr(g, B)
err = ((C[ref_rows[:200]].double() - want).abs().max() / want.abs().max()).item()
ms = [t(lambda: spmm_csr(g, B)) for _ in range(3)]
print(f"{name:28s} " + "  ".join(f"{m:.2f}" for m in ms) + f" ms   err {err:.1e}{share}", flush=True)
