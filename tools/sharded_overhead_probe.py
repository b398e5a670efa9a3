"""Local overhead of the sharded code path: one rank (RCCL, world size 1 — no exchange partner) on
a graph the size of one rank's share of C4 at N = 8 (1.25 M vertices / 12.5 M sampled pairs),
epoch by epoch, next to the single-GPU paths on the same graph.  What the sharded path adds per
epoch beyond its kernels (Python, small torch ops, count all-gathers) caps the N = 8 epoch time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch.distributed as dist
from pygcn_amd import GCN, CSRGraph
from pygcn_amd.functional import nll_loss
from pygcn_amd.sharded import ShardedGCN, ShardedGraph
from pygcn_amd.utils import rmat_graph
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, e, F = int(os.environ.get("N", 1_250_000)), int(os.environ.get("E", 12_500_000)), 256
x = torch.randn(n, F, device=dev); labels = torch.randint(0, F, (n,), device=dev)
idx = torch.arange(n * 140 // 2708, device=dev)
def timed(fn, k=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, e0.elapsed_time(e1) / k
res = {}
for exchange in ("halo", "allgather"):
    sg = ShardedGraph.from_rmat(n, e, 0, 1, dev, seed=42, perm_seed=43, exchange=exchange)
    torch.manual_seed(1); model = GCN(F, F, F, dropout=0.5).to(dev); sm = ShardedGCN(model, sg)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    def step():
        model.train(); opt.zero_grad(set_to_none=True)
        loss = sm.nll_loss(sm(x, sg).float(), labels, idx); loss.backward(); sm.allreduce_grads(); opt.step()
    res["sharded/" + exchange] = timed(step)
    if exchange == "halo":
        lt = labels[idx]
        def step_rows():
            model.train(); opt.zero_grad(set_to_none=True)
            loss = sm.nll_loss(sm(x, sg, rows=idx).float(), lt); loss.backward(); sm.allreduce_grads(); opt.step()
        res["sharded/halo, rows="] = timed(step_rows)
rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
g = CSRGraph(rowptr, col, val, (n, n))
torch.manual_seed(1); model = GCN(F, F, F, dropout=0.5).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
def one_node():
    model.train(); opt.zero_grad(set_to_none=True)
    nll_loss(model(x, g, rows=idx).float(), labels[idx]).backward(); opt.step()
def layers():
    model.train(); opt.zero_grad(set_to_none=True)
    torch.nn.functional.nll_loss(model(x, g)[idx].float(), labels[idx]).backward(); opt.step()
res["single GPU, rows="] = timed(one_node); res["single GPU, upstream lines"] = timed(layers)
for k, (wall, gpu) in res.items():
    print(f"{k:32s} wall {wall:7.3f} ms/epoch   (device timeline {gpu:7.3f} ms)", flush=True)
dist.destroy_process_group()
