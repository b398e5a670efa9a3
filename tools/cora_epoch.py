#!/usr/bin/env python3
"""Config C2: Cora 2-layer GCN (1433 -> 16 -> 7) epoch time on one MI355X through the HIP SpMM,
eager and replayed from a hipGraph (the epoch is launch-latency bound)."""
import os, sys, time
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gin
from pygcn_amd import GCN, as_graph
from pygcn_amd.utils import load_data

dev = torch.device("cuda:0")
adj, _, _, idx_train, _, _ = load_data()
adj, idx_train = adj.to(dev), idx_train.to(dev)
x = torch.from_numpy(gin.cora_features()).to(dev)
y = torch.from_numpy(gin.cora_labels()).to(dev)
torch.manual_seed(42)
model = GCN(1433, 16, 7, dropout=0.5).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4, capturable=True)
g = as_graph(adj); g.plan(); g.t().plan()

def step():
    model.train()
    opt.zero_grad(set_to_none=False)
    out = model(x, g)
    loss = F.nll_loss(out[idx_train], y[idx_train])
    loss.backward()
    opt.step()
    return loss

for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("eager: %.3f ms/epoch" % ((time.perf_counter() - t0) / 200 * 1e3))

try:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): graph.replay()
    torch.cuda.synchronize()
    print("hipGraph replay: %.3f ms/epoch (loss %.4f)" % ((time.perf_counter() - t0) / 200 * 1e3, loss.item()))
except Exception as e:
    print("hipGraph capture failed:", repr(e)[:300])
