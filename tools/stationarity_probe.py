"""Why does the C4 epoch time depend on how many steps were run?
Per step: wall ms (synchronised), the backward products' durations, the count of non-zero rows of
each layer's grad_pre (the row bitmap's count) and the share of hidden units that are dead.
Usage: python tools/stationarity_probe.py [steps]   (writes one JSON line per step)."""
import importlib
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import GCN, CSRGraph  # noqa: E402
from pygcn_amd.utils import rmat_graph  # noqa: E402

spmm_mod = importlib.import_module("pygcn_amd.spmm")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
n, e, feat = 10_000_000, 100_000_000, 256
rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
g = CSRGraph(rowptr, col, val, (n, n))
g.plan(), g.t().plan()
x = torch.randn(n, feat, generator=torch.Generator(device=dev).manual_seed(44), device=dev)
labels = torch.randint(0, feat, (n,), generator=torch.Generator(device=dev).manual_seed(45), device=dev)
model = GCN(feat, feat, feat, dropout=0.5).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
idx = torch.arange(int(n * 140 / 2708), device=dev)

counts = []
orig = spmm_mod._grad_pre_and_bias


def spy(*a, **kw):
    r = orig(*a, **kw)
    if r[2] is not None:
        counts.append(r[2][1])
    return r


spmm_mod._grad_pre_and_bias = spy
for step in range(steps):
    rec = []
    spmm_mod.set_timing_records(rec)
    counts.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.train()
    opt.zero_grad(set_to_none=True)
    out = model(x, g)
    loss = F.nll_loss(out[idx], labels[idx])
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    spmm_mod.set_timing_records(None)
    with torch.no_grad():
        h = torch.relu(torch.mm(x[:200000], model.gc1.weight) + model.gc1.bias)   # no adjacency: a proxy
        dead_units = float((h.max(0).values <= 0).float().mean())
    print(json.dumps({"step": step, "ms": round(ms, 2), "loss": round(loss.item(), 5),
                      "spmm_ms": [(t, round(a.elapsed_time(b), 2)) for t, a, b, _ in rec],
                      "grad_pre_nnz_rows_l2_l1": [int(c.item()) for c in counts],
                      "w1_absmax": round(float(model.gc1.weight.abs().max()), 4),
                      "b1_mean": round(float(model.gc1.bias.mean()), 4),
                      "dead_unit_share_proxy": round(dead_units, 4)}), flush=True)
