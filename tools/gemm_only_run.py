import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd.spmm import gemm_xw256
dev = torch.device("cuda:0")
X = torch.randn(10_000_000, 256, device=dev); W = torch.randn(256, 256, device=dev)
for _ in range(3): gemm_xw256(X, W)
torch.cuda.synchronize()
