"""Workload of the GEMM counter passes (profiles/r0N_gemm_pmc.md): three launches of the plain
fp32 256 -> 256 product and three of the forward-epilogue instantiation (bias + ReLU + dropout) at
M = 10^7, with a bound supplied (no reduction pass in between)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd.spmm import gemm_xw256
dev = torch.device("cuda:0")
X = torch.randn(10_000_000, 256, device=dev); W = torch.randn(256, 256, device=dev) * 0.06
bias = torch.randn(256, device=dev) * 0.1
b = X.abs().max().reshape(1)
for _ in range(3): gemm_xw256(X, W, x_bound=b)
for _ in range(3): gemm_xw256(X, W, x_bound=b, bias=bias, relu=True, dropout_p=0.5, seed=1234)
torch.cuda.synchronize()
