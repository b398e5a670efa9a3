import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd import CSRGraph, spmm_csr
from pygcn_amd.utils import rmat_graph
dev = torch.device("cuda:0")
n, e, F = 10_000_000, 100_000_000, 256
rowptr, col, val = rmat_graph(n, e, device=dev)
A = CSRGraph(rowptr, col, val, (n, n)); A.plan()
B = torch.randn(n, F, device=dev); X = torch.randn(n, F, device=dev); W = torch.randn(F, F, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def spmm_only(): spmm_csr(A, B)
def gemm_only(): torch.mm(X, W)
def serial(): spmm_csr(A, B); torch.mm(X, W)
def concurrent():
    with torch.cuda.stream(s1): spmm_csr(A, B)
    with torch.cuda.stream(s2): torch.mm(X, W)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
def concurrent_gemm_first():
    with torch.cuda.stream(s2): torch.mm(X, W)
    with torch.cuda.stream(s1): spmm_csr(A, B)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
for name, fn in (("spmm", spmm_only), ("gemm", gemm_only), ("serial", serial), ("concurrent spmm-first", concurrent), ("concurrent gemm-first", concurrent_gemm_first)):
    print(f"{name:24s} {timeit(fn):7.2f} ms", flush=True)
