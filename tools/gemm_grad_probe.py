"""Time of the gather-fused weight-gradient kernel (gcn_gemm_atg256_f32) and the row-list GEMM against
the torch / hipBLASLt forms they replace, at the bench's backward shapes (|R2| = 1.6 M of 10^7 rows),
plus the bf16 128 -> 128 GEMM at config C5's height — one process, interleaved rounds."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd import spmm as S
dev = torch.device("cuda:0")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N, n2 = 10_000_000, 1_620_000
A = torch.randn(N, 256, device=dev); G = torch.randn(N, 256, device=dev) * 0.01
W = torch.randn(256, 256, device=dev)
rows = torch.sort(torch.randperm(N, device=dev)[:n2]).values
r32 = rows.to(torch.int32); rpad = S.padded_row_list(rows)
ab, gb = A.abs().max().reshape(1), G.abs().max().reshape(1)
Ac, Gc = A[rows].contiguous(), G[rows].contiguous()
for rnd in range(3):
    print("round %d | grad_W: kernel+lists %.2f  kernel compact %.2f  gathers+K-split bmm %.2f  (gathers alone %.2f)"
          " | grad_in: kernel+list %.2f  gather+kernel %.2f | dense 1e7 rows: kernel %.2f  K-split bmm %.2f" % (
        rnd, t(lambda: S.weight_grad_rows(A, G, rpad, rpad, ab, gb, n_list=n2)), t(lambda: S.weight_grad_rows(Ac, Gc, None, None, ab, gb)),
        t(lambda: (S.set_gemm_scheme("bf16x3"), S._weight_grad(A.index_select(0, rows), G.index_select(0, rows)), S.set_gemm_scheme("h2"))),
        t(lambda: (A.index_select(0, rows), G.index_select(0, rows))),
        t(lambda: S.gemm_xw256(G, W, gb, rows=r32)), t(lambda: S.gemm_xw256(G.index_select(0, rows), W, gb)),
        t(lambda: S.weight_grad_rows(A, G, None, None, ab, gb), 3),
        t(lambda: (S.set_gemm_scheme("bf16x3"), S._weight_grad(A, G), S.set_gemm_scheme("h2")), 3)), flush=True)
del A, G, Ac, Gc
torch.cuda.empty_cache()
M = int(os.environ.get("BF16_M", 50_000_000))
X = torch.randn(M, 128, device=dev).bfloat16(); Wb = torch.randn(128, 128, device=dev).bfloat16()
Y = torch.empty(M, 128, device=dev, dtype=torch.bfloat16)
for rnd in range(3):
    a, b = t(lambda: S.gemm_bf16(X, Wb)), t(lambda: torch.mm(X, Wb, out=Y))
    print("round %d | bf16 [%d,128]x[128,128]: kernel %.2f ms (%.2f TB/s)   torch.mm %.2f ms" % (
        rnd, M, a, 2 * M * 256 / a / 1e9, b), flush=True)
