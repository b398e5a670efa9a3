"""Round 4: the fp32-equivalent (three bf16 parts) GEMMs in the round-3 pipeline, next to the scaled
two-part fp16 scheme and hipBLASLt — one process, interleaved rounds.
  * gcn_gemm_xw256_f32_b3 plain == round 1's gcn_gemm_xw256_f32 BIT FOR BIT (same MFMA order);
  * every store variant (bias / ReLU / dropout ½ / dropout p / backward mask, row lists) against the
    h2 kernel: same masks, values within the two schemes' rounding;
  * gcn_gemm_atg256_f32_b3 against an fp64 product;
  * times at M = GEMM_M (default 10^7)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygcn_amd import _native, spmm as S   # noqa: E402

dev = torch.device("cuda:0")
L = _native.lib()


def old_b3(X, W):
    Y = torch.empty((X.shape[0], 256), device=dev)
    ws = torch.empty(L.gcn_gemm_xw256_workspace_bytes(), dtype=torch.uint8, device=dev)
    rc = L.gcn_gemm_xw256_f32(X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), Y.data_ptr(), Y.stride(0),
                              X.shape[0], ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "old")
    return Y


def run(scheme, *a, **k):
    S.set_gemm_scheme(scheme)
    try:
        return S.gemm_xw256(*a, **k)
    finally:
        S.set_gemm_scheme("bf16x3")


ok = True
for M in (777, 1_000_003):
    g = torch.Generator(device=dev).manual_seed(M)
    X = torch.randn(M, 256, generator=g, device=dev) * 3
    W = torch.randn(256, 256, generator=g, device=dev) * 0.1
    b = torch.randn(256, generator=g, device=dev)
    ref = X.double() @ W.double()
    sc = ref.abs().max().item()
    Yn, Yo = run("bf16x3", X, W), old_b3(X, W)
    same = torch.equal(Yn, Yo)
    ok &= same
    ymax = torch.zeros(1, device=dev)
    Ym = run("bf16x3", X, W, y_absmax=ymax)
    ok &= torch.equal(Ym, Yn) and float(ymax) == float(Yn.abs().max())
    print(f"M={M}: b3 == round-1 kernel bitwise: {same}; err vs fp64 b3 {(Yn.double()-ref).abs().max().item()/sc:.2e} "
          f"h2 {(run('h2', X, W).double()-ref).abs().max().item()/sc:.2e} torch {((X@W).double()-ref).abs().max().item()/sc:.2e}; "
          f"y_absmax exact: {float(ymax) == float(Yn.abs().max())}", flush=True)
    for name, kw in (("bias", dict(bias=b)), ("bias+relu", dict(bias=b, relu=True)),
                     ("drop .5", dict(bias=b, relu=True, dropout_p=0.5, seed=1234, row_base=77)),
                     ("drop .3", dict(bias=b, relu=True, dropout_p=0.3, seed=99))):
        a3, a2 = run("bf16x3", X, W, **kw), run("h2", X, W, **kw)
        # same keep decisions wherever the pre-activation is not within rounding of zero
        pre = ref + b.double()
        clear = (pre.abs() > 1e-4 * sc) if kw.get("relu") else torch.ones_like(pre, dtype=torch.bool)
        mask_same = bool((((a3 != 0) == (a2 != 0)) | ~clear).all())
        d = (a3.double() - a2.double()).abs().max().item() / max(a2.abs().max().item(), 1e-30)
        ok &= mask_same and d < 2e-6
        print(f"   {name:10s} masks equal {mask_same}  |b3 - h2| {d:.2e}", flush=True)
    rows = torch.randperm(M, generator=g, device=dev)[: M // 3].to(torch.int32)
    h = torch.randn(M, 256, generator=g, device=dev)
    mr = torch.randint(0, M, (rows.numel(),), generator=g, device=dev).to(torch.int32)
    a3 = run("bf16x3", X, W, rows=rows, mask_src=h, mask_rows=mr, mask_scale=2.0)
    want = torch.where(h[mr.long()] > 0, (X[rows.long()].double() @ W.double()) * 2.0, 0.0)
    e = (a3.double() - want).abs().max().item() / sc
    ok &= e < 2e-6
    print(f"   rows + mask vs fp64: {e:.2e}", flush=True)
    # weight gradient
    G = torch.randn(M, 256, generator=g, device=dev) * 0.01
    ra = torch.randint(0, M, (M // 2,), generator=g, device=dev).to(torch.int32)
    rg = torch.randint(0, M, (M // 2,), generator=g, device=dev).to(torch.int32)
    for sch in ("bf16x3", "h2"):
        S.set_gemm_scheme(sch)
        gw = S.weight_grad_rows(X, G, ra, rg)
        S.set_gemm_scheme("bf16x3")
        r64 = X[ra.long()].double().t() @ G[rg.long()].double()
        summ = (X[ra.long()].abs().double().t() @ G[rg.long()].abs().double()).max().item()
        print(f"   atg {sch}: err/result {(gw.double()-r64).abs().max().item()/r64.abs().max().item():.2e} "
              f"err/summands {(gw.double()-r64).abs().max().item()/summ:.2e} "
              f"(torch: {((X[ra.long()].t() @ G[rg.long()]).double()-r64).abs().max().item()/r64.abs().max().item():.2e})", flush=True)
        ok &= (gw.double() - r64).abs().max().item() <= 3e-7 * summ
print("CORRECT" if ok else "MISMATCH", flush=True)

M = int(os.environ.get("GEMM_M", 10_000_000))
X = torch.randn(M, 256, device=dev)
W = torch.randn(256, 256, device=dev) * 0.06
bias = torch.randn(256, device=dev)
Gd = torch.randn(M, 256, device=dev)
Y = torch.empty(M, 256, device=dev)
xb = X.abs().max().reshape(1)
gb = Gd.abs().max().reshape(1)


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def sch(name, fn):
    def f():
        S.set_gemm_scheme(name)
        try:
            return fn()
        finally:
            S.set_gemm_scheme("bf16x3")
    return f


cases = [
    ("plain", lambda: S.gemm_xw256(X, W, x_bound=xb)),
    ("bias+relu+drop.5", lambda: S.gemm_xw256(X, W, x_bound=xb, bias=bias, relu=True, dropout_p=0.5, seed=5)),
    ("masked", lambda: S.gemm_xw256(X, W, x_bound=xb, mask_src=Gd, mask_scale=2.0)),
    ("atg all rows", lambda: S.weight_grad_rows(X, Gd, a_bound=xb, g_bound=gb)),
]
for rnd in range(3):
    line = [f"round {rnd}:"]
    for name, fn in cases:
        line.append(f"{name}: b3 {t(sch('bf16x3', fn)):.2f} / h2 {t(sch('h2', fn)):.2f} ms;")
    line.append(f"old b3 {t(lambda: old_b3(X, W)):.2f}; torch.mm {t(lambda: torch.mm(X, W, out=Y)):.2f}")
    print(" ".join(line), flush=True)
