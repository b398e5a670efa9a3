"""Row pitch of X / Y against the time of gcn_gemm_xw256_f32_h2 (both kernel forms): with a lane
per row, a 1-KiB pitch sends the 32 rows of one load instruction to the same few L2 channels."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd import _native
L = _native.lib()
dev = torch.device("cuda:0")
M = int(os.environ.get("GEMM_M", 10_000_000))
W = torch.randn(256, 256, device=dev) / 16
ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = None
for ldx, ldy in ((256, 256), (288, 256), (256, 288), (288, 288), (272, 272), (260, 260), (320, 320)):
    torch.manual_seed(0)
    Xb = torch.empty(M, ldx, device=dev); Xb[:, :256] = torch.randn(M, 256, device=dev)
    Yb = torch.empty(M, ldy, device=dev)
    b = Xb[:, :256].abs().max().reshape(1)
    def run():
        rc = L.gcn_gemm_xw256_f32_h2(Xb.data_ptr(), ldx, None, W.data_ptr(), 256, Yb.data_ptr(), ldy, M,
                                     b.data_ptr(), None, None, ws.data_ptr(), ws.numel(),
                                     torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    res = []
    for v in ("n", "w"):
        os.environ["GCN_GEMM_H2_VARIANT"] = v
        res.append("%s %.2f ms" % (v, t(run)))
        if ref is None: ref = Yb[:50000, :256].clone()
        else: assert torch.equal(Yb[:50000, :256], ref)
    print("ldx %d ldy %d: %s" % (ldx, ldy, "  ".join(res)), flush=True)
    del Xb, Yb
