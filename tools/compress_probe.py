"""Local cost of the compressed hidden-layer exchange (pygcn_amd/sharded.py: pack = bitmask + non-zero
values of the rows to send, unpack = expand on arrival) at the size of one rank's halo at C4 / 8:
1.86 M rows x 256 fp32, 75 % zeros (ReLU + dropout 0.5)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pygcn_amd.sharded import pack_bits, unpack_bits
dev = torch.device("cuda:0")
m, F = int(os.environ.get("ROWS", 1_860_000)), 256
h = torch.relu(torch.randn(m, F, device=dev)) * (torch.rand(m, F, device=dev) > 0.5)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def pack():
    mask = h != 0
    return pack_bits(mask), h[mask], mask.sum(1)
bits, vals, per_row = pack()
def unpack():
    out = torch.zeros((m, F), dtype=h.dtype, device=dev)
    out[unpack_bits(bits, F)] = vals
    return out
assert torch.equal(unpack(), h)
from pygcn_amd.spmm import rows_pack, rows_unpack
# the HIP path reads the requested rows in place from the rank's block (no gathered copy)
n_local = 2 * m
blk = torch.relu(torch.randn(n_local, F, device=dev)) * (torch.rand(n_local, F, device=dev) > 0.5)
idx = torch.randperm(n_local, device=dev)[:m].sort().values
hb, ho, hv = rows_pack(blk, idx)
assert torch.equal(rows_unpack(hb, hv, F), blk[idx])
t_gather = t(lambda: blk.index_select(0, idx))
t_hp, t_hu = t(lambda: rows_pack(blk, idx)), t(lambda: rows_unpack(hb, hv, F))
print(f"HIP kernels: pack (gather fused) {t_hp:.2f} ms, unpack {t_hu:.2f} ms; the dense exchange's gather alone {t_gather:.2f} ms")
print(f"rows {m}: dense {m*F*4/1e9:.2f} GB -> compressed {(bits.numel()*4 + vals.numel()*4)/1e9:.2f} GB; "
      f"pack {t(pack):.2f} ms, unpack {t(unpack):.2f} ms (torch ops)")
