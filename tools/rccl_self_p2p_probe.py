"""Does RCCL accept a grouped isend/irecv to the calling rank itself (world size 1)?  If yes, the
batch_isend_irecv code path of pygcn_amd/sharded.py can be exercised on the real backend on a
one-GPU box.  Run under a short timeout."""
import os, sys, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("init ok", flush=True)
t = torch.arange(1 << 20, device=dev, dtype=torch.float32)
r = torch.empty_like(t)
ops = [dist.P2POp(dist.isend, t, 0), dist.P2POp(dist.irecv, r, 0)]
print("posting", flush=True)
for w in dist.batch_isend_irecv(ops):
    w.wait()
torch.cuda.synchronize()
print("self p2p equal:", bool(torch.equal(t, r)), flush=True)
dist.destroy_process_group()
