"""Rehearsal of the sharded path on ONE MI355X: two ranks share cuda:0 and run the real HIP local
product; the collectives go over gloo staged through host memory (RCCL refuses two ranks on one
device, and a gpurun box has a single GPU).  The sharded result must match the single-GPU HIP
result and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, n_edges, out_dir, exchange):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.sharded import ShardedGCN, ShardedGraph
    from pygcn_amd.utils import rmat_graph
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # stage device tensors through the host for gloo (rehearsal only)
    real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce

    def ag(out, inp, group=None):
        o, i = out.cpu(), inp.cpu()
        real_ag(o, i, group=group)
        out.copy_(o)

    def ar(t, op=dist.ReduceOp.SUM, group=None):
        c = t.cpu()
        real_ar(c, op=op, group=group)
        t.copy_(c)
    dist.all_gather_into_tensor, dist.all_reduce = ag, ar
    import pygcn_amd.sharded as sh
    real_p2p = sh._p2p_round

    def p2p(sends, recvs, group):   # grouped isend/irecv staged through the host
        hs = [(t.cpu(), peer) for t, peer in sends]
        hr = [(torch.empty(t.shape, dtype=t.dtype), peer) for t, peer in recvs]
        real_p2p(hs, hr, group)
        for (t, _), (h, _) in zip(recvs, hr):
            t.copy_(h)
    sh._p2p_round = p2p
    try:
        dev = torch.device("cuda:0")
        F = 256
        rowptr, col, val = rmat_graph(n, n_edges, seed=5, device="cpu")
        sg = ShardedGraph.from_global_csr(rowptr.to(dev), col.to(dev), val.to(dev), n, rank, world,
                                          exchange=exchange)
        recv, full = sg.exchange_rows()
        assert (recv < 0.8 * full) if exchange == "halo" else recv == full
        x = torch.from_numpy(np.random.default_rng(1).standard_normal((n, F)).astype(np.float32))
        labels = torch.from_numpy(np.random.default_rng(2).integers(0, F, n))
        torch.manual_seed(42)
        model = GCN(F, F, F, dropout=0.0).to(dev)
        smodel = ShardedGCN(model, sg)
        model.train()
        logp = smodel(x[sg.r0:sg.r1].to(dev), sg)
        loss = smodel.nll_loss(logp, labels[sg.r0:sg.r1].to(dev))
        loss.backward()
        smodel.allreduce_grads()
        gl = smodel.global_loss(loss)

        # single-GPU HIP result on the whole graph (same seed -> same parameters)
        torch.manual_seed(42)
        ref = GCN(F, F, F, dropout=0.0).to(dev)
        ref.train()
        g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))
        rl = ref(x.to(dev), g)
        rloss = torch.nn.functional.nll_loss(rl, labels.to(dev))
        rloss.backward()

        def close(a, b, what, rel=1e-5):
            err = (a.double() - b.double()).abs().max().item()
            assert err <= rel * b.double().abs().max().item(), f"rank {rank}: {what} {err:.3e}"
        close(logp, rl[sg.r0:sg.r1], "logp block")
        assert abs(gl - rloss.item()) <= 1e-5 * abs(rloss.item())
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            close(p.grad, q.grad, k + ".grad", rel=2e-5)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["halo", "allgather"])
def test_two_ranks_on_one_gpu_match_single_gpu(tmp_path, exchange):
    assert torch.cuda.is_available()
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), 60000, 600000, str(tmp_path), exchange), nprocs=2,
             join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]
