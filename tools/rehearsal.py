"""One-GPU REHEARSAL of the multi-GPU path — not a product path and not a measurement.

A gpurun box has a single MI355X and RCCL refuses two ranks on one device, so the N > 1 code path
(partition, halo exchange, pipelined product, gradient all-reduce) is rehearsed with all ranks
sharing cuda:0 and the collectives going over gloo, staged through host memory.  Lives outside the
package on purpose (test scaffolding: it patches torch.distributed process-wide); used by
`bench.py --rehearsal` and tests/test_sharded_gpu.py only."""
import torch
import torch.distributed as dist


def install_host_staging():
    """Route the collectives pygcn_amd.sharded uses through host copies (gloo moves host memory)."""
    from pygcn_amd import sharded as sh
    real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce
    real_begin, real_end = sh._p2p_begin, sh._p2p_end

    def ag(out, inp, group=None):
        o, i = out.cpu(), inp.cpu()
        real_ag(o, i, group=group)
        out.copy_(o)

    def ar(t, op=dist.ReduceOp.SUM, group=None):
        c = t.cpu()
        real_ar(c, op=op, group=group)
        t.copy_(c)

    def begin(sends, recvs, group):
        hs = [(t.cpu(), peer) for t, peer in sends]
        hr = [(torch.empty(t.shape, dtype=t.dtype), peer) for t, peer in recvs]
        return [(real_begin(hs, hr, group), recvs, hr, hs)]

    def end(pending):
        for works, recvs, hr, _ in pending:
            real_end(works)
            for (t, _), (h, _) in zip(recvs, hr):
                t.copy_(h)
    dist.all_gather_into_tensor, dist.all_reduce = ag, ar
    sh._p2p_begin, sh._p2p_end = begin, end
