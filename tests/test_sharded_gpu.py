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


def _worker(rank, world, port, n, n_edges, out_dir, exchange, build="global"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.sharded import ShardedGCN, ShardedGraph
    from pygcn_amd.utils import rmat_graph
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tools.rehearsal import install_host_staging
    install_host_staging()      # gloo moves host memory: stage device tensors through the host
    try:
        dev = torch.device("cuda:0")
        F = 256
        if build == "global":
            rowptr, col, val = rmat_graph(n, n_edges, seed=5, device="cpu")
            sg = ShardedGraph.from_global_csr(rowptr.to(dev), col.to(dev), val.to(dev), n, rank,
                                              world, exchange=exchange)
        else:   # shard-local: each rank generates its own rows on the device, Âᵀ by triplet exchange
            sg = ShardedGraph.from_rmat(n, n_edges, rank, world, dev, seed=5, exchange=exchange)
            rowptr, col, val = rmat_graph(n, n_edges, seed=5, device=dev)   # (reference only)
        assert sg.overlap == (exchange == "halo")
        recv, full = sg.exchange_rows()
        assert (recv < 0.8 * full) if exchange == "halo" else recv == full
        x = torch.from_numpy(np.random.default_rng(1).standard_normal((n, F)).astype(np.float32))
        labels = torch.from_numpy(np.random.default_rng(2).integers(0, F, n))
        torch.manual_seed(42)
        model = GCN(F, F, F, dropout=0.0).to(dev)
        smodel = ShardedGCN(model, sg)
        model.train()
        h1 = {}
        hook = model.gc1.register_forward_hook(lambda m, i, o: h1.__setitem__("sharded", o.detach()))
        logp = smodel(x[sg.r0:sg.r1].to(dev), sg)
        hook.remove()
        loss = smodel.nll_loss(logp, labels[sg.r0:sg.r1].to(dev))
        loss.backward()
        smodel.allreduce_grads()
        gl = smodel.global_loss(loss)

        # single-GPU HIP result on the whole graph (same seed -> same parameters)
        torch.manual_seed(42)
        ref = GCN(F, F, F, dropout=0.0).to(dev)
        ref.train()
        g = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), (n, n))

        # (composed from its two layers by hand: the hidden activation and its gradient are needed
        #  below, and `ref(x, g)` runs as ONE autograd node whose inside no module hook sees)
        hid = ref.gc1(x.to(dev), g, relu=True)
        h1["ref"] = hid.detach()
        hid.register_hook(lambda gr: h1.__setitem__("ref_grad", gr.detach()))
        rl = ref.gc2(hid, g, log_softmax=True)
        rloss = torch.nn.functional.nll_loss(rl, labels.to(dev))
        rloss.backward()

        def close(a, b, what, rel=1e-5, extra=0.0):
            err = (a.double() - b.double()).abs().max().item()
            scale = b.double().abs().max().item()
            assert err <= rel * scale + extra, \
                f"rank {rank}: {what} {err:.3e} vs scale {scale:.3e} (+{extra:.3e})"
        close(logp, rl[sg.r0:sg.r1], "logp block")
        close(h1["sharded"], h1["ref"][sg.r0:sg.r1], "hidden block")
        assert abs(gl - rloss.item()) <= 1e-5 * abs(rloss.item())
        # The two paths associate layer 1 differently (sharded: (Â_r·X)·W, single GPU with the loss
        # on all rows: Â·(X·W)), so pre-activations within rounding of zero land on different sides
        # of the ReLU.  The loss is not differentiable there: each such element (r, j) moves
        # gc1.weight.grad[:, j] by grad_h1[r, j] · (Â·X)[r, :] and gc1.bias.grad[j] by grad_h1[r, j].
        # Those elements are identified and priced exactly; everything else must agree to 2e-5
        # (fp32 sums over 60 000 rows in two association orders, each within 1e-5 of exact).
        from pygcn_amd.spmm import spmm_csr
        hs, hr = h1["sharded"], h1["ref"][sg.r0:sg.r1]
        flips = (hs > 0) != (hr > 0)
        assert int(flips.sum()) <= 1e-4 * flips.numel()
        assert (torch.maximum(hs, hr) * flips).max().item() <= 1e-5 * hr.max().item()
        gh = h1["ref_grad"][sg.r0:sg.r1].abs() * flips
        zmax = spmm_csr(g, x.to(dev))[sg.r0:sg.r1].abs().amax(1, keepdim=True)
        budget = torch.stack([(gh * zmax).sum(0), gh.sum(0)]).cpu()
        dist.all_reduce(budget)
        extra = {"gc1.weight": budget[0].max().item(), "gc1.bias": budget[1].max().item()}
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            close(p.grad, q.grad, k + ".grad", rel=2e-5, extra=extra.get(k, 0.0))

        # the one-node path per rank (rows= : static halo of gradient rows, transpose block) against
        # the single-GPU one-node path on the union of the ranks' rows; unsorted rows on purpose
        if exchange == "halo":
            gen = torch.Generator().manual_seed(100 + rank)
            idx_local = torch.randperm(sg.n_local, generator=gen)[: sg.n_local // 15].to(dev)
            parts = [torch.empty(0, dtype=torch.int64) for _ in range(world)]
            dist.all_gather_object(parts, (idx_local.cpu() + sg.r0))
            idx_global = torch.cat(parts).to(dev)
            model.zero_grad(set_to_none=True)
            xl = x[sg.r0:sg.r1].to(dev)
            out_rows = smodel(xl, sg, rows=idx_local)
            assert type(out_rows.grad_fn).__name__ == "ShardedGCN2RowsFunctionBackward"
            lb = labels[sg.r0:sg.r1].to(dev)[idx_local]
            loss_r = smodel.nll_loss(out_rows, lb)
            loss_r.backward()
            smodel.allreduce_grads()
            ref.zero_grad(set_to_none=True)
            rr = ref(x.to(dev), g, rows=idx_global)
            rloss_r = torch.nn.functional.nll_loss(rr, labels.to(dev)[idx_global])
            rloss_r.backward()
            mine = slice(int(sum(p.numel() for p in parts[:rank])), int(sum(p.numel() for p in parts[:rank + 1])))
            close(out_rows, rr[mine], "rows= log-probabilities")
            assert abs(smodel.global_loss(loss_r) - rloss_r.item()) <= 1e-5 * abs(rloss_r.item())
            for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                close(p.grad, q.grad, "rows= " + k + ".grad", rel=2e-5)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,build", [("halo", "global"), ("allgather", "global"),
                                            ("halo", "local")])
def test_two_ranks_on_one_gpu_match_single_gpu(tmp_path, exchange, build):
    assert torch.cuda.is_available()
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), 60000, 600000, str(tmp_path), exchange, build),
             nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _compress_worker(rank, world, port, n, n_edges, out_dir, dtype_name):
    """ShardedGraph(compress_hidden=True) on the device: the halo rows of the hidden activation
    travel as bitmask + values through the HIP pack / unpack kernels (gcn_rows_pack_*,
    gcn_rows_unpack).  The format is lossless, so a training step with dropout must give the SAME
    BITS as the dense exchange — log-probabilities and every gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygcn_amd import GCN
    from pygcn_amd.sharded import ShardedGCN, ShardedGraph
    import pygcn_amd.spmm as S
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tools.rehearsal import install_host_staging
    install_host_staging()
    try:
        dev = torch.device("cuda:0")
        dtype = getattr(torch, dtype_name)
        F = 256 if dtype == torch.float32 else 128
        x = torch.from_numpy(np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)).to(dtype)
        labels = torch.from_numpy(np.random.default_rng(2).integers(0, F, n))
        results, calls = {}, []
        real_pack, real_unpack = S.rows_pack, S.rows_unpack
        import pygcn_amd.sharded as SH
        SH.rows_pack = lambda *a, **k: (calls.append("pack"), real_pack(*a, **k))[1]
        SH.rows_unpack = lambda *a, **k: (calls.append("unpack"), real_unpack(*a, **k))[1]
        for compress in (False, True):
            sg = ShardedGraph.from_rmat(n, n_edges, rank, world, dev, seed=5, exchange="halo",
                                        compress_hidden=compress)
            torch.manual_seed(42)
            model = GCN(F, F, F, dropout=0.5).to(dev).to(dtype)
            smodel = ShardedGCN(model, sg)
            model.train()
            torch.manual_seed(99)           # (the fused dropout's seed comes from the device generator)
            logp = smodel(x[sg.r0:sg.r1].to(dev), sg)
            smodel.nll_loss(logp.float(), labels[sg.r0:sg.r1].to(dev)).backward()
            smodel.allreduce_grads()
            results[compress] = (logp.detach().clone(), [p.grad.clone() for p in model.parameters()],
                                 dict(sg.last_recv_bytes))
            if compress:
                dense = sg.halo.n_halo * F * x.element_size()
                assert 0 < sg.last_recv_bytes["fwd"] < 0.5 * dense      # relu + dropout 0.5: <= 25 % kept
        assert "pack" in calls and "unpack" in calls                     # the HIP kernels carried it
        (la, ga, _), (lb, gb, _) = results[False], results[True]
        assert torch.equal(la, lb), f"rank {rank}: log-probabilities differ"
        for a, b in zip(ga, gb):
            assert torch.equal(a, b), f"rank {rank}: a gradient differs"
        if dtype == torch.float32:
            # the halo rows of h come from OTHER ranks — under the scaled fp16 scheme
            # ("h2") the receiving rank's own bound of max|h| does not cover them.  Rank 1's rows are
            # 64 x larger than rank 0's; every rank passes ITS OWN maximum as the bound: the result
            # must be finite and equal the dense exchange (the halo block takes its own bound)
            before = S.gemm_scheme()
            S.set_gemm_scheme("h2")
            try:
                gen = torch.Generator(device=dev).manual_seed(7 + rank)
                h = torch.relu(torch.randn(sg.n_local, F, generator=gen, device=dev)) * (64.0 if rank == 1 else 1.0)
                w = torch.randn(F, F, generator=torch.Generator(device=dev).manual_seed(3), device=dev) * 0.06
                bound = h.abs().max().reshape(1)
                got = sg.product_hidden(h, w, h_bound=bound)
                want = sg.product(S._dense_forward(h, w, bound))
                assert torch.isfinite(got).all(), f"rank {rank}: the halo GEMM overflowed its fp16 parts"
                err = (got - want).abs().max().item()
                assert err <= 1e-5 * want.abs().max().item(), (rank, err)
            finally:
                S.set_gemm_scheme(before)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_compressed_hidden_exchange_same_bits_as_dense(tmp_path, dtype_name):
    import torch.multiprocessing as mp
    mp.spawn(_compress_worker, args=(2, _free_port(), 60000, 600000, str(tmp_path), dtype_name),
             nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _bf16_worker(rank, world, port, n, n_edges, out_dir):
    """Config C5's storage (bf16, F = 128) through the sharded path: two ranks on one GPU against
    the single-GPU HIP model on the same bf16 parameters and inputs.  Both run bf16 pipelines with
    one rounding per stage; they differ by summation order and by the ranks' split of the
    gradient sums, so the comparison is at bf16 resolution."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.sharded import ShardedGCN, ShardedGraph
    from pygcn_amd.utils import rmat_graph
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tools.rehearsal import install_host_staging
    install_host_staging()
    try:
        dev = torch.device("cuda:0")
        F = 128
        sg = ShardedGraph.from_rmat(n, n_edges, rank, world, dev, seed=7, exchange="halo")
        rowptr, col, val = rmat_graph(n, n_edges, seed=7, device=dev)
        x = torch.from_numpy(np.random.default_rng(3).standard_normal((n, F)).astype(np.float32)).bfloat16()
        labels = torch.from_numpy(np.random.default_rng(4).integers(0, F, n))
        torch.manual_seed(7)
        model = GCN(F, F, F, dropout=0.0).to(dev).bfloat16()
        smodel = ShardedGCN(model, sg)
        model.train()
        seen = []
        import pygcn_amd.spmm as S
        orig = S.layer_gemm
        S.layer_gemm = lambda *a, **k: (seen.append(a[0].dtype), orig(*a, **k))[1]
        try:
            logp = smodel(x[sg.r0:sg.r1].to(dev), sg)
        finally:
            S.layer_gemm = orig
        assert seen == [torch.bfloat16]          # the first layer took the reassociated branch
        loss = smodel.nll_loss(logp.float(), labels[sg.r0:sg.r1].to(dev))
        loss.backward()
        smodel.allreduce_grads()
        torch.manual_seed(7)
        ref = GCN(F, F, F, dropout=0.0).to(dev).bfloat16()
        ref.train()
        rl = ref(x.to(dev), CSRGraph(rowptr, col, val, (n, n)))
        rloss = torch.nn.functional.nll_loss(rl.float(), labels.to(dev))
        rloss.backward()

        def close(a, b, what, rel):
            err = (a.double() - b.double()).abs().max().item()
            scale = b.double().abs().max().item()
            assert err <= rel * scale, f"rank {rank}: {what} {err:.3e} vs scale {scale:.3e}"
        close(logp, rl[sg.r0:sg.r1], "logp block", 2.0 ** -6)
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert p.grad is not None and torch.isfinite(p.grad).all()
            close(p.grad, q.grad, k + ".grad", 2.0 ** -4)
        # the one-node path per rank at bf16 (compact rows through index_select + hipBLASLt)
        idx_local = torch.arange(sg.n_local // 10, device=dev)
        parts = [None] * world
        dist.all_gather_object(parts, (idx_local.cpu() + sg.r0))
        idx_global = torch.cat(parts).to(dev)
        model.zero_grad(set_to_none=True)
        out_rows = smodel(x[sg.r0:sg.r1].to(dev), sg, rows=idx_local)
        assert type(out_rows.grad_fn).__name__ == "ShardedGCN2RowsFunctionBackward"
        smodel.nll_loss(out_rows.float(), labels[sg.r0:sg.r1].to(dev)[idx_local]).backward()
        smodel.allreduce_grads()
        ref.zero_grad(set_to_none=True)
        rr = ref(x.to(dev), CSRGraph(rowptr, col, val, (n, n)), rows=idx_global)
        torch.nn.functional.nll_loss(rr.float(), labels.to(dev)[idx_global]).backward()
        lo = int(sum(p.numel() for p in parts[:rank]))
        close(out_rows, rr[lo:lo + idx_local.numel()], "rows= logp", 2.0 ** -6)
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            close(p.grad, q.grad, "rows= " + k + ".grad", 2.0 ** -4)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_bf16_storage(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_bf16_worker, args=(2, _free_port(), 40000, 400000, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _nccl_worker(rank, world, port, n, n_edges, out_dir):
    """The sharded path on the REAL backend (RCCL), world size 1 — the only RCCL execution a
    one-GPU box allows: communicator init bound to the device, all_gather_into_tensor (degree
    counts, halo count matrix, row all-gather), all_reduce (loss count, gradient bucket), barrier,
    the zero-peer point-to-point rounds, and grouped isend / irecv to the rank itself through
    `_p2p_begin` / `_p2p_end`, end to end against the plain single-GPU model."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygcn_amd import GCN, CSRGraph
    from pygcn_amd.sharded import ShardedGCN, ShardedGraph
    from pygcn_amd.utils import rmat_graph
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        F = 256
        rowptr, col, val = rmat_graph(n, n_edges, seed=5, device=dev)
        x = torch.randn(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        labels = torch.randint(0, F, (n,), device=dev,
                               generator=torch.Generator(device=dev).manual_seed(2))
        idx = torch.arange(n // 20, device=dev)
        torch.manual_seed(42)
        ref = GCN(F, F, F, dropout=0.0).to(dev)
        ref.train()
        g = CSRGraph(rowptr, col, val, (n, n))
        rl = ref(x, g)
        rloss = torch.nn.functional.nll_loss(rl[idx], labels[idx])
        rloss.backward()
        for exchange in ("halo", "allgather", "rccl-allgather"):
            sg = ShardedGraph.from_rmat(n, n_edges, rank, world, dev, seed=5, exchange=exchange)
            assert sg._rccl_gather == (exchange == "rccl-allgather")
            assert sg.bounds == [0, n] and sg.nnz_local == int(col.numel())
            assert torch.equal(sg.At.col, g.t().col) and torch.equal(sg.At.val, g.t().val)
            torch.manual_seed(42)
            model = GCN(F, F, F, dropout=0.0).to(dev)
            smodel = ShardedGCN(model, sg)
            model.train()
            logp = smodel(x, sg)
            loss = smodel.nll_loss(logp, labels, idx)
            loss.backward()
            smodel.allreduce_grads()
            dist.barrier()
            assert abs(smodel.global_loss(loss) - rloss.item()) <= 1e-5 * abs(rloss.item())
            err = (logp - rl).abs().max().item()
            assert err <= 1e-5 * rl.abs().max().item(), (exchange, err)
            for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                e = (p.grad - q.grad).abs().max().item()
                assert e <= 2e-5 * q.grad.abs().max().item(), (exchange, k, e)
            if exchange == "halo":
                # the forward exchange of the hidden layer switched to RCCL's all-gather collective
                # (dist.all_gather_into_tensor, in place on the padded buffer) on the same graph:
                # same log-probabilities as the halo form
                sg.set_forward_exchange("rccl-allgather")
                with torch.no_grad():
                    again = smodel(x, sg)
                assert (again - rl).abs().max().item() <= 1e-5 * rl.abs().max().item()
                sg.set_forward_exchange("halo")
                # the self-validation of a multi-GPU run degenerates gracefully at one rank
                from pygcn_amd import selfcheck as sc
                assert sc.overlap_selftest(sg, [x])["agrees"] is None
                assert sc.link_rate(dev, 1 << 20)["gb_per_s"] is None
            if exchange == "halo":      # the one-node path: its collectives and its static P2P round on RCCL
                model.zero_grad(set_to_none=True)
                out_rows = smodel(x, sg, rows=idx)
                assert type(out_rows.grad_fn).__name__ == "ShardedGCN2RowsFunctionBackward"
                loss_r = smodel.nll_loss(out_rows, labels[idx])
                loss_r.backward()
                smodel.allreduce_grads()
                assert abs(smodel.global_loss(loss_r) - rloss.item()) <= 1e-5 * abs(rloss.item())
                assert (out_rows - rl[idx]).abs().max().item() <= 1e-5 * rl.abs().max().item()
                for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                    e = (p.grad - q.grad).abs().max().item()
                    assert e <= 2e-5 * q.grad.abs().max().item(), ("rows=", k, e)
        # RCCL point-to-point through the path's own grouped-transfer helpers: at world size 1 the
        # only peer is the rank itself, which RCCL accepts — two messages to the same peer in one
        # group (the layout of the row-sparse gradient exchange), a zero-length one skipped on both
        # sides, and a product launched between post and wait (the pipelined exchange)
        from pygcn_amd import spmm_csr
        from pygcn_amd.sharded import _p2p_begin, _p2p_end
        pos = torch.arange(77, device=dev, dtype=torch.int32)
        rows_out = torch.randn(5000, F, device=dev)
        empty = torch.empty((0, F), device=dev)
        got_pos, got_rows = torch.empty_like(pos), torch.empty_like(rows_out)
        pending = _p2p_begin([(pos, 0), (rows_out, 0), (empty, 0)],
                             [(got_pos, 0), (got_rows, 0), (torch.empty_like(empty), 0)], None)
        overlapped = spmm_csr(g, x)                       # runs while the transfers are in flight
        _p2p_end(pending)
        assert torch.equal(got_pos, pos) and torch.equal(got_rows, rows_out)
        assert torch.equal(overlapped, spmm_csr(g, x))
        open(os.path.join(out_dir, "ok_nccl"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_path_on_rccl_world_size_one(tmp_path):
    assert torch.cuda.is_available()
    import torch.multiprocessing as mp
    mp.spawn(_nccl_worker, args=(1, _free_port(), 60000, 600000, str(tmp_path)), nprocs=1, join=True)
    assert os.listdir(tmp_path) == ["ok_nccl"]
