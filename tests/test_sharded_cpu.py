"""Multi-process tests of the row-block sharded path on CPU (gloo, world_size 2 and 3).

The partition, column remap, padded all-gather, autograd exchange and gradient all-reduce are the
product's code (pygcn_amd/sharded.py); only the rank-local product is supplied by the test as the
CPU oracle (tests may use the oracle; the product never does).  The sharded result must equal
the unsharded oracle result on the whole graph."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _CpuGraph:
    """Oracle-backed stand-in for CSRGraph (CPU tensors)."""

    def __init__(self, rowptr, col, val, shape, **_):
        self.rowptr, self.col, self.val, self.shape = rowptr, col, val, tuple(shape)
        self.nnz = int(col.numel())


def _cpu_spmm(graph, B, bias=None, relu=False, out=None, tag="fwd", B2=None):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gcn_oracle
    if B2 is not None:
        B = torch.cat([B, B2])
    assert B.shape[0] == graph.shape[1]
    y = gcn_oracle.spmm_csr(graph.rowptr.numpy(), graph.col.numpy(), graph.val.numpy(),
                            B.detach().numpy())
    if bias is not None:
        y = y + bias.detach().numpy()
    if relu:
        y = np.maximum(y, 0)
    return torch.from_numpy(y)


def _cpu_bwd(grad_out, out, relu, scale, want_bias):
    """CPU stand-in for the HIP backward pass (mask through out > 0, bias gradient)."""
    if relu:
        grad_out = torch.where(out > 0, grad_out * scale, torch.zeros_like(grad_out))
    return grad_out, (grad_out.sum(0) if want_bias else None), None


def _cpu_bwd_with_hint(grad_out, out, relu, scale, want_bias):
    """Same, plus the row bitmap / count the HIP pass produces (pygcn_amd.spmm.row_bitmap is pure
    torch), so the bitmap branch of the row-sparse gradient exchange runs on CPU too."""
    from pygcn_amd.spmm import row_bitmap
    g, gb, _ = _cpu_bwd(grad_out, out, relu, scale, want_bias)
    return g, gb, (row_bitmap(g) if g.shape[0] else None)


def _worker(rank, world, port, n, n_edges, out_dir, exchange, bitmap_hint=False, build="global",
            overlap=True, static_rows=False, compress=False, fwd_switch=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import gcn_oracle
    from pygcn_amd import GCN
    from pygcn_amd.sharded import ShardedGCN, ShardedGraph
    from pygcn_amd.utils import rmat_graph
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fin, nhid, ncls = 24, 32, 16
        rowptr, col, val = rmat_graph(n, n_edges, seed=5, device="cpu")   # same on every rank
        kw = dict(exchange=exchange, graph_factory=_CpuGraph, spmm_fn=_cpu_spmm, overlap=overlap,
                  bwd_fn=_cpu_bwd_with_hint if bitmap_hint else _cpu_bwd, compress_hidden=compress)
        if build == "global":
            sg = ShardedGraph.from_global_csr(rowptr, col, val, n, rank, world, **kw)
        else:
            # shard-local construction: this rank generates only its own rows, the transpose
            # blocks come from the triplet exchange — must give exactly the blocks of the global build
            import pygcn_amd.utils as U
            passes, real_chunks = [], U.rmat_edge_chunks
            U.rmat_edge_chunks = lambda *a_, **k_: (passes.append(1), real_chunks(*a_, **k_))[1]
            try:
                sg = ShardedGraph.from_rmat(n, n_edges, rank, world, "cpu", seed=5, **kw)
            finally:
                U.rmat_edge_chunks = real_chunks
            # ONE replay of the edge stream per rank; rows that changed owner between the provisional
            # uniform blocks and the nnz-balanced ones travelled point to point
            assert len(passes) == 1 and sg.setup_stats["edge_stream_passes"] == 1
            assert sg.setup_stats["entries_received_in_rebalance"] >= 0
            ref = ShardedGraph.from_global_csr(rowptr, col, val, n, rank, world, **kw)
            assert sg.bounds == ref.bounds
            for a, b in ((sg.A, ref.A), (sg.At, ref.At)):
                assert torch.equal(a.rowptr.long(), b.rowptr.long()) and torch.equal(a.col, b.col)
                assert torch.equal(a.val, b.val) and a.shape == b.shape
            del ref
        assert sg.overlap == (overlap and exchange == "halo" and world > 1)
        recv, full = sg.exchange_rows()
        assert recv <= full and (exchange != "halo") == (recv == full and sg.halo is None)
        assert sg._rccl_gather == (exchange == "rccl-allgather")
        if fwd_switch is not None:       # a halo-mode graph whose FORWARD exchange is an all-gather
            sg.set_forward_exchange(fwd_switch)
        assert sg.bounds[0] == 0 and sg.bounds[-1] == n and sg.n_local == sg.r1 - sg.r0
        x = torch.from_numpy(np.random.default_rng(1).standard_normal((n, fin)).astype(np.float32))
        labels = torch.from_numpy(np.random.default_rng(2).integers(0, ncls, n))
        train = torch.from_numpy(np.sort(np.random.default_rng(3).choice(n, n // 5, False)))
        torch.manual_seed(42)
        model = GCN(fin, nhid, ncls, dropout=0.0)
        smodel = ShardedGCN(model, sg)
        model.train()
        x_loc, y_loc = x[sg.r0:sg.r1], labels[sg.r0:sg.r1]
        idx_loc = train[(train >= sg.r0) & (train < sg.r1)] - sg.r0
        if static_rows:           # (flag on every rank alike: it selects the exchange protocol)
            smodel.declare_loss_rows(idx_loc)
        logp = smodel(x_loc, sg)
        if compress:
            # layer 2's input rows (ReLU output: about half zeros here) travelled as bitmask + values
            assert sg.compress_hidden
            dense_bytes = sg.halo.n_halo * nhid * 4
            assert sg.halo.n_halo == 0 or 0 < sg.last_recv_bytes["fwd"] < 0.75 * dense_bytes
        loss = smodel.nll_loss(logp, y_loc, idx_loc)
        loss.backward()
        smodel.allreduce_grads()

        # unsharded oracle on the whole graph
        a = gcn_oracle.CSR(rowptr.numpy(), col.numpy(), val.numpy(), (n, n))
        p = {k: v.detach().numpy() for k, v in model.state_dict().items()}
        ref_loss, fw, grads, _ = gcn_oracle.gcn2_loss_backward(x.numpy(), a, p, labels.numpy(),
                                                                train.numpy())

        def close(got, ref, what, rel=1e-5):
            err = np.abs(np.asarray(got, np.float64) - ref).max()
            assert err <= rel * np.abs(ref).max(), f"rank {rank}: {what} off by {err:.3e}"
        close(logp.detach().numpy(), fw["logp"][sg.r0:sg.r1], "logp block")
        gl = smodel.global_loss(loss)
        assert abs(gl - ref_loss) <= 1e-5 * abs(ref_loss), (gl, ref_loss)
        for k, v in grads.items():
            mod, name = k.split(".")
            close(getattr(getattr(model, mod), name).grad.numpy(), v, k + ".grad", rel=2e-5)
        # the feature block is a registered constant input in halo mode: its halo rows were
        # exchanged once; a second step exchanges nothing for layer 1; an in-place edit of the
        # features is noticed; an input that requires grad takes the general (exchanging) path and
        # gives the same parameter gradients
        if fwd_switch is not None:
            # the hidden layer's forward exchange moved EVERY remote row (the collective's volume),
            # the backward exchange is still the row-sparse halo
            assert sg.last_recv_bytes["fwd"] == (n - sg.n_local) * ncls * 4
            sent, dense = sg.halo_t.last_sparse_rows
            assert sent <= dense
        if exchange == "halo":
            # the layer-2 gradient rows travelled sparsely: only rows of the n/5 labelled vertices
            sent, dense = sg.halo_t.last_sparse_rows
            assert sg.sparse_grad_exchange and sent <= dense and (dense == 0 or sent < 0.5 * dense)
            assert sg.n_const_exchanges == 1 and sg.is_constant_input(x_loc)
            first = {k: v.grad.clone() for k, v in model.named_parameters()}
            model.zero_grad()
            smodel.nll_loss(smodel(x_loc, sg), y_loc, idx_loc).backward()
            smodel.allreduce_grads()
            assert sg.n_const_exchanges == 1
            # count exchanges (all-gather + host read) of the row-sparse gradient exchange: one per
            # step by default, ONE IN TOTAL once the loss rows are declared (static structure)
            assert sg.halo_t.n_count_exchanges == (1 if static_rows else 2)
            for k, v in model.named_parameters():
                assert torch.equal(v.grad, first[k]), k
            x_loc.mul_(1.0)
            smodel(x_loc, sg)
            assert sg.n_const_exchanges == 2
            model.zero_grad()
            xg = x_loc.clone().requires_grad_(True)
            smodel.nll_loss(smodel(xg, sg), y_loc, idx_loc).backward()
            smodel.allreduce_grads()
            assert sg.n_const_exchanges == 2 and xg.grad is not None
            for k, v in model.named_parameters():
                close(v.grad.numpy(), first[k].numpy().astype(np.float64), k + " (general path)", rel=2e-5)
        else:
            assert sg.n_const_exchanges == 0 and not sg.is_constant_input(x_loc)
        # blocks are nnz-balanced and cover every stored entry exactly once
        tot = torch.tensor([float(sg.nnz_local), float(sg.At.nnz)])
        dist.all_reduce(tot)
        assert int(tot[0]) == int(tot[1]) == int(col.numel())
        assert abs(sg.nnz_local - col.numel() / world) <= 0.15 * col.numel() / world + int(
            (rowptr[1:] - rowptr[:-1]).max())
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange", [(2, "halo"), (3, "halo"), (8, "halo"), (2, "allgather"),
                                            (3, "allgather"), (2, "rccl-allgather"), (3, "rccl-allgather")])
def test_sharded_gcn_matches_unsharded_oracle(world, exchange, tmp_path, oracle):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 4000, 30000, str(tmp_path), exchange), nprocs=world,
             join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


@pytest.mark.parametrize("world", [2, 4])
def test_shard_local_construction_matches_global_build(world, tmp_path, oracle):
    """ShardedGraph.from_rmat: no rank ever holds the whole matrix; blocks, bounds and the
    end-to-end result equal those of the global build (and the unsharded oracle)."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), 4000, 30000, str(tmp_path), "halo", False, "local"),
             nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


def test_unpipelined_exchange_gives_the_same_result(tmp_path, oracle):
    """overlap=False (exchange, then one product over [own | halo]) against the same oracle."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(3, _free_port(), 4000, 30000, str(tmp_path), "halo", False, "global",
                            False), nprocs=3, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1", "ok2"]


def test_sparse_gradient_exchange_with_bitmap_hint(tmp_path, oracle):
    """The backward exchange driven by the row bitmap of the fused backward pass (3 ranks)."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(3, _free_port(), 4000, 30000, str(tmp_path), "halo", True), nprocs=3,
             join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1", "ok2"]


@pytest.mark.parametrize("world", [2, 3, 4])
def test_declared_loss_rows_make_the_gradient_exchange_static(world, tmp_path, oracle):
    """ShardedGCN.declare_loss_rows: the layer-by-layer path's row-sparse backward exchange runs
    its count exchange once instead of every step — same gradients (oracle parity in the worker)."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), 4000, 30000, str(tmp_path), "halo", False, "global",
                            True, True), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


@pytest.mark.parametrize("world,mode", [(2, "rccl-allgather"), (3, "rccl-allgather"), (3, "allgather")])
def test_forward_exchange_switched_to_the_collective(world, mode, tmp_path, oracle):
    """ShardedGraph.set_forward_exchange: a halo-mode graph whose hidden-layer FORWARD exchange is
    the all-gather — through dist.all_gather_into_tensor (the north star's literal "RCCL all-gather of
    activations"; gloo here) or the grouped point-to-point round — while the constant-input halo
    and the row-sparse backward exchange stay as they are: oracle parity in the worker."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), 4000, 30000, str(tmp_path), "halo", False, "global",
                            True, False, False, mode), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_compressed_hidden_layer_exchange(world, tmp_path, oracle):
    """ShardedGraph(compress_hidden=True): the halo rows of a hidden activation travel as bitmask +
    non-zero values and meet the weight on arrival (product_hidden / ShardedHiddenLayerFunction);
    same forward values and gradients as the unsharded oracle, fewer bytes than dense rows."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), 4000, 30000, str(tmp_path), "halo", False, "global",
                            True, False, True), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


def test_partition_and_remap_are_consistent():
    sys.path.insert(0, ROOT)
    from pygcn_amd.sharded import partition_rows, remap_columns, row_block, transpose_row_block
    from pygcn_amd.utils import rmat_graph
    n = 2500
    rowptr, col, val = rmat_graph(n, 20000, seed=9, device="cpu")
    for world in (1, 2, 4, 8):
        b = partition_rows(rowptr, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == n and sorted(b) == b
        max_rows = max(b[i + 1] - b[i] for i in range(world))
        rc = remap_columns(col, b, max_rows).long()
        owner = rc // max_rows
        back = rc - owner * max_rows + torch.tensor(b)[owner]
        assert torch.equal(back, col.long())                  # remap is invertible
        assert bool(((rc % max_rows) < (torch.tensor(b)[owner + 1] - torch.tensor(b)[owner])).all())
        cover = 0
        for r in range(world):
            rp, c, v = row_block(rowptr, col, val, b[r], b[r + 1])
            assert rp[0] == 0 and rp[-1] == c.numel() == v.numel()
            cover += c.numel()
            rpt, ct, vt = transpose_row_block(rowptr, col, val, n, b[r], b[r + 1])
            assert rpt.numel() == b[r + 1] - b[r] + 1 and rpt[-1] == ct.numel()
            # dense check of the transposed block
            dense = torch.zeros(n, n)
            rows = torch.repeat_interleave(torch.arange(n), (rowptr[1:] - rowptr[:-1]).long())
            dense[rows, col.long()] = val
            blk = torch.zeros(b[r + 1] - b[r], n)
            trow = torch.repeat_interleave(torch.arange(b[r + 1] - b[r]),
                                           (rpt[1:] - rpt[:-1]).long())
            blk[trow, ct.long()] = vt
            assert torch.equal(blk, dense.t()[b[r]:b[r + 1]])
        assert cover == col.numel()


def _halo_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygcn_amd.sharded import HaloExchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bounds = [0, 5, 5, 12, 20]           # rank 1 owns nothing
        n = bounds[-1]
        # block structure: rank 0 references only its own rows; rank 2 references rows of 0 and 3;
        # rank 3 references row 19 (own) and row 0
        cols = {0: [0, 1, 4, 4, 2], 1: [], 2: [0, 19, 7, 3, 12, 12, 19], 3: [19, 0]}[rank]
        col = torch.tensor(cols, dtype=torch.int32)
        h = HaloExchange(col, bounds, rank, world)
        table = torch.arange(n, dtype=torch.float32).view(n, 1) * torch.tensor([[1.0, 10.0]])
        local = table[bounds[rank]:bounds[rank + 1]].contiguous()
        buf = torch.cat([local, h.exchange(local)])
        assert buf.shape == (h.n_buf, 2)
        got = buf[h.col_local.long()]
        assert torch.equal(got, table[col.long()]), (rank, got, table[col.long()])
        expect_halo = {0: 0, 1: 0, 2: 4, 3: 1}[rank]      # distinct remote rows referenced
        assert h.n_halo == expect_halo
        # row-sparse variant: same buffer as the dense exchange for every zero pattern, including
        # all rows zero and no row zero (zero-length messages on both sides)
        for pattern in (lambda r: r % 3 == 0, lambda r: True, lambda r: False, lambda r: r in (0, 19)):
            t2 = table.clone()
            for r in range(n):
                if pattern(r):
                    t2[r] = 0
            loc2 = t2[bounds[rank]:bounds[rank + 1]].contiguous()
            flags = (loc2 != 0).any(1)
            assert torch.equal(h.exchange_sparse(loc2, lambda idx: flags[idx]), h.exchange(loc2))
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_halo_exchange_degenerate_blocks(tmp_path):
    """Empty block, ranks that need nothing, duplicate references, zero-length transfers."""
    import torch.multiprocessing as mp
    mp.spawn(_halo_worker, args=(4, _free_port(), str(tmp_path)), nprocs=4, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(4)]


def _transpose_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygcn_amd.sharded import row_block, transpose_block_by_exchange, transpose_row_block
    from pygcn_amd.utils import rmat_graph
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 300
        rowptr, col, val = rmat_graph(n, 2500, seed=11, device="cpu")
        for bounds in ([0, 70, 70, 180, 300], [0, 0, 0, 300, 300], [0, 1, 2, 3, 300]):
            a = row_block(rowptr, col, val, bounds[rank], bounds[rank + 1])
            got = transpose_block_by_exchange(a, bounds, rank, world)
            ref = transpose_row_block(rowptr, col, val, n, bounds[rank], bounds[rank + 1])
            for g, r in zip(got, ref):
                assert g.dtype == r.dtype and torch.equal(g, r), (rank, bounds)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_transpose_by_exchange_degenerate_bounds(tmp_path):
    """Empty blocks, one-row blocks, a rank that owns everything: the exchanged transpose blocks
    equal the ones cut from the global matrix, entry order included."""
    import torch.multiprocessing as mp
    mp.spawn(_transpose_worker, args=(4, _free_port(), str(tmp_path)), nprocs=4, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(4)]


def _rowsets_worker(rank, world, port, n, n_edges, out_dir):
    """Structure of the sharded one-node path (pygcn_amd/sharded_fused.py) on CPU: the static halo
    of gradient rows and the [R2_r, R] block of each rank's rows of the transpose must reproduce
    rows R2_r of Aᵀ·G for a G that is non-zero on the union of the ranks' loss rows only."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scipy.sparse as sp
    import torch.distributed as dist
    from pygcn_amd.sharded import ShardedGraph
    from pygcn_amd.sharded_fused import ShardedRowSets
    from pygcn_amd.utils import rmat_graph
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rowptr, col, val = rmat_graph(n, n_edges, seed=9, device="cpu")
        sg = ShardedGraph.from_global_csr(rowptr, col, val, n, rank, world, graph_factory=_CpuGraph,
                                          spmm_fn=_cpu_spmm, bwd_fn=_cpu_bwd)
        gen = torch.Generator().manual_seed(50 + rank)
        k = 0 if (world > 2 and rank == 1) else max(1, sg.n_local // 7)      # one rank without loss rows
        rows_local = torch.randperm(sg.n_local, generator=gen)[:k]
        rs = ShardedRowSets(sg, rows_local)
        assert rs.n_u == k and rs.at_block.shape == (rs.n2, rs.n_u + rs.hx.n_halo)
        # G: random on the union of the loss rows, zero elsewhere (every rank builds the same G)
        parts = [None] * world
        dist.all_gather_object(parts, (rows_local + sg.r0))
        r_global = torch.sort(torch.cat(parts)).values
        C = 5
        G = torch.zeros(n, C, dtype=torch.float64)
        G[r_global] = torch.from_numpy(np.random.default_rng(77).standard_normal((r_global.numel(), C)))
        a = sp.csr_matrix((val.numpy().astype(np.float64), col.numpy(), rowptr.numpy()), shape=(n, n))
        want = torch.from_numpy((a.T @ G.numpy())[sg.r0:sg.r1])              # my rows of Aᵀ·G
        # what the backward pass does: my compact rows, the static exchange, the block product
        gp = G[sg.r0:sg.r1][rs.rows_u].float()
        halo, pending = rs.hx.exchange_begin(gp)
        rs.hx.exchange_end(pending)
        got = _cpu_spmm(rs.at_block, gp, B2=halo) if rs.n_u else (
            _cpu_spmm(rs.at_block, halo) if halo.shape[0] else torch.zeros(rs.n2, C))
        assert np.abs(got.numpy() - want[rs.rows2].numpy()).max() <= 1e-5 * max(1e-30, want.abs().max().item())
        outside = torch.ones(sg.n_local, dtype=torch.bool)
        outside[rs.rows2] = False
        assert float(want[outside].abs().max()) == 0.0 if outside.any() else True
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_static_gradient_halo_and_transpose_block(tmp_path, world):
    import torch.multiprocessing as mp
    mp.spawn(_rowsets_worker, args=(world, _free_port(), 900, 6000, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


def _selfcheck_worker(rank, world, port, out_dir, corrupt):
    """pygcn_amd/selfcheck.py over gloo: healthy exchanges pass; a halo that is overwritten after
    its wait (the symptom of a transfer that had not landed) and a gradient halo that loses its rows
    must both be caught on EVERY rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import gcn_oracle
    from pygcn_amd import GCN, selfcheck as sc
    from pygcn_amd.sharded import ShardedGCN, ShardedGraph
    from pygcn_amd.utils import rmat_graph
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, fin, nhid, ncls = 3000, 24, 32, 16
        rowptr, col, val = rmat_graph(n, 24000, seed=7, device="cpu")
        sg = ShardedGraph.from_global_csr(rowptr, col, val, n, rank, world, exchange="halo",
                                          graph_factory=_CpuGraph, spmm_fn=_cpu_spmm, bwd_fn=_cpu_bwd,
                                          overlap=True)
        x = torch.from_numpy(np.random.default_rng(1).standard_normal((n, fin)).astype(np.float32))
        labels = torch.from_numpy(np.random.default_rng(2).integers(0, ncls, n))
        train = torch.from_numpy(np.sort(np.random.default_rng(3).choice(n, n // 5, False)))
        x_loc, y_loc = x[sg.r0:sg.r1], labels[sg.r0:sg.r1]
        idx_loc = train[(train >= sg.r0) & (train < sg.r1)] - sg.r0
        torch.manual_seed(42)
        model = GCN(fin, nhid, ncls, dropout=0.0)
        smodel = ShardedGCN(model, sg)
        model.train()
        smodel.declare_loss_rows(idx_loc)
        state = {}
        if corrupt == "overlap" and rank == 1:
            begin, end = sg.halo.exchange_begin, sg.halo.exchange_end

            def bad_begin(local):
                halo, pending = begin(local)
                state["halo"] = halo
                return halo, pending

            def bad_end(pending):
                end(pending)
                if sg.overlap and state["halo"].shape[0]:          # pipelined form only: stale rows
                    state["halo"][::2] = 123.0
            sg.halo.exchange_begin, sg.halo.exchange_end = bad_begin, bad_end
        ops = [x_loc[:, :ncls].contiguous() * (k + 1) for k in range(3)]
        res = sc.overlap_selftest(sg, ops, tol=1e-5)
        if corrupt == "overlap":
            assert res["agrees"] is False and res["overlap_in_use"] is False and sg.overlap is False, res
            assert res["max_err"] > 1e-3 and "fell back" in res["note"]
        else:
            assert res["agrees"] is True and sg.overlap is True and res["max_err"] <= 1e-5, res
        # ---- gradient check: the sharded step vs the unsharded oracle step "on rank 0"
        if corrupt == "gradient" and rank == 1:
            sparse = sg.halo_t.exchange_sparse
            sg.halo_t.exchange_sparse = lambda local, row_nonzero, static_key=None: \
                torch.zeros_like(sparse(local, row_nonzero, static_key))       # the gradient rows are lost

        def sharded_step():
            model.zero_grad()
            smodel.nll_loss(smodel(x_loc, sg), y_loc, idx_loc).backward()
            smodel.allreduce_grads()

        def reference_step():
            a = gcn_oracle.CSR(rowptr.numpy(), col.numpy(), val.numpy(), (n, n))
            p = {k: v.detach().numpy() for k, v in model.state_dict().items()}
            _, _, grads, _ = gcn_oracle.gcn2_loss_backward(x.numpy(), a, p, labels.numpy(), train.numpy())
            return [torch.from_numpy(np.ascontiguousarray(grads[k])) for k, _ in model.named_parameters()]
        res = sc.sharded_grad_check(list(model.parameters()), sharded_step, reference_step, tol=5e-5)
        if corrupt == "gradient":
            assert res["ok"] is False and res["max_err"] > 1e-3, res
        else:
            assert res["ok"] is True and res["max_err"] <= 5e-5 and len(res["per_param"]) == 4, res
        # ---- the forward exchange A/B leaves every rank in the same (fastest) mode, all forms agree
        if corrupt is None:
            with torch.no_grad():
                h1 = torch.relu(torch.from_numpy(np.random.default_rng(5 + rank).standard_normal(
                    (sg.n_local, nhid)).astype(np.float32)))
                ab = sc.forward_exchange_ab(sg, h1, model.gc2.weight, model.gc2.bias,
                                            ["halo", "allgather", "rccl-allgather", "compress-hidden"],
                                            reps=1, log_softmax=False)
            assert set(ab["ms"]) == {"halo", "allgather", "rccl-allgather", "compress-hidden"}
            assert all(v <= 1e-5 for v in ab["max_err_vs_first_mode"].values()), ab
            chosen = [None] * world
            dist.all_gather_object(chosen, (ab["chosen"], sg.fwd_exchange, sg.compress_hidden))
            assert len(set(chosen)) == 1, chosen
            lr = sc.link_rate(torch.device("cpu"), 1 << 20)
            assert lr["gb_per_s"] > 0
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,corrupt", [(2, None), (3, None), (2, "overlap"), (3, "overlap"),
                                           (2, "gradient"), (3, "gradient")])
def test_multi_gpu_selfcheck_fires_on_a_corrupted_exchange(world, corrupt, tmp_path, oracle):
    """the first run between GPUs validates itself (pygcn_amd/selfcheck.py,
    called by bench.py at world > 1).  Healthy exchanges pass; a halo overwritten after its wait makes
    the overlap self-test fall back to the unpipelined exchange on every rank; lost gradient rows
    make the sharded gradient check fail on every rank."""
    import torch.multiprocessing as mp
    mp.spawn(_selfcheck_worker, args=(world, _free_port(), str(tmp_path), corrupt), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]
