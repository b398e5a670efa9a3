"""The 2-layer training step on a SHARDED graph as one autograd node per rank, when the loss reads a
known set of rows — the multi-GPU form of pygcn_amd/fused.py.

Upstream's epoch feeds `output[idx_train]` to the loss (reference pygcn/train.py:153).  On a row-block
shard every rank owns the labelled vertices R_r inside its block; which rows of every later gradient
can be non-zero follows from the graph and R = ∪ R_r alone, so it is worked out ONCE per
(sharded graph, rows):

    grad_pre2 = d loss / d (Â·h1·W2 + b2)         non-zero on rows R
    grad_sup2 = (Âᵀ)_r · grad_pre2                 this rank's rows R2_r = its vertices with a neighbour in R;
                                                   needs the rows of grad_pre2 in R that other ranks own
                                                   and this block references: a STATIC halo of gradient rows
    everything after that is local and runs on the compact rows R2_r, as in fused.py

Setup (collective): the ranks all-gather their R_r, each cuts the [R2_r, R] block of its rows of Âᵀ,
and a HaloExchange over that block's columns fixes who sends which gradient rows to whom.  Per epoch
the backward pass then has ONE grouped point-to-point round of known sizes (no count exchange, no
host synchronisation), one product on the small block, three GEMMs on compact rows, and the
256 KiB gradient all-reduce of ShardedGCN.allreduce_grads.  The forward pass is the ordinary
sharded one (first layer without exchange, evaluated as (Â_r·[X_r ; X_halo])·W1; second layer with
the pipelined dense halo exchange and the log_softmax in the completing launch).
"""
import torch
import torch.distributed as dist

from . import spmm as _spmm
from .sharded import HaloExchange
from .spmm import _dense_forward, _weight_grad, gemm_xw256, log_softmax_fusable


class ShardedRowSets:
    """Static structure of the backward pass for loss rows `rows_local` (local row ids of this
    rank's block).  Collective: every rank of the group constructs it together."""

    def __init__(self, sg, rows_local):
        dev, n_loc = rows_local.device, sg.n_local
        rows = rows_local.to(torch.int64)
        if rows.numel() and (int(rows.min()) < 0 or int(rows.max()) >= n_loc):
            raise RuntimeError("rows: index out of range of this rank's block")
        self.rows_user = rows
        self.rows_u, self.inverse = torch.unique(rows, return_inverse=True)          # sorted
        self.n_u = int(self.rows_u.numel())
        self.has_duplicates = self.n_u != rows.numel()
        self.sorted_unique = bool(not self.has_duplicates and (self.n_u == 0 or bool((rows == self.rows_u).all())))
        # R as global ids on every rank (rank blocks ascend, each part sorted: the whole is sorted)
        W = sg.world
        counts = torch.empty(W, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(counts, torch.tensor([self.n_u], dtype=torch.int64, device=dev),
                                    group=sg.group)
        counts = counts.tolist()
        cap = max(max(counts), 1)
        mine = torch.full((cap,), -1, dtype=torch.int64, device=dev)
        mine[:self.n_u] = self.rows_u + sg.r0
        everyone = torch.empty(W * cap, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(everyone, mine, group=sg.group)
        r_global = torch.cat([everyone[r * cap:r * cap + counts[r]] for r in range(W)])
        self.n_global_rows = int(r_global.numel())
        # this rank's rows of Âᵀ with GLOBAL source ids, restricted to sources in R
        rowptr, val = sg._raw[True]
        h = sg.halo_t
        col = sg.At.col.to(torch.int64)
        col_g = col + sg.r0
        if h.n_halo:
            col_g = torch.where(h.is_own, col_g, h.halo_global[(col - n_loc).clamp_(min=0)])
        pos = torch.searchsorted(r_global, col_g).clamp_(max=max(self.n_global_rows - 1, 0))
        keep = (r_global[pos] == col_g) if self.n_global_rows else torch.zeros_like(col_g, dtype=torch.bool)
        deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
        erow = torch.repeat_interleave(torch.arange(n_loc, device=dev, dtype=torch.int64), deg)[keep]
        ecol, eval_ = col_g[keep], val[keep]
        del col, col_g, pos, keep
        self.rows2 = torch.unique(erow)                                              # sorted local row ids
        self.n2 = int(self.rows2.numel())
        self.rows2_i32 = self.rows2.to(torch.int32)
        self.rows2_padded = _spmm.padded_row_list(self.rows2)
        # who sends which rows of grad_pre2 to whom: a halo exchange over the block's columns
        self.hx = HaloExchange(ecol, sg.bounds, sg.rank, W, sg.group)                # (collective)
        own = self.hx.is_own
        cl = self.hx.col_local.to(torch.int64)
        pos_own = torch.searchsorted(self.rows_u, cl.clamp(max=max(n_loc - 1, 0)))
        col_c = torch.where(own, pos_own, cl - n_loc + self.n_u)
        if bool(own.any()):
            assert bool((self.rows_u[pos_own[own]] == cl[own]).all())                # own sources are own loss rows
        # the rows peers asked of me, as positions in MY compact grad_pre2 (rows_u order)
        send_pos = torch.searchsorted(self.rows_u, self.hx.send_idx)
        if send_pos.numel():
            assert bool((self.rows_u[send_pos] == self.hx.send_idx).all())
        self.hx.send_idx = send_pos
        # CSR over the compact rows R2_r; the entries keep the order of CSR(Âᵀ) (increasing source row)
        dst = torch.searchsorted(self.rows2, erow)
        rp = torch.zeros(self.n2 + 1, dtype=torch.int64, device=dev)
        if erow.numel():
            torch.cumsum(torch.bincount(dst, minlength=self.n2), 0, out=rp[1:])
        self.at_block = sg._graph_factory(rp.to(torch.int32 if erow.numel() < 2 ** 31 - 1 else torch.int64),
                                          col_c.to(torch.int32), eval_.contiguous(),
                                          (self.n2, self.n_u + self.hx.n_halo))


def fusable(sg, model, x_local):
    """Can ShardedGCN take the one-node path for this model / input?"""
    gc1, gc2 = getattr(model, "gc1", None), getattr(model, "gc2", None)
    return (gc1 is not None and gc2 is not None and not hasattr(model, "gc3")
            and sg.exchange_mode == "halo" and sg._hinted_product and x_local.is_cuda
            and x_local.dim() == 2 and not x_local.requires_grad
            and _spmm.layer_gemm_reassociable(x_local, gc1.weight, gc1.bias)
            and log_softmax_fusable(gc2.out_features, gc2.weight.dtype))


class ShardedGCN2RowsFunction(torch.autograd.Function):
    """log_softmax(Â·dropout(relu(Â·X·W1 + b1))·W2 + b2)[rows] on this rank's block
    (models.py:47-71 upstream form); parameter gradients are this rank's partial sums."""

    @staticmethod
    def forward(ctx, sg, rs, x_local, x_halo, w1, b1, w2, b2, dropout_p, seed):
        ctx.sg, ctx.rs = sg, rs
        ctx.scale = _spmm.dropout_scale(dropout_p)
        f32 = x_local.dtype == torch.float32
        ev = sg._tic(x_local)
        z = sg._spmm(sg.A, x_local, tag="fwd_local", B2=x_halo)
        sg._toc(ev, "fwd")
        ctx.z_bound = h_bound = None
        if f32:
            ctx.z_bound = sg.A.inf_norm() * sg.constant_absmax(x_local) * 1.0001
            h_bound = torch.zeros(1, dtype=torch.float32, device=x_local.device)
        kw = {"dropout_p": dropout_p, "seed": seed, "row_base": sg.r0} if dropout_p > 0.0 else {}
        # (`h1 > 0` as one bit per element for the masked grad_input GEMM: pygcn_amd/fused.py)
        ctx.keep_bits = None
        if f32 and _spmm.gemm_keep_bits_usable(z, None, dropout_p) and any(ctx.needs_input_grad):
            ctx.keep_bits = torch.empty((z.shape[0], 8), dtype=torch.int32, device=z.device)
            kw = dict(kw, keep_bits_out=ctx.keep_bits)
        h1 = _spmm.layer_gemm(z, w1, ctx.z_bound, h_bound, bias=b1, relu=True, **kw)
        if h1 is None:
            raise RuntimeError("sharded one-node path: the layer GEMM declined the operands")
        ctx.h_bound = h_bound
        if sg.compress_hidden and h1.shape[1] % 32 == 0:
            # h1 is >= 75 % zeros (ReLU + dropout): its halo rows travel as bitmask + values and meet
            # W2 on arrival (ShardedGraph.product_hidden) instead of dense rows of h1·W2 travelling
            logp = sg.product_hidden(h1, w2, bias=b2, log_softmax=True, h_bound=h_bound)
        else:
            logp = sg.product(_dense_forward(h1, w2, h_bound), bias=b2, log_softmax=True)
        out_rows = logp.index_select(0, rs.rows_user)
        ctx.save_for_backward(z, w1, w2, h1, out_rows)
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.bias_dtypes = (b1.dtype if b1 is not None else None, b2.dtype if b2 is not None else None)
        return out_rows

    @staticmethod
    def backward(ctx, grad_rows):
        z, w1, w2, h1, out_rows = ctx.saved_tensors
        sg, rs = ctx.sg, ctx.rs
        need_w1, need_b1, need_w2, need_b2 = ctx.needs_input_grad[4:8]
        dev, dt = z.device, h1.dtype
        one_pass = _spmm.backward_with_colsum(grad_rows.contiguous(), out_rows, log_softmax=True) \
            if (grad_rows.dtype == out_rows.dtype and not rs.has_duplicates and rs.n_u) else None
        if one_pass is not None:
            gp, colsum, _ = one_pass
            grad_b2 = colsum.to(ctx.bias_dtypes[1]) if (ctx.has_bias[1] and need_b2) else None
        else:
            g = grad_rows.float()
            gp = g - out_rows.float().exp() * g.sum(1, keepdim=True)
            grad_b2 = gp.sum(0).to(ctx.bias_dtypes[1]) if (ctx.has_bias[1] and need_b2) else None
        gp = gp.to(dt)
        if rs.has_duplicates:
            gp = torch.zeros((rs.n_u, gp.shape[1]), dtype=dt, device=dev).index_add_(0, rs.inverse, gp)
        elif not rs.sorted_unique:
            gp = torch.empty_like(gp).index_copy_(0, rs.inverse, gp)
        gp = gp.contiguous()
        # ---- layer 2: the static halo of gradient rows, then the block product (compact in / out)
        ev = sg._tic(gp)
        halo, pending = rs.hx.exchange_begin(gp)
        rs.hx.exchange_end(pending)
        sg.last_recv_bytes["bwd"] = rs.hx.last_recv_bytes
        f32 = dt == torch.float32
        # max|grad_sup2| — the bound of the scaled GEMMs below — comes out of the product itself
        # (gcn_epilogue.c_absmax; test stand-ins of the product take a reduction pass instead)
        gs_max = torch.zeros(1, dtype=torch.float32, device=dev) if (f32 and sg._hinted_product) else None
        kw = {"c_absmax": gs_max} if gs_max is not None else {}
        if rs.n_u:
            grad_sup2 = sg._spmm(rs.at_block, gp, tag="bwd_local", B2=halo, **kw)
        elif halo.shape[0]:                       # no labelled vertex of my own: halo rows only
            grad_sup2 = sg._spmm(rs.at_block, halo, tag="bwd_local", **kw)
        else:
            grad_sup2 = gp.new_zeros((rs.n2, gp.shape[1]))
        sg._toc(ev, "bwd")
        gs_bound = None
        if f32:
            gs_bound = gs_max * 1.0001 if gs_max is not None else (
                torch.linalg.vector_norm(grad_sup2, ord=float("inf")).float().reshape(1) * 1.0001
                if rs.n2 else grad_sup2.new_zeros(1).float())
        fast = f32 and _spmm.gemm_handwritten() and grad_sup2.shape[1] == 256 and h1.shape[1] == 256 \
            and rs.n2 > 0
        grad_w1 = grad_w2 = grad_b1 = None
        h1c = None
        if need_w2:
            grad_w2 = _spmm.weight_grad_rows(h1, grad_sup2, rs.rows2_padded, None, ctx.h_bound, gs_bound,
                                             n_list=rs.n2) if (fast or (not f32 and rs.n2 > 0)) else None
            if grad_w2 is None:
                h1c = h1.index_select(0, rs.rows2)
                grad_w2 = _weight_grad(h1c, grad_sup2)
        gh_max = torch.zeros(1, dtype=torch.float32, device=dev) if f32 else None
        w2t = w2.t().contiguous()
        gpre1 = gemm_xw256(grad_sup2, w2t, gs_bound, gh_max, mask_src=h1, mask_rows=rs.rows2_i32,
                           mask_bits=getattr(ctx, "keep_bits", None), mask_scale=ctx.scale) if fast else None
        if gpre1 is None and dt == torch.bfloat16 and rs.n2:
            gpre1 = _spmm.gemm_bf16(grad_sup2, w2t, mask_src=h1, mask_rows=rs.rows2_i32, mask_scale=ctx.scale)
        if gpre1 is None:
            h1c = h1.index_select(0, rs.rows2) if h1c is None else h1c
            gh1 = _dense_forward(grad_sup2, w2t, gs_bound, gh_max) if rs.n2 else grad_sup2.new_zeros((0, w2.shape[0]))
            gpre1 = torch.where(h1c > 0, gh1 * ctx.scale if ctx.scale != 1.0 else gh1,
                                torch.zeros((), dtype=dt, device=dev))
            if gh_max is not None:
                gh_max = gh_max * ctx.scale
        want_b1 = ctx.has_bias[0] and need_b1
        if want_b1 and need_w1 and fast and rs.n2 > 0:          # (grad_W1 and grad_b1 from one pass over grad_pre1)
            both = _spmm.weight_grad_rows(z, gpre1, rs.rows2_padded, None, ctx.z_bound, gh_max if f32 else None,
                                          n_list=rs.n2, colsum_g=True)
            if both is not None:
                grad_w1, grad_b1 = both[0], both[1].to(ctx.bias_dtypes[0])
        if want_b1 and grad_b1 is None:
            sums = _spmm.backward_with_colsum(gpre1) if (gpre1.is_contiguous() and rs.n2) else None
            grad_b1 = (sums[1] if sums is not None else gpre1.float().sum(0)).to(ctx.bias_dtypes[0])
        if need_w1 and grad_w1 is None:
            grad_w1 = _spmm.weight_grad_rows(z, gpre1, rs.rows2_padded, None, ctx.z_bound,
                                             gh_max if f32 else None, n_list=rs.n2) \
                if (fast or (not f32 and rs.n2 > 0)) else None
            if grad_w1 is None:
                grad_w1 = _weight_grad(z.index_select(0, rs.rows2), gpre1)
        return None, None, None, None, grad_w1, grad_b1, grad_w2, grad_b2, None, None
