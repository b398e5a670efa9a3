"""The 2-layer GCN training step as ONE autograd node when the loss reads a known set of rows.

Upstream's epoch (reference pygcn/train.py:140-157, kept there as comments) is

    output = model(features, adj)                                   models.py:47-71 (upstream form)
    loss_train = F.nll_loss(output[idx_train], labels[idx_train])   train.py:153
    loss_train.backward()                                           train.py:157

so the gradient that enters the model is non-zero on the rows `idx_train` only — and which rows of
every later gradient CAN be non-zero follows from the graph alone:

    grad_pre2 = d loss / d (Â·h1·W2 + b2)      rows  R  = idx_train
    grad_sup2 = Âᵀ · grad_pre2                 rows  R2 = vertices that have a neighbour in R
    grad_W2   = h1ᵀ · grad_sup2,  grad_h1 = grad_sup2 · W2ᵀ,  grad_pre1 = mask(grad_h1)   rows R2
    grad_W1   = (Â·X)ᵀ · grad_pre1 = (Â·X)[R2]ᵀ · grad_pre1[R2]
                (Â·X is the first product of the forward pass itself when the layer is evaluated
                 as (Â·X)·W1 — 256 -> 256 fp32 — so layer 1 needs no sparse product in backward)

`model(features, adj, rows=idx_train)` returns `output[idx_train]` from a node that keeps all of
this inside: R2 and the block of Âᵀ with rows R2 and columns R are cut once per (graph, rows) from
the CSR structure, every intermediate gradient lives in COMPACT form ([|R|, C] and [|R2|, H]
tensors), the layer-2 transpose product is the HIP kernel on that block (layer 1 needs none, see
below; other widths use the row-restricted launch, `gcn_epilogue.c_row_select`), and nothing of
size [N, ·] is allocated, zero-filled, scattered into, or swept to find its non-zero rows.  Compared with letting the dense
`grad_output [N, C]` travel through autograd this removes per epoch at config C4: the 10 GB zero
fill of index_put's backward and its scatter, the NLL kernels over [N, C], two full-height
backward sweeps and the scatter of grad_h1 — ≈ 11 ms of an 84 ms epoch — and every host
synchronisation of the backward pass (the row sets are static, their sizes are known on the host).
Only structurally-zero rows are skipped: the result is the same sum of the same products.

The forward pass is the ordinary one (both products over all rows, fused bias / ReLU / dropout /
log_softmax epilogues): the full log-probability matrix exists and can be kept (`keep_full=True`)
for validation on other rows, as upstream's --fastmode does.
"""
import weakref

import torch

from .graph import CSRGraph
from . import spmm as _spmm
from .spmm import (_dense_forward, _weight_grad, gemm_xw256, log_softmax_fusable, pack_row_flags,
                   spmm_csr)


def _maybe_poisoned(shape, dtype, device):      # (reads the test switch at call time)
    return _spmm._maybe_poisoned(shape, dtype, device)


class RowSets:
    """What the backward pass needs to know about the loss rows R, computed ONCE per
    (graph, rows tensor): R sorted and unique, R2 = the columns that occur in rows R of Â (= the
    rows of Âᵀ·grad_pre2 that can be non-zero) with its bitmap / count, and `at_block` = the
    [|R2|, |R|] block of Âᵀ the layer-2 backward product runs on."""

    def __init__(self, graph, rows):
        dev, n = graph.device, graph.shape[0]
        rows = rows.to(device=dev, dtype=torch.int64)
        if rows.numel() and (int(rows.min()) < 0 or int(rows.max()) >= n):
            raise RuntimeError("rows: index out of range")
        self.rows_user = rows
        self.rows_u, self.inverse = torch.unique(rows, return_inverse=True)      # sorted
        self.has_duplicates = self.rows_u.numel() != rows.numel()
        self.n_u = int(self.rows_u.numel())
        rp = graph.rowptr.to(torch.int64)
        starts, lens = rp[self.rows_u], rp[self.rows_u + 1] - rp[self.rows_u]
        total = int(lens.sum())
        idx = torch.repeat_interleave(starts - torch.cumsum(lens, 0) + lens, lens) + \
            torch.arange(total, device=dev)
        cols = graph.col[idx].to(torch.int64)
        self.rows2 = torch.unique(cols)                                          # sorted
        self.n2 = int(self.rows2.numel())
        # The block of Âᵀ the backward product needs — rows R2, columns R — as its own small CSR
        # (compact row / column numbering): entry (r, c) of Â with r in R becomes entry
        # (pos of c in R2, pos of r in R) of the block.  Sorted by (row, source row): within a row
        # the entries come in the order of the full CSR(Âᵀ).  Built once per (graph, rows).
        src = torch.repeat_interleave(torch.arange(self.n_u, device=dev, dtype=torch.int64), lens)
        dst = torch.searchsorted(self.rows2, cols)
        order = torch.argsort(dst * max(self.n_u, 1) + src)
        rp = torch.zeros(self.n2 + 1, dtype=torch.int64, device=dev)
        if total:
            torch.cumsum(torch.bincount(dst, minlength=self.n2), 0, out=rp[1:])
        self.at_block = CSRGraph(rp.to(torch.int32 if total < 2 ** 31 - 1 else torch.int64),
                                 src[order].to(torch.int32), graph.val[idx][order].contiguous(),
                                 (self.n2, self.n_u))
        self.sorted_unique = bool(self.n_u == rows.numel() and (self.n_u == 0 or bool(
            (rows == self.rows_u).all())))
        self.rows2_i32 = self.rows2.to(torch.int32)
        self.rows2_padded = _spmm.padded_row_list(self.rows2)     # (for the weight-gradient kernel)
        mask2 = torch.zeros(graph.shape[1], dtype=torch.bool, device=dev)
        mask2[self.rows2] = True
        self.hint2 = pack_row_flags(mask2)


_ROWSETS = weakref.WeakKeyDictionary()    # graph -> {index-tensor identity: (RowSets, rows)}


def rows_key(rows):
    """Identity of an index tensor for the row-set caches: everything that determines WHICH
    elements it reads (two strided views of one storage share pointer, length and version counter
    — idx[:100] vs idx[0:200:2] — so stride and storage offset are part of it), plus the version
    counter for in-place edits."""
    return (rows.data_ptr(), rows.numel(), rows._version, rows.dtype, tuple(rows.stride()),
            rows.storage_offset(), str(rows.device))


def row_sets(graph, rows):
    per_graph = _ROWSETS.setdefault(graph, {})
    key = rows_key(rows)
    rs = per_graph.get(key)
    if rs is None:
        if len(per_graph) >= 4:
            per_graph.clear()
        rs = per_graph[key] = (RowSets(graph, rows), rows)     # (keeps `rows` alive: ptr stays unique)
    return rs[0]


def fusable(model_dtype, nclass, graph, x):
    """Shapes / layouts the one-node path covers; anything else takes the layer-by-layer path."""
    return (isinstance(graph, CSRGraph) and x.dim() == 2 and x.is_cuda and graph.shape[0] == graph.shape[1]
            and log_softmax_fusable(nclass, model_dtype))


def _rows_honoured(n, width, dtype, count):
    """The device-side rule of the product (use_row_flags in gcn_spmm.hip, mirrored by
    spmm._hint_will_be_used) for a freshly allocated [n, width] operand: rows whose hint bit is
    clear are skipped below 3/4 non-zero rows in the wide kernel, below 1/8 in the narrow one."""
    v = 16 // torch.empty((), dtype=dtype).element_size()
    wide = width % v == 0 and width // v > 32
    from . import tuning
    return tuning.below(count, n, tuning.HINT_WIDE_MAX_SHARE if wide else tuning.HINT_NARROW_MAX_SHARE)


def _operand_buffer(n, width, dtype, device, rows, values, count):
    """[n, width] tensor holding `values` at `rows`; the other rows are left UNWRITTEN when the
    product that reads it is certain to honour the row hint (it then never touches them), and are
    zero otherwise."""
    buf = _maybe_poisoned((n, width), dtype, device) if _rows_honoured(n, width, dtype, count) \
        else torch.zeros((n, width), dtype=dtype, device=device)
    if rows is not None:           # (None: the caller writes the rows itself)
        buf.index_copy_(0, rows, values)
    return buf


_CACHE_INPUT_PRODUCT = False


def set_input_product_cache(enabled):
    """OPT-IN (default off).  With layer 1 evaluated as (Â·X)·W1, its sparse product z = Â·X is a
    product of two CONSTANTS of a training run (the adjacency and the feature matrix): switched on,
    z is computed once per (graph, feature tensor, their version counters) and reused by every later
    step — an epoch then contains ONE forward sparse product instead of two (−16 ms of 50 at C4).
    The result is bitwise the same.  Off by default because the benchmark's epoch is defined with
    both forward products inside it; bench.py reports the cached figure beside it."""
    global _CACHE_INPUT_PRODUCT
    _CACHE_INPUT_PRODUCT = bool(enabled)


def input_product(graph, x):
    """z = Â·X for the layer-1 input, from the per-graph cache when set_input_product_cache(True)
    and neither the graph's values nor X changed since it was computed."""
    if not _CACHE_INPUT_PRODUCT or x.requires_grad:
        return spmm_csr(graph, x)
    hit = getattr(graph, "_input_product", None)
    if (hit is not None and hit[0]() is x and hit[1] == x._version and hit[2] == graph.val._version):
        return hit[3]
    z = spmm_csr(graph, x)
    graph._input_product = (weakref.ref(x), x._version, graph.val._version, z)
    return z


def _gcn2_forward(ctx, x, w1, b1, w2, b2, graph, dropout_p, seed):
    """Forward pass shared by the one-node functions: (tensor saved in place of x, h1, logp).
    Fills ctx.scale / x_bound / z_bound / h_bound / reassoc / has_bias / bias_dtypes."""
    ctx.graph = graph
    ctx.scale = _spmm.dropout_scale(dropout_p)
    # bounds of max|operand| for the scaled fp16 GEMMs (set_gemm_scheme("h2") only; the default
    # three-part bf16 GEMMs need none), without a pass over the data:
    # X is constant (cached), and |Â·B| <= ‖Â‖∞·max|B|
    bounded = x.dtype == torch.float32 and _spmm.gemm_needs_bounds()
    ctx.x_bound = _spmm.absmax_cached(x) if bounded else None
    # Layer 1 REASSOCIATED when a GEMM kernel can carry the layer's epilogue (256 -> 256 fp32;
    # bf16 128 -> 128 / 256):
    #     h1 = dropout(relu((Â·X)·W1 + b1))        instead of   dropout(relu(Â·(X·W1) + b1))
    # — the same two kernels and the same bytes in the forward pass (an SpMM at the input's
    # width, a GEMM), but the product z = Â·X of THIS forward pass is then all the backward
    # pass needs for grad_W1 = zᵀ·grad_pre1: no second sparse product for layer 1 (12.7 ms at
    # C4, 35.5 ms at C5).
    ctx.reassoc = bool(_spmm.layer_gemm_reassociable(x, w1, b1))
    ctx.keep_bits = None
    h1 = h_bound = z = None
    ctx.z_bound = None
    if ctx.reassoc:
        z = input_product(graph, x)
        if bounded:
            ctx.z_bound = graph.inf_norm() * ctx.x_bound * 1.0001
            h_bound = torch.zeros(1, dtype=torch.float32, device=x.device)   # max|h1|, exact
        # `h1 > 0` — all the backward of ReLU / dropout asks of h1 — as one bit per element, written by the
        # same launch where it can: the masked grad_input GEMM then reads 32 bytes per row instead of 1 KiB
        if _spmm.gemm_keep_bits_usable(z, None, dropout_p) and any(ctx.needs_input_grad):
            ctx.keep_bits = torch.empty((z.shape[0], 8), dtype=torch.int32, device=z.device)
        h1 = _spmm.layer_gemm(z, w1, ctx.z_bound, h_bound, bias=b1, relu=True, dropout_p=dropout_p,
                              seed=seed, keep_bits_out=ctx.keep_bits)
        if h1 is None:                     # (alignment the kernel cannot take)
            ctx.reassoc, z, h_bound, ctx.keep_bits = False, None, None, None
    if h1 is None:
        s_max = torch.zeros(1, dtype=torch.float32, device=x.device) if bounded else None
        sup1 = _dense_forward(x, w1, ctx.x_bound, s_max)
        h1 = spmm_csr(graph, sup1, bias=b1, relu=True, dropout_p=dropout_p, seed=seed)
        del sup1
        if bounded:   # |relu/dropout(Â·S + b)| <= (‖Â‖∞·max|S| + max|b|) / (1 - p)
            h_bound = graph.inf_norm() * s_max
            if b1 is not None:
                h_bound = h_bound + b1.detach().abs().max().float()
            h_bound = h_bound * (1.0001 * ctx.scale)
    ctx.h_bound = h_bound
    logp = spmm_csr(graph, _dense_forward(h1, w2, h_bound), bias=b2, log_softmax=True)
    ctx.has_bias = (b1 is not None, b2 is not None)
    ctx.bias_dtypes = (b1.dtype if b1 is not None else None, b2.dtype if b2 is not None else None)
    return (z if ctx.reassoc else x), h1, logp


def _gcn2_backward_rows(ctx, x, w1, w2, h1, out_rows, rs, grad_rows, needs):
    """Backward pass for a gradient that is non-zero on the loss rows only (module docstring):
    `grad_rows` [|rows|, C] in the user's row order, `out_rows` = logp at those rows.
    Returns (grad_x, grad_w1, grad_b1, grad_w2, grad_b2)."""
    graph = ctx.graph
    need_x, need_w1, need_b1, need_w2, need_b2 = needs
    n, dev, dt = graph.shape[0], x.device, h1.dtype
    graph_t = graph.t()
    # ---- loss rows: log_softmax backward on the compact [|R|, C] tensors — one HIP pass
    # (gcn_log_softmax_backward_colsum: grad_pre and the bias gradient's column sums together)
    one_pass = _spmm.backward_with_colsum(grad_rows.contiguous(), out_rows, log_softmax=True) \
        if (grad_rows.dtype == out_rows.dtype and not rs.has_duplicates) else None
    if one_pass is not None:
        gp, colsum, _ = one_pass
        grad_b2 = colsum.to(ctx.bias_dtypes[1]) if (ctx.has_bias[1] and need_b2) else None
    else:                                              # (class counts the kernel does not take)
        g = grad_rows.float()
        gp = g - out_rows.float().exp() * g.sum(1, keepdim=True)
        grad_b2 = gp.sum(0).to(ctx.bias_dtypes[1]) if (ctx.has_bias[1] and need_b2) else None
    gp = gp.to(dt)
    if rs.has_duplicates:                              # the same vertex listed twice: add up
        gp = torch.zeros((rs.n_u, gp.shape[1]), dtype=dt, device=dev).index_add_(0, rs.inverse, gp)
    elif not rs.sorted_unique:                         # rows of R in sorted order (the block's columns)
        gp = torch.empty_like(gp).index_copy_(0, rs.inverse, gp)
    grad_w1 = grad_w2 = grad_b1 = grad_x = None
    if not (need_x or need_w1 or need_b1 or need_w2):
        return grad_x, grad_w1, grad_b1, grad_w2, grad_b2
    # ---- layer 2: Âᵀ · grad_pre2 — only rows R of grad_pre2 are non-zero and only rows R2 of
    # the result can be: the product runs on that block of Âᵀ (RowSets.at_block), compact
    # operand [|R|, C] in, compact result [|R2|, C] out; nothing of size [N, C] exists
    f32 = dt == torch.float32
    # bound of max|grad_sup2| for the scaled GEMMs: its EXACT maximum, reported by the product
    # itself (gcn_epilogue.c_absmax: one conditional atomic per wave) — ‖Âᵀ‖∞·max|grad_pre2| would
    # overshoot by the hub column sums, a separate reduction pass costs 0.5 ms at C4
    gs_max = torch.zeros(1, dtype=torch.float32, device=dev) if f32 else None
    grad_sup2 = spmm_csr(rs.at_block, gp.contiguous(), tag="bwd_l2", c_absmax=gs_max)
    gs_bound = gs_max * 1.0001 if f32 else None
    # h1 is read at the rows R2 in place (row lists), no compacting copy; the ReLU / dropout
    # mask (h1 > 0 encodes ReLU and keep) is applied in the GEMM's store
    fast = f32 and _spmm.gemm_handwritten() and grad_sup2.shape[1] == 256 and h1.shape[1] == 256
    h1c = None
    if need_w2:
        # (fp32 256 x 256 with bounds, or bf16 128 x 128: rows of h1 read in place through the list)
        grad_w2 = _spmm.weight_grad_rows(h1, grad_sup2, rs.rows2_padded, None,
                                         ctx.h_bound, gs_bound, n_list=rs.n2) if (fast or not f32) else None
        if grad_w2 is None:
            h1c = h1.index_select(0, rs.rows2)
            grad_w2 = _weight_grad(h1c, grad_sup2)
    gh_max = torch.zeros(1, dtype=torch.float32, device=dev) if f32 else None
    w2t = w2.t().contiguous()
    gpre1 = gemm_xw256(grad_sup2, w2t, gs_bound, gh_max, mask_src=h1, mask_rows=rs.rows2_i32,
                       mask_bits=getattr(ctx, "keep_bits", None),
                       mask_scale=ctx.scale) if fast else None
    if gpre1 is None and dt == torch.bfloat16:       # (C5: the bf16 GEMM carries the mask too)
        gpre1 = _spmm.gemm_bf16(grad_sup2, w2t, mask_src=h1, mask_rows=rs.rows2_i32, mask_scale=ctx.scale)
    if gpre1 is None:
        h1c = h1.index_select(0, rs.rows2) if h1c is None else h1c
        gh1 = _dense_forward(grad_sup2, w2t, gs_bound, gh_max)
        gpre1 = torch.where(h1c > 0, gh1 * ctx.scale if ctx.scale != 1.0 else gh1,
                            torch.zeros((), dtype=dt, device=dev))
        if gh_max is not None:
            gh_max = gh_max * ctx.scale
        del gh1
    del h1c, grad_sup2
    gpre_bound = gh_max if f32 else None
    want_b1 = ctx.has_bias[0] and need_b1
    if want_b1 and ctx.reassoc and need_w1 and f32:
        # grad_W1 and grad_b1 from ONE pass over grad_pre1: the weight-gradient kernel sums the rows it loads
        both = _spmm.weight_grad_rows(x, gpre1, rs.rows2_padded, None, ctx.z_bound, gpre_bound, n_list=rs.n2,
                                      colsum_g=True)
        if both is not None:
            grad_w1, grad_b1 = both[0], both[1].to(ctx.bias_dtypes[0])
    if want_b1 and grad_b1 is None:
        sums = _spmm.backward_with_colsum(gpre1) if gpre1.is_contiguous() else None   # (one HIP pass)
        grad_b1 = (sums[1] if sums is not None else gpre1.float().sum(0)).to(ctx.bias_dtypes[0])
    # ---- layer 1
    if ctx.reassoc:
        z = x                                           # this step's Â·X, saved by forward
        if need_w1 and grad_w1 is None:
            grad_w1 = _spmm.weight_grad_rows(z, gpre1, rs.rows2_padded, None, ctx.z_bound,
                                             gpre_bound, n_list=rs.n2)
            if grad_w1 is None:
                grad_w1 = _weight_grad(z.index_select(0, rs.rows2), gpre1)
        if need_x:                                      # grad_X = Âᵀ·(grad_pre1·W1ᵀ)
            gz = _dense_forward(gpre1, w1.t().contiguous(), gpre_bound)
            grad_z = _operand_buffer(n, gz.shape[1], dt, dev, rs.rows2, gz, rs.n2)
            grad_x = spmm_csr(graph_t, grad_z, tag="bwd_l1", b_hint=rs.hint2)
    elif not need_x and x.shape[1] <= _spmm.REASSOC_MAX_WIDTH_RATIO * gpre1.shape[1]:
        if need_w1:
            # grad_W1 = (Â·X)[R2]ᵀ · grad_pre1[R2]: a forward product restricted to rows R2
            z = spmm_csr(graph, x, tag="bwd_l1", c_select=rs.hint2[0],
                         out=_maybe_poisoned((n, x.shape[1]), x.dtype, dev))
            if f32 and _spmm.gemm_handwritten() and x.shape[1] == 256 and gpre1.shape[1] == 256:
                z_bound = graph.inf_norm() * ctx.x_bound * 1.0001 if ctx.x_bound is not None else None
                grad_w1 = _spmm.weight_grad_rows(z, gpre1, rs.rows2_padded, None, z_bound,
                                                 gpre_bound, n_list=rs.n2)
            if grad_w1 is None:
                grad_w1 = _weight_grad(z.index_select(0, rs.rows2), gpre1)
    elif need_x or need_w1:
        grad_pre1 = _operand_buffer(n, gpre1.shape[1], dt, dev, rs.rows2, gpre1, rs.n2)
        grad_sup1 = spmm_csr(graph_t, grad_pre1, tag="bwd_l1", b_hint=rs.hint2)
        if need_w1:
            grad_w1 = _weight_grad(x, grad_sup1)
        if need_x:
            grad_x = _dense_forward(grad_sup1, w1.t().contiguous())
    return grad_x, grad_w1, grad_b1, grad_w2, grad_b2


def _gcn2_backward_dense(ctx, x, w1, w2, h1, logp, grad, needs):
    """Backward pass for a gradient that is non-zero on EVERY row (a loss over all vertices —
    the fork's live case reduces over all nodes, reference pygcn/train.py:151-155): nothing can be
    skipped, so the pass is the minimum number of full-height sweeps, each stage handing the next
    what it needs — no host synchronisation, no search for zero rows:

        grad_pre2 (+ grad_b2)   log_softmax backward sweep; for the mean-NLL gradient (NLLGrad,
                                pygcn_amd/functional.py) the loss gradient is never materialised:
                                coef·(onehot(target) − exp(logp)) straight from logp and the labels
        grad_sup2 = Âᵀ·grad_pre2        the SpMM kernel on CSR(Âᵀ), full height
        grad_W2   = h1ᵀ·grad_sup2       gather-fused MFMA kernel (all rows, in order)
        grad_pre1 = mask(grad_sup2·W2ᵀ) MFMA GEMM, ReLU / dropout mask (h1 > 0) in its store
        grad_b1                         one column-sum sweep
        grad_W1   = zᵀ·grad_pre1        with z = Â·X of the forward pass (layer 1 reassociated);
                                        other shapes: Âᵀ·grad_pre1 first
    Returns (grad_x, grad_w1, grad_b1, grad_w2, grad_b2)."""
    from .functional import NLLGrad
    graph = ctx.graph
    need_x, need_w1, need_b1, need_w2, need_b2 = needs
    dev, dt = x.device, h1.dtype
    f32 = dt == torch.float32
    graph_t = graph.t()
    one_pass = None
    if isinstance(grad, NLLGrad):
        one_pass = _spmm.nll_log_softmax_backward(logp, grad.target, grad.coef)
        if one_pass is None:
            grad = grad.dense()
    if one_pass is None:
        grad = grad.to(logp.dtype).contiguous()
        one_pass = _spmm.backward_with_colsum(grad, logp, log_softmax=True)
    if one_pass is not None:
        gp, colsum = one_pass[0], one_pass[1]
        grad_b2 = colsum.to(ctx.bias_dtypes[1]) if (ctx.has_bias[1] and need_b2) else None
    else:                                              # (class counts the kernel does not take)
        g = grad.float()
        gp = g - logp.float().exp() * g.sum(1, keepdim=True)
        grad_b2 = gp.sum(0).to(ctx.bias_dtypes[1]) if (ctx.has_bias[1] and need_b2) else None
        gp = gp.to(dt)
    del grad
    grad_w1 = grad_w2 = grad_b1 = grad_x = None
    if not (need_x or need_w1 or need_b1 or need_w2):
        return grad_x, grad_w1, grad_b1, grad_w2, grad_b2
    # exact max|grad_sup2| from the product itself (gcn_epilogue.c_absmax); the analytic
    # ‖Âᵀ‖∞·max|g| is loose by the hub column sums and would cost the scaled GEMMs their low-order
    # bits, a separate reduction over [10⁷, 256] costs 2 ms
    gs_max = torch.zeros(1, dtype=torch.float32, device=dev) if f32 else None
    grad_sup2 = spmm_csr(graph_t, gp.contiguous(), tag="bwd_l2", c_absmax=gs_max)
    del gp
    gs_bound = gs_max * 1.0001 if f32 else None
    fast = f32 and _spmm.gemm_handwritten() and grad_sup2.shape[1] == 256 and h1.shape[1] == 256
    if need_w2:
        grad_w2 = _spmm.weight_grad_rows(h1, grad_sup2, None, None, ctx.h_bound, gs_bound) \
            if (fast or not f32) else None
        if grad_w2 is None:
            grad_w2 = _weight_grad(h1, grad_sup2)
    gh_max = torch.zeros(1, dtype=torch.float32, device=dev) if f32 else None
    w2t = w2.t().contiguous()
    gpre1 = gemm_xw256(grad_sup2, w2t, gs_bound, gh_max, mask_src=h1, mask_scale=ctx.scale,
                       mask_bits=getattr(ctx, "keep_bits", None)) if fast else None
    if gpre1 is None and dt == torch.bfloat16:       # (C5: the bf16 GEMM carries the mask in its store too)
        gpre1 = _spmm.gemm_bf16(grad_sup2, w2t, mask_src=h1, mask_scale=ctx.scale)
    if gpre1 is None:
        gh1 = _dense_forward(grad_sup2, w2t, gs_bound, gh_max)
        gpre1 = _spmm.relu_dropout_backward(gh1.contiguous(), h1, ctx.scale)
        if gh_max is not None:
            gh_max = gh_max * ctx.scale
        del gh1
    del grad_sup2
    gpre_bound = gh_max if f32 else None
    want_b1 = ctx.has_bias[0] and need_b1
    if want_b1 and ctx.reassoc and need_w1 and fast and x.shape[1] == 256:
        both = _spmm.weight_grad_rows(x, gpre1, None, None, ctx.z_bound, gpre_bound, colsum_g=True)   # (one pass for both)
        if both is not None:
            grad_w1, grad_b1 = both[0], both[1].to(ctx.bias_dtypes[0])
    if want_b1 and grad_b1 is None:
        sums = _spmm.backward_with_colsum(gpre1) if gpre1.is_contiguous() else None   # (one HIP pass)
        grad_b1 = (sums[1] if sums is not None else gpre1.float().sum(0)).to(ctx.bias_dtypes[0])
    if ctx.reassoc:
        z = x
        if need_w1 and grad_w1 is None:
            grad_w1 = _spmm.weight_grad_rows(z, gpre1, None, None, ctx.z_bound, gpre_bound) \
                if ((fast and z.shape[1] == 256) or not f32) else None
            if grad_w1 is None:
                grad_w1 = _weight_grad(z, gpre1)
        if need_x:
            gz = _dense_forward(gpre1, w1.t().contiguous(), gpre_bound)
            grad_x = spmm_csr(graph_t, gz, tag="bwd_l1")
    elif need_x or need_w1:
        grad_sup1 = spmm_csr(graph_t, gpre1.contiguous(), tag="bwd_l1")
        if need_w1:
            grad_w1 = _weight_grad(x, grad_sup1)
        if need_x:
            grad_x = _dense_forward(grad_sup1, w1.t().contiguous())
    return grad_x, grad_w1, grad_b1, grad_w2, grad_b2


class GCN2RowsFunction(torch.autograd.Function):
    """log_softmax(Â·dropout(relu(Â·X·W1 + b1))·W2 + b2)[rows] — models.py:47-71 (upstream form) —
    with the whole backward pass of the module docstring.  Outputs: (out_rows, full log-probability
    matrix or None); the second output is not differentiable."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, graph, rows, dropout_p, seed, keep_full):
        ctx.rs = row_sets(graph, rows)
        saved_x, h1, logp = _gcn2_forward(ctx, x, w1, b1, w2, b2, graph, dropout_p, seed)
        out_rows = logp.index_select(0, rows.to(torch.int64))
        ctx.save_for_backward(saved_x, w1, w2, h1, out_rows)
        if keep_full:
            ctx.mark_non_differentiable(logp)
            return out_rows, logp
        return out_rows, None

    @staticmethod
    def backward(ctx, grad_rows, _grad_full):
        x, w1, w2, h1, out_rows = ctx.saved_tensors          # (x is z = Â·X on the reassociated path)
        grads = _gcn2_backward_rows(ctx, x, w1, w2, h1, out_rows, ctx.rs, grad_rows,
                                    ctx.needs_input_grad[:5])
        return (*grads, None, None, None, None, None)


class GCN2Function(torch.autograd.Function):
    """`model(features, adj)` — the reference call, log-probabilities of EVERY vertex
    (models.py:47-71 upstream form) — as one autograd node.  What the backward pass does depends on
    the gradient that arrives:

      * a `RowGrad` (the caller selected `output[idx_train]`, pygcn/train.py:153 — pygcn_amd/rowgrad.py):
        the row-restricted pass of the module docstring, on the cached row sets of (graph, idx);
      * an `NLLGrad` (pygcn_amd.functional.nll_loss over all rows) or any dense tensor: the
        full-height pass of `_gcn2_backward_dense`.
    Neither synchronises with the host."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, graph, dropout_p, seed):
        saved_x, h1, logp = _gcn2_forward(ctx, x, w1, b1, w2, b2, graph, dropout_p, seed)
        ctx.save_for_backward(saved_x, w1, w2, h1, logp)
        return logp

    @staticmethod
    def backward(ctx, grad):
        from .rowgrad import RowGrad
        x, w1, w2, h1, logp = ctx.saved_tensors
        needs = ctx.needs_input_grad[:5]
        if isinstance(grad, RowGrad) and grad.rows.numel() and grad.values.dtype == logp.dtype:
            rs = row_sets(ctx.graph, grad.rows)
            grads = _gcn2_backward_rows(ctx, x, w1, w2, h1, logp.index_select(0, rs.rows_user), rs,
                                        grad.values, needs)
        else:
            if isinstance(grad, RowGrad):
                grad = grad.dense()
            grads = _gcn2_backward_dense(ctx, x, w1, w2, h1, logp, grad, needs)
        return (*grads, None, None, None)


def gcn2_rows(x, gc1, gc2, graph, rows, dropout_p, seed, keep_full=False):
    """(output[rows], full output or None) of the 2-layer model through the one-node path."""
    return GCN2RowsFunction.apply(x, gc1.weight, gc1.bias, gc2.weight, gc2.bias, graph, rows,
                                  float(dropout_p), seed, bool(keep_full))


def gcn2_full(x, gc1, gc2, graph, dropout_p, seed):
    """Log-probabilities of every vertex through the one-node path (GCN2Function)."""
    return GCN2Function.apply(x, gc1.weight, gc1.bias, gc2.weight, gc2.bias, graph,
                              float(dropout_p), seed)
