"""The sparse x dense product of the GraphConvolution layer as a HIP custom op with autograd.

Replaces `torch.spmm(adj, support)` (reference pygcn/layers.py:34) and the product
`adj.t() @ grad_output` that autograd runs for it (pygcn/train.py:157), plus — optionally fused
into the kernel's store — `output + self.bias` (pygcn/layers.py:35-36) and the ReLU that follows
the layer in the model (upstream form, pygcn/models.py:48).

Every call goes through the C-ABI `gcn_spmm_csr` (include/gcn_spmm.h).  No CPU / PyTorch
fallback exists: non-HIP tensors or a missing library raise.
"""
import torch

from . import _native, tuning
from .graph import CSRGraph, _require_cuda, as_graph
from .tuning import K_SPLIT, MIN_ROWS, REASSOC_MAX_WIDTH_RATIO      # (dispatch thresholds: one table)

_DTYPES = {torch.float32: _native.GCN_DTYPE_F32, torch.bfloat16: _native.GCN_DTYPE_BF16}

# Optional launch timing for bench.py: when a list is installed here, every gcn_spmm_csr launch
# is bracketed by HIP events recorded on the launch stream and (tag, start, end, graph) appended.
_timing_records = None

# Test switch: tensors whose unflagged rows are deliberately left unwritten (skip_zero_rows,
# c_select) are pre-filled with NaN instead of being allocated empty, so a consumer that reads a
# row it must not read shows up as NaN in the gradients (tests/test_spmm_gpu.py).
_poison_unwritten = False


def _maybe_poisoned(shape, dtype, device):
    if _poison_unwritten:
        return torch.full(shape, float("nan"), dtype=dtype, device=device)
    return torch.empty(shape, dtype=dtype, device=device)


def set_timing_records(records):
    """Install (or remove, with None) the list that receives per-launch HIP event pairs."""
    global _timing_records
    _timing_records = records


def spmm_csr(graph, B, bias=None, relu=False, out=None, tag="fwd", dropout_p=0.0, seed=0,
             b_hint=None, B2=None, c_flags=None, log_softmax=False, c_select=None,
             skip_zero_rows=False, row_base=0, c_absmax=None):
    """C = A · B (+ bias, ReLU, inverted dropout) on the current HIP stream; A is a CSRGraph,
    B dense [n_cols, F].  The epilogue order is that of the reference model: bias
    (layers.py:35-36), F.relu (models.py:48), F.dropout (models.py:50).  `b_hint` = (row bitmap
    int32 [ceil(n_cols/32)], nnz_rows int32 [1]) device tensors from backward_with_colsum (or
    row_bitmap()): rows of B whose bit is clear are not gathered (same result, less traffic).
    `B2`: optional second block of the dense operand — the operand is then [B; B2] stacked by rows
    without being materialised (the sharded path's own rows | halo rows).
    `c_flags`: optional uint8 [n_rows] tensor of ZEROS; the kernel sets c_flags[r] = 1 where the
    stored row r has a non-zero element (gcn_epilogue.c_row_nonzero).
    `log_softmax`: store log_softmax over each row of A·B + bias (`F.log_softmax(x, dim=1)`, the
    reference model's last line) — see log_softmax_fusable() for the shapes that allow it.
    `skip_zero_rows` (with c_flags): rows of the result that are entirely zero are not stored at
    all — only the flagged rows of the returned tensor are defined.
    `c_select`: optional int32 bitmap [ceil(n_rows/32)] over the OUTPUT rows: rows whose bit is
    clear are not wanted and may be left unwritten (gcn_epilogue.c_row_select).
    `seed` may be a 1-element int64 DEVICE tensor: the kernel then reads the seed when it executes
    (hipGraph replays draw a fresh mask if the graph updates the tensor, see dropout_seed_for).
    `row_base`: added to the row index in the dropout counter — a row-block shard passes the global
    index of its first row and draws the masks of the single-GPU run.
    `c_absmax`: optional DEVICE float32 [1], zeroed by the caller: receives max|stored value| of the
    launch (gcn_epilogue.c_absmax) — the bound a scaled GEMM consuming the result needs, without a
    reduction pass; non-finite if the result holds inf / NaN."""
    if not isinstance(graph, CSRGraph):
        raise RuntimeError("spmm_csr: graph must be a CSRGraph")
    _require_cuda(B, "dense operand")
    n_b = B.shape[0] + (B2.shape[0] if B2 is not None else 0)
    if B.dim() != 2 or n_b != graph.shape[1]:
        raise RuntimeError(f"size mismatch, adj {graph.shape} x dense {(n_b,) + tuple(B.shape[1:])}")
    if B2 is not None:
        if B2.dim() != 2 or B2.shape[1] != B.shape[1] or B2.dtype != B.dtype or B2.device != B.device:
            raise RuntimeError("B2 must match B in width, dtype and device")
        if B2.shape[1] > 0 and B2.stride(1) != 1:
            B2 = B2.contiguous()
    if B.dtype not in _DTYPES:
        raise RuntimeError(f"spmm_csr supports float32 and bfloat16, got {B.dtype}")
    if B.device != graph.device:
        raise RuntimeError(f"adj is on {graph.device} but dense operand on {B.device}")
    if B.shape[1] > 0 and B.stride(1) != 1:
        B = B.contiguous()
    n_rows, F = graph.shape[0], B.shape[1]
    if out is None:
        out = torch.empty((n_rows, F), dtype=B.dtype, device=B.device)
    elif out.shape != (n_rows, F) or out.dtype != B.dtype or out.stride(1) != 1:
        raise RuntimeError("spmm_csr: bad `out`")
    if c_flags is not None and (c_flags.dtype != torch.uint8 or c_flags.numel() != n_rows
                                or not c_flags.is_contiguous() or c_flags.device != B.device):
        raise RuntimeError("spmm_csr: c_flags must be a contiguous uint8 [n_rows] device tensor")
    if c_select is not None and (c_select.dtype != torch.int32 or c_select.device != B.device
                                 or c_select.numel() != (n_rows + 31) // 32
                                 or not c_select.is_contiguous()):
        raise RuntimeError("spmm_csr: c_select must be a contiguous int32 bitmap [ceil(n_rows/32)]")
    if c_absmax is not None and (c_absmax.dtype != torch.float32 or c_absmax.numel() != 1
                                 or c_absmax.device != B.device):
        raise RuntimeError("spmm_csr: c_absmax must be one float32 on the operand's device")
    if n_rows == 0 or F == 0:
        return out
    if bias is not None:
        _require_cuda(bias, "bias")
        bias = bias.detach().to(torch.float32).contiguous()
        if bias.numel() != F:
            raise RuntimeError("bias must have F entries")
    L = _native.lib()
    plan = graph.plan(B.dtype)
    ws_bytes = L.gcn_spmm_workspace_bytes(plan, F)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=B.device) if ws_bytes else None
    with torch.cuda.device(B.device):
        stream = torch.cuda.current_stream().cuda_stream
        rec = _timing_records
        if rec is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        seed_dev = None
        if isinstance(seed, torch.Tensor):
            if seed.dtype != torch.int64 or seed.numel() != 1 or seed.device != B.device:
                raise RuntimeError("spmm_csr: a tensor seed must be one int64 on the operand's device")
            seed_dev, seed = seed.data_ptr(), 0
        ep = _native.GcnEpilogue(bias.data_ptr() if bias is not None else None, int(bool(relu)),
                                 float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                 b_hint[0].data_ptr() if b_hint is not None else None,
                                 b_hint[1].data_ptr() if b_hint is not None else None,
                                 B2.data_ptr() if B2 is not None and B2.numel() else None,
                                 B2.stride(0) if B2 is not None and B2.numel() else 0,
                                 B.shape[0] if B2 is not None else 0,
                                 c_flags.data_ptr() if c_flags is not None else None,
                                 int(bool(log_softmax)), seed_dev,
                                 c_select.data_ptr() if c_select is not None else None,
                                 int(bool(skip_zero_rows) and c_flags is not None), int(row_base),
                                 c_absmax.data_ptr() if c_absmax is not None else None)
        rc = L.gcn_spmm_csr_ep(plan, _DTYPES[B.dtype], B.data_ptr(), B.stride(0) if F else 0,
                               out.data_ptr(), out.stride(0), F, ep,
                               ws.data_ptr() if ws is not None else None, ws_bytes, stream)
        if rec is not None:
            ev1.record()
            rec.append((tag, ev0, ev1, graph))
    _native.check(rc, "gcn_spmm_csr_ep")
    return out


def sddmm_csr(graph, G, B):
    """values[e] = < G[row(e), :], B[col[e], :] > for every stored entry e of `graph` (fp32 [nnz]) —
    the gradient of the adjacency VALUES of C = A · B for the gradient G of C (C-ABI gcn_sddmm_csr;
    PyTorch's `mm` derivative for a sparse first operand, sampled on A's pattern).  The reference
    never needs it (adj is a constant, pygcn/train.py:80,123); SURVEY row f4 lists it as optional."""
    if not isinstance(graph, CSRGraph):
        raise RuntimeError("sddmm_csr: graph must be a CSRGraph")
    _require_cuda(G, "G")
    _require_cuda(B, "B")
    if (G.dim() != 2 or B.dim() != 2 or G.shape[0] != graph.shape[0] or B.shape[0] != graph.shape[1]
            or G.shape[1] != B.shape[1] or G.dtype != B.dtype or G.dtype not in _DTYPES
            or G.device != graph.device or B.device != graph.device):
        raise RuntimeError(f"sddmm_csr: adj {graph.shape}, G {tuple(G.shape)} {G.dtype}, B {tuple(B.shape)} {B.dtype}")
    if G.shape[1] > 0 and G.stride(1) != 1:
        G = G.contiguous()
    if B.shape[1] > 0 and B.stride(1) != 1:
        B = B.contiguous()
    out = torch.empty(graph.nnz, dtype=torch.float32, device=G.device)
    if graph.nnz == 0:
        return out
    with torch.cuda.device(G.device):
        rc = _native.lib().gcn_sddmm_csr(graph.plan(), _DTYPES[G.dtype], G.data_ptr(), G.stride(0),
                                         B.data_ptr(), B.stride(0), G.shape[1], out.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "gcn_sddmm_csr")
    return out


def rows_pack(src, rows=None):
    """Rows `rows` (int64 index tensor; None = all) of `src` [*, F] (fp32 / bf16, F a multiple of 32)
    as BITMASK + NON-ZERO VALUES (C-ABI gcn_rows_pack_count / gcn_rows_pack_values; wire format of
    the compressed halo exchange, pygcn_amd/sharded.py): -> (bits int32 [m, F/32], offsets int64
    [m + 1] (exclusive scan of the per-row counts), vals [offsets[-1]]).  Sizing `vals` reads the
    total count on the host (one synchronisation; the exchange needs the counts there anyway)."""
    _require_cuda(src, "src")
    if src.dim() != 2 or src.dtype not in _DTYPES or src.shape[1] % 32 != 0 or src.shape[1] == 0:
        raise RuntimeError(f"rows_pack: src {tuple(src.shape)} {src.dtype}: 2-D fp32 / bf16, width a multiple of 32")
    if src.stride(1) != 1 or src.stride(0) % 4 != 0:
        src = src.contiguous()
    F, dev = src.shape[1], src.device
    if rows is not None:
        if rows.dtype != torch.int64 or rows.device != dev or rows.dim() != 1:
            raise RuntimeError("rows_pack: rows must be a 1-D int64 tensor on the device of src")
        rows = rows.contiguous()
    m = src.shape[0] if rows is None else rows.numel()
    bits = torch.empty((m, F // 32), dtype=torch.int32, device=dev)
    counts = torch.empty(m, dtype=torch.int32, device=dev)
    offsets = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    L, dt = _native.lib(), _DTYPES[src.dtype]
    rp = rows.data_ptr() if rows is not None else None
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        _native.check(L.gcn_rows_pack_count(dt, src.data_ptr(), src.stride(0), rp, m, F, bits.data_ptr(),
                                            counts.data_ptr(), st), "gcn_rows_pack_count")
        torch.cumsum(counts, 0, out=offsets[1:])
        vals = torch.empty(int(offsets[-1]), dtype=src.dtype, device=dev)
        _native.check(L.gcn_rows_pack_values(dt, src.data_ptr(), src.stride(0), rp, m, F, offsets.data_ptr(),
                                             vals.data_ptr(), st), "gcn_rows_pack_values")
    return bits, offsets, vals


def rows_unpack(bits, vals, F):
    """Inverse of rows_pack on the receiving side: bits int32 [m, F/32] and the non-zero values in
    row-major order -> dense [m, F] of vals.dtype (C-ABI gcn_bits_row_counts + gcn_rows_unpack; the
    offsets are the receiver's own scan of the bit counts, no host synchronisation)."""
    _require_cuda(bits, "bits")
    _require_cuda(vals, "vals")
    if (bits.dim() != 2 or bits.dtype != torch.int32 or bits.shape[1] * 32 != F or vals.dim() != 1
            or vals.dtype not in _DTYPES or vals.device != bits.device):
        raise RuntimeError(f"rows_unpack: bits {tuple(bits.shape)} {bits.dtype}, vals {tuple(vals.shape)} {vals.dtype}, F {F}")
    bits, vals = bits.contiguous(), vals.contiguous()
    m, dev = bits.shape[0], bits.device
    counts = torch.empty(m, dtype=torch.int32, device=dev)
    offsets = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    out = torch.empty((m, F), dtype=vals.dtype, device=dev)
    L = _native.lib()
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        _native.check(L.gcn_bits_row_counts(bits.data_ptr(), m, F // 32, counts.data_ptr(), st), "gcn_bits_row_counts")
        torch.cumsum(counts, 0, out=offsets[1:])
        _native.check(L.gcn_rows_unpack(_DTYPES[vals.dtype], bits.data_ptr(), offsets.data_ptr(), vals.data_ptr(),
                                        m, F, out.data_ptr(), out.stride(0), st), "gcn_rows_unpack")
    return out


def log_softmax_fusable(F, dtype):
    """True when log_softmax over rows of width F can run inside the SpMM's store: the whole row
    must sit in one wavefront (F <= 64, or 16-byte lanes with F / lane width <= 64; a freshly
    allocated support tensor satisfies the kernel's 16-byte alignment rule)."""
    if dtype not in _DTYPES:
        return False
    v = 128 // torch.finfo(dtype).bits        # elements per 16-byte lane
    return F <= 64 or (F % v == 0 and F // v <= 64)


def relu_dropout_backward(grad_out, out, scale=1.0):
    """grad_pre = out > 0 ? grad_out * scale : 0 in one streaming HIP pass
    (C-ABI gcn_relu_dropout_backward): backward of the fused ReLU (+ dropout) epilogue."""
    _require_cuda(grad_out, "grad_out")
    if grad_out.dtype not in _DTYPES or out.dtype != grad_out.dtype or out.shape != grad_out.shape:
        raise RuntimeError("relu_dropout_backward: dtype/shape mismatch")
    grad_out, out = grad_out.contiguous(), out.contiguous()
    res = torch.empty_like(grad_out)
    with torch.cuda.device(grad_out.device):
        rc = _native.lib().gcn_relu_dropout_backward(
            _DTYPES[grad_out.dtype], grad_out.data_ptr(), out.data_ptr(), res.data_ptr(),
            grad_out.numel(), float(scale), torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "gcn_relu_dropout_backward")
    return res


def backward_with_colsum(grad_out, out=None, scale=1.0, log_softmax=False, skip_zero_rows=False):
    """(grad_pre, column sums of grad_pre, row-sparsity hint) in ONE pass over fp32 / bf16 [N, F]
    tensors (C-ABI gcn_relu_dropout_backward_colsum); `out=None`: no masking, grad_pre is grad_out.
    `log_softmax=True`: `out` holds log-probabilities and grad_pre = grad_out - exp(out) *
    rowsum(grad_out) (C-ABI gcn_log_softmax_backward_colsum; rows need F / lane width <= 64).  The
    hint — (row bitmap int32 [ceil(N/32)], nnz_rows int32 [1]) or None when F > 256 — can be handed to
    spmm_csr(b_hint=...) when grad_pre is the dense operand of the following product.
    `skip_zero_rows=True` (needs `out` and a hint): rows of grad_pre that are entirely zero are
    NOT written — only the rows whose hint bit is set are defined; for consumers that read those
    rows only.
    Returns None when the shape/dtype is outside the kernel's envelope (caller falls back to
    relu_dropout_backward + torch's sum)."""
    _require_cuda(grad_out, "grad_out")
    L = _native.lib()
    if (grad_out.dtype not in _DTYPES or grad_out.dim() != 2
            or not grad_out.is_contiguous() or grad_out.shape[0] == 0
            or L.gcn_bwd_colsum_workspace_bytes(grad_out.shape[0], grad_out.shape[1],
                                                _DTYPES[grad_out.dtype]) == 0
            or (out is not None and (out.dtype != grad_out.dtype or not out.is_contiguous()
                                     or out.shape != grad_out.shape))
            or (log_softmax and (out is None
                                 or grad_out.shape[1] // (16 // grad_out.element_size()) > 64))):
        return None
    n, F = grad_out.shape
    grad_pre = _maybe_poisoned(grad_out.shape, grad_out.dtype, grad_out.device) if out is not None \
        else grad_out
    colsum = torch.empty(F, dtype=torch.float32, device=grad_out.device)
    hint = None
    if F <= (256 if grad_out.dtype == torch.float32 else 512):   # a row lives inside one wavefront
        hint = (torch.empty((n + 31) // 32, dtype=torch.int32, device=grad_out.device),   # bitmap
                torch.empty(1, dtype=torch.int32, device=grad_out.device))
    skip = int(bool(skip_zero_rows) and hint is not None and out is not None)
    ws_bytes = L.gcn_bwd_colsum_workspace_bytes(n, F, _DTYPES[grad_out.dtype])
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        if log_softmax:
            rc = L.gcn_log_softmax_backward_colsum(
                _DTYPES[grad_out.dtype], grad_out.data_ptr(), out.data_ptr(), grad_pre.data_ptr(),
                colsum.data_ptr(), n, F, hint[0].data_ptr(), hint[1].data_ptr(), skip,
                ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream)
        else:
            rc = L.gcn_relu_dropout_backward_colsum(
                _DTYPES[grad_out.dtype], grad_out.data_ptr(), out.data_ptr() if out is not None else None,
                grad_pre.data_ptr() if out is not None else None, colsum.data_ptr(), n, F, float(scale),
                hint[0].data_ptr() if hint else None, hint[1].data_ptr() if hint else None, skip,
                ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "gcn_log_softmax_backward_colsum" if log_softmax
                  else "gcn_relu_dropout_backward_colsum")
    return grad_pre, colsum.to(grad_out.dtype), hint


def nll_log_softmax_backward(logp, target, coef):
    """(grad_pre, column sums) for the gradient of a mean NLL loss over ALL rows of log-probabilities
    `logp` [n, F] (C-ABI gcn_nll_log_softmax_backward_colsum): grad_pre = coef·(onehot(target) −
    exp(logp)) and the bias gradient's column sums in ONE sweep that reads logp and the labels and
    writes grad_pre — the [n, F] loss gradient (one non-zero per row) is never materialised.
    `target` int64 [n] with entries in [0, F), `coef` DEVICE float32 [1].  None when the shape is
    outside the kernel's envelope (a row must sit in one wavefront)."""
    _require_cuda(logp, "logp")
    L = _native.lib()
    if (logp.dtype not in _DTYPES or logp.dim() != 2 or not logp.is_contiguous() or logp.shape[0] == 0
            or target.dtype != torch.int64 or target.numel() != logp.shape[0] or not target.is_contiguous()
            or target.device != logp.device or coef.dtype != torch.float32 or coef.numel() != 1
            or coef.device != logp.device
            or L.gcn_bwd_colsum_workspace_bytes(logp.shape[0], logp.shape[1], _DTYPES[logp.dtype]) == 0
            or logp.shape[1] // (16 // logp.element_size()) > 64):
        return None
    n, F = logp.shape
    grad_pre = torch.empty_like(logp)
    colsum = torch.empty(F, dtype=torch.float32, device=logp.device)
    ws_bytes = L.gcn_bwd_colsum_workspace_bytes(n, F, _DTYPES[logp.dtype])
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=logp.device)
    with torch.cuda.device(logp.device):
        rc = L.gcn_nll_log_softmax_backward_colsum(
            _DTYPES[logp.dtype], target.data_ptr(), coef.data_ptr(), logp.data_ptr(),
            grad_pre.data_ptr(), colsum.data_ptr(), n, F, ws.data_ptr(), ws_bytes,
            torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "gcn_nll_log_softmax_backward_colsum")
    return grad_pre, colsum.to(logp.dtype)


def pack_row_flags(nz):
    """bool [n] -> the operand hint (bitmap int32 [ceil(n/32)], count int32 [1]); torch ops."""
    n = nz.numel()
    pad = (-n) % 32
    if pad:
        nz = torch.cat([nz, torch.zeros(pad, dtype=torch.bool, device=nz.device)])
    w = (nz.view(-1, 32).to(torch.int64) << torch.arange(32, device=nz.device)).sum(1)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
    return w, nz.sum().to(torch.int32).reshape(1)


def unpack_row_flags(bits, n):
    """Inverse of pack_row_flags: bitmap int32 -> bool [n]."""
    sh = torch.arange(32, device=bits.device, dtype=torch.int32)
    return ((bits[:, None] >> sh) & 1).bool().reshape(-1)[:n]


def row_bitmap(B):
    """(bitmap, count) hint for a dense operand: bit r set iff row r of B has a non-zero element
    (torch ops; the backward pass gets the same thing for free from backward_with_colsum)."""
    return pack_row_flags((B != 0).any(1))


def _grad_pre_and_bias(grad_out, out, relu, scale, want_bias, log_softmax=False,
                       skip_zero_rows=False):
    """Shared by the single-GPU and sharded autograd functions: apply the fused-epilogue mask (or
    the log_softmax backward) and (optionally) produce the bias gradient, in one HIP pass when the
    shape allows.  Returns (grad_pre, grad_bias, row-sparsity hint or None).  With
    `skip_zero_rows` and a hint in the result, only the rows of grad_pre whose hint bit is set
    are defined (backward_with_colsum)."""
    grad_bias = None
    if want_bias or log_softmax:
        fused = backward_with_colsum(grad_out.contiguous(), out if (relu or log_softmax) else None,
                                     scale, log_softmax, skip_zero_rows and (relu or log_softmax))
        if fused is not None:
            return fused if want_bias else (fused[0], None, fused[2])
    if log_softmax:   # shapes outside the kernel's envelope (e.g. 7 classes): torch ops
        g32, o32 = grad_out.float(), out.float()
        grad_out = (g32 - o32.exp() * g32.sum(1, keepdim=True)).to(grad_out.dtype)
    if relu:
        grad_out = relu_dropout_backward(grad_out, out, scale)
    if want_bias:
        grad_bias = grad_out.sum(0)
    return grad_out, grad_bias, None


def dropout_scale(p):
    """1 / (keep probability) of the fused dropout at rate `p` — of the probability the kernels
    actually keep with: the threshold is 16 bits, T = clamp(round(p · 65536), 1, 65535), an element
    survives with probability (65536 − T) / 65536, and both the forward scale (C-ABI, struct
    gcn_epilogue) and the backward mask scale are 65536 / (65536 − T), so E[dropout(x)] = x
    exactly.  Equal to 1 / (1 − p) at p = 1/2 and within 2^-17 relative elsewhere."""
    if not p > 0.0:
        return 1.0
    import numpy as np
    t = int(np.float64(np.float32(p)) * 65536.0 + 0.5)
    thresh = min(65535, max(1, t))
    return float(np.float32(65536.0) / np.float32(65536 - thresh))


def next_dropout_seed(device=None):
    """64-bit seed of one fused-dropout launch, drawn the way `F.dropout` draws on a GPU in the
    reference model (pygcn/models.py:50 upstream): from the DEVICE's default generator — its seed
    (set by torch.manual_seed / torch.cuda.manual_seed) and its Philox offset, which is advanced —
    so it is reproducible under torch.manual_seed, needs no device synchronisation and leaves
    torch's CPU generator alone (a caller sharing that generator sees the stream the reference
    would give it).  CPU tensors (test stand-ins of the sharded path only) draw from the CPU
    generator, as F.dropout on CPU does."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
            else torch.device("cpu")
    if device.type != "cuda":
        return int(torch.empty((), dtype=torch.int64).random_().item())
    index = device.index if device.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[index]
    offset = gen.get_offset()
    gen.set_offset(offset + 4)          # one Philox counter step, like one dropout launch
    mixed = (gen.initial_seed() * 0x9E3779B97F4A7C15 + (offset // 4 + 1) * 0xD1B54A32D192ED03)
    return mixed & 0xFFFFFFFFFFFFFFFF


_device_seeds = {}


def dropout_seed_for(tensor):
    """Seed of one fused-dropout launch on `tensor`'s device.  Normally a host integer derived
    from the device generator's (seed, offset) (next_dropout_seed).  While the stream is being captured into a
    hipGraph a host seed would be frozen into the captured kernel arguments — the same mask on
    every replay — so the launch then reads a per-device int64 tensor that an op recorded in the
    same capture advances: every replay masks differently.  The tensor is created by the first
    eager dropout launch on the device (the warm-up steps torch requires before a capture)."""
    if not tensor.is_cuda:
        return next_dropout_seed(tensor.device)
    capturing = torch.cuda.is_current_stream_capturing()
    key = tensor.device.index if tensor.device.index is not None else torch.cuda.current_device()
    t = _device_seeds.get(key)
    if t is None:
        if capturing:
            raise RuntimeError("run one training step with dropout before capturing it into a "
                               "hipGraph (the device-resident dropout seed is created on first use)")
        # start value: the default generator's seed (torch.manual_seed), read without advancing
        # the generator — creating the tensor must not shift the stream of host seeds
        t = _device_seeds[key] = torch.full((1,), torch.initial_seed() & (2 ** 63 - 1),
                                            dtype=torch.int64, device=tensor.device)
    if not capturing:
        return next_dropout_seed(tensor.device)
    t.add_(0x9E3779B97F4A7C15 - (1 << 64))   # odd increment, wraps; recorded in the capture
    return t


class SpMMFunction(torch.autograd.Function):
    """out = dropout(relu(A · B + bias)) with every stage optional and fused into the kernel's
    store;  grad_B = A^T · grad_pre;  grad_bias = column sums of grad_pre, where
    grad_pre = grad_out, masked and scaled through `out > 0` when ReLU (+ dropout) was fused.
    `adj` never receives a gradient (it is a loaded constant in the reference: train.py:80,123).
    Only the graph handle (and the output, for a fused ReLU) is kept for backward — not B and no
    dropout mask."""

    @staticmethod
    def forward(ctx, graph, B, bias, relu=False, dropout_p=0.0, seed=0):
        if dropout_p > 0.0 and not relu:
            raise RuntimeError("fused dropout needs the fused ReLU (out > 0 encodes the mask)")
        ctx.graph = graph
        ctx.has_bias = bias is not None
        ctx.relu = bool(relu)
        ctx.scale = dropout_scale(dropout_p)
        out = spmm_csr(graph, B, bias=bias, relu=relu, dropout_p=dropout_p, seed=seed)
        if relu:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        grad_B = None
        out = ctx.saved_tensors[0] if ctx.relu else None
        grad_out, grad_bias, hint = _grad_pre_and_bias(grad_out, out, ctx.relu, ctx.scale,
                                                       ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.needs_input_grad[1]:
            # gradients of a loss on few labelled rows are non-zero on few rows: the hint lets the
            # transpose product skip the all-zero rows of its dense operand
            grad_B = spmm_csr(ctx.graph.t(), grad_out.contiguous(), tag="bwd", b_hint=hint)
        return None, grad_B, grad_bias, None, None, None


def gemm_keep_bits_usable(X, rows=None, dropout_p=0.0):
    """Does gemm_xw256 run this operand on the kernel that can write / read the ONE-BIT form of a ReLU /
    dropout result (C-ABI gcn_gemm_epilogue.keep_bits_out / mask_bits: contiguous rows, the three-part
    scheme, dropout_p in {0, 1/2})?"""
    return (_gemm_scheme == "bf16x3" and rows is None and X.dtype == torch.float32 and X.is_cuda and X.dim() == 2
            and X.shape[1] == 256 and X.stride(1) == 1 and X.stride(0) < (1 << 21) and dropout_p in (0.0, 0.5))


def gemm_xw256(X, W, x_bound=None, y_absmax=None, rows=None, mask_src=None, mask_scale=1.0,
               bias=None, relu=False, dropout_p=0.0, seed=0, mask_rows=None, row_base=0,
               keep_bits_out=None, mask_bits=None):
    """X[M,256] · W[256,256] through the hand-written MFMA kernels (fp32 in/out, fp32-level
    accuracy).  None if the operands do not fit the kernels' fixed shape / alignment (the caller
    then uses torch.mm — hipBLASLt).

    Default scheme "bf16x3": C-ABI gcn_gemm_xw256_f32_b3 — three bf16 parts per operand, six MFMAs
    per product: a 24-bit significand, the fp32-equivalent of the reference's `torch.mm`; no scaling,
    `x_bound` is ignored.  Scheme "h2" (set_gemm_scheme; 22-bit significand, half the matrix work):
    C-ABI gcn_gemm_xw256_f32_h2 — power-of-two scaling + two fp16 parts,
    three MFMAs per product.  `x_bound` (DEVICE float tensor [1]) is any upper bound of max|X|; if
    the caller has none, max|X| is computed here by one reduction pass.  `y_absmax` (DEVICE float
    tensor [1], zeroed by the caller) receives max|Y|, from which a layer derives the next bound
    without a pass over the data.  `rows` (int32 device list): output row r is the product of
    input row rows[r] — a gather fused into the kernel's loads.  `mask_src` ([*, 256] fp32, read
    at the same input rows, or — `mask_rows`, an int32 device list — at mask_rows[r] for output
    row r): the store becomes mask_src > 0 ? y * mask_scale : 0, the backward of a
    fused ReLU / dropout epilogue, in the GEMM's own store (None if it cannot be fused).
    `bias` / `relu` / `dropout_p` / `seed`: FORWARD epilogue in the store, y = dropout(relu(acc +
    bias)) with the same Philox keep function as the SpMM epilogue — for a layer evaluated as
    (Â·X)·W + b, whose last stage is the GEMM (None if it cannot be fused).
    `keep_bits_out` (int32 [M, 8], with relu; only where gemm_keep_bits_usable()): the launch also writes
    `out > 0` as one bit per element; `mask_bits` (such a tensor, given NEXT TO mask_src): the backward mask
    is read from the bits (32 bytes per row instead of 1 KiB) where the launch can, from mask_src where not.
    Both schemes carry every option; "bf16x3" keeps full accuracy for 1e-30 <= |x| <= 3e38 (below
    that its low-order parts underflow — tests/test_gemm_gpu.py)."""
    if (_gemm_scheme == "exact" or X.dtype != torch.float32 or W.dtype != torch.float32 or not X.is_cuda
            or X.dim() != 2 or tuple(W.shape) != (256, 256) or X.shape[1] != 256 or X.shape[0] == 0
            or X.stride(1) != 1 or X.stride(0) % 4 or X.data_ptr() % 16 or W.stride(1) != 1):
        return None
    L = _native.lib()
    has_fwd_ep = bias is not None or relu or dropout_p > 0.0
    if has_fwd_ep and (mask_src is not None
                       or (bias is not None and (bias.dtype != torch.float32 or bias.numel() != 256
                                                 or not bias.is_contiguous() or bias.data_ptr() % 16))):
        return None
    if mask_src is not None and (mask_src.dtype != torch.float32
                                 or mask_src.dim() != 2 or mask_src.shape[1] != 256
                                 or mask_src.stride(1) != 1 or mask_src.stride(0) % 4
                                 or mask_src.data_ptr() % 16 or mask_src.device != X.device):
        return None
    if mask_rows is not None and (mask_rows.dtype != torch.int32 or not mask_rows.is_contiguous()
                                  or mask_rows.device != X.device
                                  or mask_rows.numel() < (rows.numel() if rows is not None else X.shape[0])):
        raise RuntimeError("gemm_xw256: mask_rows must be a contiguous int32 device list, one entry per output row")
    if rows is not None:
        if rows.dtype != torch.int32 or not rows.is_contiguous() or rows.device != X.device:
            raise RuntimeError("gemm_xw256: rows must be a contiguous int32 device tensor")
    m_out = rows.numel() if rows is not None else X.shape[0]
    Y = torch.empty((m_out, 256), dtype=torch.float32, device=X.device)
    if m_out == 0:
        return Y
    for t, name in ((keep_bits_out, "keep_bits_out"), (mask_bits, "mask_bits")):
        if t is not None and (t.dtype != torch.int32 or t.dim() != 2 or t.shape[1] != 8 or not t.is_contiguous()
                              or t.device != X.device):
            raise RuntimeError(f"gemm_xw256: {name} must be a contiguous int32 [rows, 8] device tensor")
    if keep_bits_out is not None and not (relu and keep_bits_out.shape[0] >= m_out
                                          and gemm_keep_bits_usable(X, rows, dropout_p)
                                          and Y.stride(0) < (1 << 21)):
        raise RuntimeError("gemm_xw256: keep_bits_out needs relu, contiguous rows, the bf16x3 scheme and "
                           "dropout_p in {0, 1/2} (gemm_keep_bits_usable)")
    if mask_bits is not None and (mask_src is None or not gemm_keep_bits_usable(X, rows)):
        mask_bits = None                      # (this launch reads the mask itself)
    if x_bound is None and _gemm_scheme == "h2":
        # no bound known: one reduction pass over X (1.4 ms at M = 10^7) and the 5.4 ms kernel
        # still beat the 7.5 ms three-part kernel — and keep full accuracy for tiny operands,
        # where the third bf16 part would fall into the denormals
        x_bound = torch.linalg.vector_norm(X.detach(), ord=float("inf")).reshape(1)
    with torch.cuda.device(X.device):
        stream = torch.cuda.current_stream().cuda_stream
        ep = None
        if has_fwd_ep or mask_src is not None:
            seed_dev = None
            by_bits = mask_bits is not None
            if isinstance(seed, torch.Tensor):       # device-resident seed (hipGraph capture)
                seed_dev, seed = seed.data_ptr(), 0
            ep = _native.GcnGemmEpilogue(
                bias.detach().data_ptr() if bias is not None else None, int(bool(relu)),
                float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF, seed_dev,
                mask_src.data_ptr() if (mask_src is not None and not by_bits) else None,
                mask_src.stride(0) if (mask_src is not None and not by_bits) else 0, float(mask_scale),
                mask_rows.data_ptr() if (mask_rows is not None and mask_src is not None) else None,
                int(row_base),
                keep_bits_out.data_ptr() if keep_bits_out is not None else None,
                mask_bits.data_ptr() if by_bits else None)
        if _gemm_scheme == "h2":
            if x_bound.dtype != torch.float32 or x_bound.numel() != 1 or x_bound.device != X.device:
                raise RuntimeError("gemm_xw256: x_bound must be one float32 on the operand's device")
            ws_bytes = L.gcn_gemm_xw256_h2_workspace_bytes()
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device)
            rc = L.gcn_gemm_xw256_f32_h2(X.data_ptr(), X.stride(0),
                                         rows.data_ptr() if rows is not None else None,
                                         W.data_ptr(), W.stride(0),
                                         Y.data_ptr(), Y.stride(0), m_out, x_bound.data_ptr(),
                                         y_absmax.data_ptr() if y_absmax is not None else None,
                                         ep, ws.data_ptr(), ws_bytes, stream)
            _native.check(rc, "gcn_gemm_xw256_f32_h2")
            if _bound_check and y_absmax is not None and not bool(torch.isfinite(y_absmax).all()):
                raise RuntimeError("gemm_xw256: non-finite output — x_bound was smaller than max|X| (the "
                                   "fp16 parts overflowed) or the operands hold inf / NaN")
            return Y
        ws_bytes = L.gcn_gemm_xw256_b3_workspace_bytes()
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device)
        rc = L.gcn_gemm_xw256_f32_b3(X.data_ptr(), X.stride(0),
                                     rows.data_ptr() if rows is not None else None,
                                     W.data_ptr(), W.stride(0), Y.data_ptr(), Y.stride(0), m_out,
                                     y_absmax.data_ptr() if y_absmax is not None else None,
                                     ep, ws.data_ptr(), ws_bytes, stream)
    _native.check(rc, "gcn_gemm_xw256_f32_b3")
    return Y


def gemm_bf16(X, W, bias=None, relu=False, dropout_p=0.0, seed=0, row_base=0, mask_src=None,
              mask_rows=None, mask_scale=1.0):
    """X[M,K] · W[K,N] for bf16 storage through the streaming MFMA kernel (C-ABI gcn_gemm_xw_bf16;
    (K, N) in {(128,128), (128,256), (256,128)} — config C5's layers are 128 -> 128).
    `bias` / `relu` / `dropout_p` / `seed`: the layer's FORWARD epilogue on the fp32 accumulators
    before the rounding to bf16 (same Philox keep function as the SpMM epilogue) — for a layer
    evaluated as (Â·X)·W + b.  `mask_src` (bf16 [*, N], read at row mask_rows[r] — an int32 device
    list — or r for output row r): the store becomes mask_src > 0 ? y * mask_scale : 0, the backward
    of a fused ReLU / dropout epilogue in the grad_input GEMM's own store (excludes the forward
    epilogue).  None if the operands do not fit (the caller then uses torch.mm)."""
    if (X.dtype != torch.bfloat16 or W.dtype != torch.bfloat16 or not X.is_cuda or X.dim() != 2
            or W.dim() != 2 or X.shape[1] != W.shape[0] or X.shape[0] == 0 or X.stride(1) != 1
            or W.stride(1) != 1 or X.stride(0) % 8 or X.data_ptr() % 16):
        return None
    L = _native.lib()
    K, N = W.shape
    ws_bytes = L.gcn_gemm_bf16_workspace_bytes(K, N)
    if ws_bytes == 0:
        return None
    ep = bias32 = None
    if mask_src is not None:
        if (bias is not None or relu or dropout_p > 0.0 or mask_src.dtype != torch.bfloat16
                or mask_src.dim() != 2 or mask_src.shape[1] != N or mask_src.stride(1) != 1
                or mask_src.stride(0) % 8 or mask_src.data_ptr() % 16 or mask_src.device != X.device):
            return None
        if mask_rows is not None and (mask_rows.dtype != torch.int32 or not mask_rows.is_contiguous()
                                      or mask_rows.device != X.device or mask_rows.numel() < X.shape[0]):
            raise RuntimeError("gemm_bf16: mask_rows must be a contiguous int32 device list, one entry per output row")
        ep = _native.GcnGemmEpilogue(None, 0, 0.0, 0, None, mask_src.data_ptr(), mask_src.stride(0),
                                     float(mask_scale), mask_rows.data_ptr() if mask_rows is not None else None, 0)
    elif bias is not None or relu or dropout_p > 0.0:
        if bias is not None:
            if bias.numel() != N or bias.device != X.device:
                return None
            bias32 = bias.detach().to(torch.float32).contiguous()     # (kept alive past the launch)
        seed_dev = None
        if isinstance(seed, torch.Tensor):       # device-resident seed (hipGraph capture)
            seed_dev, seed = seed.data_ptr(), 0
        ep = _native.GcnGemmEpilogue(bias32.data_ptr() if bias32 is not None else None, int(bool(relu)),
                                     float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF, seed_dev, None, 0, 1.0, None,
                                     int(row_base))
    Y = torch.empty((X.shape[0], N), dtype=torch.bfloat16, device=X.device)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device)
    with torch.cuda.device(X.device):
        rc = L.gcn_gemm_xw_bf16(X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), Y.data_ptr(),
                                Y.stride(0), X.shape[0], K, N, ep, ws.data_ptr(), ws_bytes,
                                torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "gcn_gemm_xw_bf16")
    return Y


def layer_gemm_reassociable(x, weight, bias):
    """Can `epilogue((A·x)·W + b)` run with the epilogue in a hand-written GEMM's store?  fp32
    256 -> 256 (gcn_gemm_xw256_f32_h2) or bf16 storage at the streaming kernel's shapes with
    Fin <= Fout (the product A·x then is no wider than A·(x·W))."""
    if x.dim() != 2 or not x.is_cuda or x.stride(1) != 1 or weight.dim() != 2 or x.dtype != weight.dtype:
        return False
    if x.dtype == torch.float32:
        return (_gemm_scheme != "exact" and tuple(weight.shape) == (256, 256) and x.shape[1] == 256
                and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous())))
    if x.dtype == torch.bfloat16:
        return (tuple(weight.shape) in ((128, 128), (128, 256)) and x.shape[1] == weight.shape[0]
                and weight.stride(1) == 1)
    return False


def layer_gemm(z, weight, z_bound=None, y_absmax=None, bias=None, relu=False, dropout_p=0.0, seed=0,
               row_base=0, keep_bits_out=None):
    """epilogue(z·W + b) through the kernel layer_gemm_reassociable() promised (None if it declines)."""
    if z.dtype == torch.float32:
        return gemm_xw256(z, weight, z_bound, y_absmax, bias=bias, relu=relu, dropout_p=dropout_p,
                          seed=seed, row_base=row_base, keep_bits_out=keep_bits_out)
    return gemm_bf16(z, weight, bias=bias, relu=relu, dropout_p=dropout_p, seed=seed, row_base=row_base)


_identity_lists = {}


def padded_row_list(rows):
    """int32 copy of a row-index list, padded to a multiple of 16 entries by repeating its last
    entry (what gcn_gemm_atg256_f32 expects: the 16 indices of a step are one scalar load)."""
    r = rows.to(torch.int32)
    pad = (-r.numel()) % 16
    if pad and r.numel():
        r = torch.cat([r, r[-1:].expand(pad)])
    return r.contiguous()


def _identity_list(n, device):
    """0, 1, …, n-1 (padded) — cached per device, grown on demand."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    have = _identity_lists.get(key)
    need = (n + 15) // 16 * 16
    if have is None or have.numel() < need:
        have = _identity_lists[key] = torch.arange(need, dtype=torch.int32, device=device).clamp_(max=max(n - 1, 0))
        have._n = n
    if getattr(have, "_n", None) != n:       # the clamp of the padding depends on n
        have = torch.arange(need, dtype=torch.int32, device=device).clamp_(max=max(n - 1, 0))
        have._n = n
        _identity_lists[key] = have
    return have


def weight_grad_rows(A, G, rows_a=None, rows_g=None, a_bound=None, g_bound=None, n_list=None, colsum_g=False):
    """Σ_r A[rows_a[r]]ᵀ ⊗ G[rows_g[r]] through the gather-fused MFMA kernels: the weight gradient
    `inputᵀ · grad_support` over a LIST of rows, without compacting either operand first.
    fp32 [*, 256] x [*, 256] (C-ABI gcn_gemm_atg256_f32_b3: three bf16 parts, fp32-equivalent — or
    gcn_gemm_atg256_f32, the scaled two-part fp16 scheme, under set_gemm_scheme("h2")) or bf16
    storage [*, 128] x [*, 128] (C-ABI gcn_gemm_atg_bf16: fp32 accumulation, result rounded once
    to bf16).  rows_*: int32 device index lists or None (= all rows, in order).  A list may be
    longer than `n_list` (padding to a multiple of 16, padded_row_list()); unpadded lists are
    padded here.  *_bound (fp32 only): DEVICE float [1] upper bounds of max|A|, max|G| (computed
    here by a reduction pass over the listed rows when missing).  None if the operands do not fit
    a kernel.
    colsum_g=True (fp32, default scheme): returns (grad_w, Σ_r G[rows_g[r]] as fp32 [256]) — the layer's bias
    gradient from the rows the kernel loads anyway (C-ABI gcn_gemm_atg256_f32_b3_colsum); None where that
    form does not exist (the caller then sums G itself)."""
    bf16 = A.dtype == torch.bfloat16 and G.dtype == torch.bfloat16
    if colsum_g and (bf16 or _gemm_scheme != "bf16x3"):
        return None
    if bf16:
        if (not A.is_cuda or A.dim() != 2 or G.dim() != 2 or A.stride(1) != 1 or G.stride(1) != 1
                or A.stride(0) % 2 or G.stride(0) % 2 or A.data_ptr() % 4 or G.data_ptr() % 4
                or _native.lib().gcn_gemm_atg_bf16_workspace_bytes(16, A.shape[1], G.shape[1]) == 0):
            return None
    elif (_gemm_scheme == "exact" or A.dtype != torch.float32 or G.dtype != torch.float32 or not A.is_cuda
            or A.dim() != 2 or G.dim() != 2 or A.shape[1] != 256 or G.shape[1] != 256 or A.stride(1) != 1
            or G.stride(1) != 1 or A.stride(0) % 4 or G.stride(0) % 4 or A.data_ptr() % 16 or G.data_ptr() % 16):
        return None
    if n_list is None:
        n_a = rows_a.numel() if rows_a is not None else A.shape[0]
        n_g = rows_g.numel() if rows_g is not None else G.shape[0]
        if n_a != n_g:
            raise RuntimeError("weight_grad_rows: the two operands list different numbers of rows")
        n_list = n_a
    if n_list == 0:
        zero = torch.zeros((A.shape[1], G.shape[1]), dtype=A.dtype, device=A.device)
        return (zero, torch.zeros(G.shape[1], dtype=torch.float32, device=A.device)) if colsum_g else zero
    lists = []
    for r, t in ((rows_a, A), (rows_g, G)):
        if r is None:
            if t.shape[0] < n_list:
                raise RuntimeError("weight_grad_rows: operand has fewer rows than n_list")
            r = _identity_list(n_list, A.device)
        elif r.dtype != torch.int32 or not r.is_contiguous() or r.device != A.device:
            raise RuntimeError("weight_grad_rows: row lists must be contiguous int32 device tensors")
        elif r.numel() < (n_list + 15) // 16 * 16:
            r = padded_row_list(r[:n_list])
        lists.append(r)
    L = _native.lib()
    if bf16:
        K, N = A.shape[1], G.shape[1]
        out = torch.empty((K, N), dtype=torch.float32, device=A.device)
        ws_bytes = L.gcn_gemm_atg_bf16_workspace_bytes(n_list, K, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device)
        with torch.cuda.device(A.device):
            rc = L.gcn_gemm_atg_bf16(A.data_ptr(), A.stride(0), lists[0].data_ptr(), G.data_ptr(),
                                     G.stride(0), lists[1].data_ptr(), n_list, K, N, out.data_ptr(),
                                     out.stride(0), ws.data_ptr(), ws_bytes,
                                     torch.cuda.current_stream().cuda_stream)
        _native.check(rc, "gcn_gemm_atg_bf16")
        return out.to(torch.bfloat16)
    out = torch.empty((256, 256), dtype=torch.float32, device=A.device)
    ws_bytes = L.gcn_gemm_atg256_workspace_bytes(n_list)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device)
    if colsum_g:
        cs = torch.empty(256, dtype=torch.float32, device=A.device)
        with torch.cuda.device(A.device):
            rc = L.gcn_gemm_atg256_f32_b3_colsum(A.data_ptr(), A.stride(0), lists[0].data_ptr(),
                                                 G.data_ptr(), G.stride(0), lists[1].data_ptr(), n_list,
                                                 out.data_ptr(), out.stride(0), cs.data_ptr(), ws.data_ptr(),
                                                 ws_bytes, torch.cuda.current_stream().cuda_stream)
        _native.check(rc, "gcn_gemm_atg256_f32_b3_colsum")
        return out, cs
    if _gemm_scheme != "h2":            # three bf16 parts: no bounds
        with torch.cuda.device(A.device):
            rc = L.gcn_gemm_atg256_f32_b3(A.data_ptr(), A.stride(0), lists[0].data_ptr(),
                                          G.data_ptr(), G.stride(0), lists[1].data_ptr(), n_list,
                                          out.data_ptr(), out.stride(0), ws.data_ptr(), ws_bytes,
                                          torch.cuda.current_stream().cuda_stream)
        _native.check(rc, "gcn_gemm_atg256_f32_b3")
        return out
    # (no bound supplied: a reduction pass — over the LISTED rows only, the others may hold anything)
    if a_bound is None:
        src = A.detach()[:n_list] if rows_a is None else A.detach().index_select(0, rows_a[:n_list].long())
        a_bound = torch.linalg.vector_norm(src, ord=float("inf")).reshape(1)
    if g_bound is None:
        src = G.detach()[:n_list] if rows_g is None else G.detach().index_select(0, rows_g[:n_list].long())
        g_bound = torch.linalg.vector_norm(src, ord=float("inf")).reshape(1)
    with torch.cuda.device(A.device):
        rc = L.gcn_gemm_atg256_f32(A.data_ptr(), A.stride(0), lists[0].data_ptr(),
                                   G.data_ptr(), G.stride(0), lists[1].data_ptr(), n_list,
                                   a_bound.data_ptr(), g_bound.data_ptr(), out.data_ptr(), out.stride(0),
                                   ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "gcn_gemm_atg256_f32")
    return out


_gemm_scheme = "bf16x3"
_bound_check = False


def gemm_scheme():
    """The current scheme of the fp32 256 -> 256 GEMMs (set_gemm_scheme)."""
    return _gemm_scheme


def gemm_handwritten():
    """Are the 256-wide fp32 GEMMs on the hand-written MFMA kernels (either decomposition)?"""
    return _gemm_scheme != "exact"


def gemm_needs_bounds():
    """Only the scaled two-part fp16 scheme needs an upper bound of max|operand|."""
    return _gemm_scheme == "h2"


def set_bound_check(enabled):
    """DEBUG switch: after every scaled GEMM that reports max|Y|, read it on the host (a stream
    synchronisation) and raise if it is not finite.  The kernels never hide an overflow — a bound
    that was too small makes y_absmax inf / NaN (its integer maximum keeps those patterns), and a
    consumer scaled by a non-finite bound stores NaN — so a wrong bound ends in a NaN loss, not in
    plausible numbers; this switch names the first launch that overflowed."""
    global _bound_check
    _bound_check = bool(enabled)



def set_gemm_scheme(name):
    """How the fp32 256 -> 256 GEMMs of the layers are evaluated:
    "bf16x3" (default since round 4): three bf16 parts per operand, six MFMAs per product — a 24-bit
        significand, the fp32-EQUIVALENT of the reference's `torch.mm` (pygcn/layers.py:33); no
        scaling, no bounds, fp32's range; forward (with the layer epilogue), grad_input (with the
        mask) and the gather-fused weight gradients;
    "h2": the scaled two-part fp16 MFMA kernels (22-bit significand, half the matrix work, 4e-7
        normwise vs fp64 on well-scaled data) wherever a bound of max|X| is known — opt-in;
    "exact": no hand-written fp32 GEMM at all — every dense product is `torch.mm` (hipBLASLt's exact
        fp32 MFMA path, the arithmetic of the reference's `torch.mm(input, self.weight)`,
        pygcn/layers.py:33) and the layers keep the reference's order Â·(X·W).  The SpMM kernels
        are the same in all three."""
    global _gemm_scheme
    if name not in ("h2", "bf16x3", "exact"):
        raise RuntimeError("gemm scheme must be 'h2', 'bf16x3' or 'exact'")
    _gemm_scheme = name


_absmax_cache = {}


def absmax_cached(t):
    """max|t| as a DEVICE float tensor [1], computed once per (tensor object, version): for operands
    that stay constant across steps (the feature matrix).  A few entries per device (an eval-mode
    forward pass sees every layer's input as "constant": they must not evict each other), each
    holding a weak reference to the tensor OBJECT — a new tensor that happens to reuse the storage
    address of a freed one can never inherit its bound (a bound that is too small would overflow
    the fp16 parts)."""
    import weakref
    key = t.device.index if t.device.index is not None else -1
    live = [e for e in _absmax_cache.get(key, []) if e[0]() is not None]
    for ref, version, val in live:
        if ref() is t and version == t._version:
            return val
    val = torch.linalg.vector_norm(t.detach(), ord=float("inf")).float().reshape(1)
    _absmax_cache[key] = live[-3:] + [(weakref.ref(t), t._version, val)]
    return val


_absmax_known = {}


def remember_absmax(t, value):
    """Record max|t| (a DEVICE float [1] a kernel produced as a side result, e.g. the GEMM's
    y_absmax) for the tensor OBJECT t at its current version, so that the next layer's GEMM needs
    no reduction pass over t.  A few entries per device; weak references, like absmax_cached."""
    import weakref
    key = t.device.index if t.device.index is not None else -1
    known = [e for e in _absmax_known.get(key, []) if e[0]() is not None][-3:]
    known.append((weakref.ref(t), t._version, value))
    _absmax_known[key] = known


def known_absmax(t):
    """The bound remember_absmax() recorded for this tensor object and version, or None."""
    key = t.device.index if t.device.index is not None else -1
    for ref, version, value in _absmax_known.get(key, ()):
        if ref() is t and version == t._version:
            return value
    return None


def _dense_forward(input, weight, x_bound=None, y_absmax=None):
    out = gemm_xw256(input, weight, x_bound, y_absmax)
    if out is None and y_absmax is None:
        out = gemm_bf16(input, weight)
    if out is None:
        out = torch.mm(input, weight)
        if y_absmax is not None:
            y_absmax.copy_(out.detach().abs().max())
    return out




def _weight_grad(input, grad, a_bound=None, g_bound=None):
    """inputᵀ · grad: the hand-written MFMA kernel for 256-wide fp32 layers — always under the
    three-part bf16 scheme; under "h2" when the caller knows bounds of both operands' maxima (the
    scaling needs them; two reduction passes over [N, 256] tensors would cost what the kernel saves)
    — otherwise hipBLASLt with the reduction over the graph's vertices cut into K_SPLIT slabs."""
    if (_gemm_scheme == "bf16x3" and input.dtype == torch.float32) or \
            (a_bound is not None and g_bound is not None and _gemm_scheme == "h2") or \
            (input.dtype == torch.bfloat16 and grad.dtype == torch.bfloat16 and input.is_cuda):
        out = weight_grad_rows(input, grad, a_bound=a_bound, g_bound=g_bound)
        if out is not None:
            return out
    n, b = input.shape[0], K_SPLIT
    if n >= MIN_ROWS and input.is_contiguous() and grad.is_contiguous():
        m = n // b * b
        grad_w = torch.bmm(input[:m].view(b, m // b, -1).transpose(1, 2),
                           grad[:m].view(b, m // b, -1)).sum(0)
        if m < n:
            grad_w = grad_w + torch.mm(input[m:].t(), grad[m:])
        return grad_w
    return torch.mm(input.t(), grad)


def _dense_grads(input, weight, grad, need_in, need_w, rows=None):
    """(grad_input, grad_weight) of `input @ weight`.  `rows` (int64 indices, sorted) names the
    only rows of `grad` that are non-zero: both GEMMs then run on those rows alone — zero rows
    add nothing to inputᵀ·grad and give zero rows of grad·weightᵀ."""
    grad_in = grad_w = None
    if rows is not None:
        grad = grad.index_select(0, rows)
        if need_w:
            grad_w = _weight_grad(input.index_select(0, rows), grad)
        if need_in:
            part = gemm_xw256(grad, weight.t().contiguous())
            if part is None:
                part = gemm_bf16(grad, weight.t().contiguous())
            if part is None:
                part = torch.mm(grad, weight.t())
            grad_in = torch.zeros((input.shape[0], weight.shape[0]), dtype=part.dtype,
                                  device=part.device)
            grad_in.index_copy_(0, rows, part)
        return grad_in, grad_w
    if need_in:
        y_max = torch.zeros(1, dtype=torch.float32, device=grad.device) \
            if (grad.is_cuda and grad.dtype == torch.float32) else None
        grad_in = gemm_xw256(grad, weight.t().contiguous(), None, y_max)
        if grad_in is not None and y_max is not None and _gemm_scheme != "exact":
            remember_absmax(grad_in, y_max)      # (the layer below bounds its masked gradient by it)
        if grad_in is None:
            grad_in = gemm_bf16(grad, weight.t().contiguous())
        if grad_in is None:
            grad_in = torch.mm(grad, weight.t())
    if need_w:
        grad_w = _weight_grad(input, grad)
    return grad_in, grad_w


class DenseMMFunction(torch.autograd.Function):
    """`torch.mm(input, weight)` (reference pygcn/layers.py:33) with a K-split weight gradient.

    grad_W = inputᵀ · grad is a [Fin, N]·[N, Fout] GEMM whose reduction runs over the N graph
    vertices (10⁷ at config C4).  hipBLASLt answers that shape with a stream-K kernel at 21.5 ms;
    cutting N into 128 slabs, one batched GEMM over the slabs and a sum of the 128 small partial
    products takes 8.7 ms on MI355X (tools/gemm_probe.py) and is at least as accurate (shorter
    fp32 accumulation chains).

    Forward and grad_input use the hand-written MFMA kernels when the layer is 256 -> 256 fp32
    (gemm_xw256: 5.2 ms vs hipBLASLt 9.95 ms at N = 10⁷) or one of the bf16 shapes of gemm_bf16,
    torch.mm otherwise.  (The one-node training path, pygcn_amd/fused.py, forms the weight
    gradient with the gather-fused kernel weight_grad_rows instead.)"""

    K_SPLIT = K_SPLIT
    MIN_ROWS = MIN_ROWS

    @staticmethod
    def forward(ctx, input, weight):
        ctx.save_for_backward(input, weight)
        bound = known_absmax(input) if (input.is_cuda and input.dtype == torch.float32) else None
        return _dense_forward(input, weight, bound)

    @staticmethod
    def backward(ctx, grad):
        input, weight = ctx.saved_tensors
        return _dense_grads(input, weight, grad, ctx.needs_input_grad[0], ctx.needs_input_grad[1])


_row_compaction = False


def set_row_compaction(enabled):
    """OPT-IN (default off since round 3): row compaction of the layer-by-layer path's gradient
    GEMMs reads two device counters on the host (two stream synchronisations per layer backward,
    only for graphs of >= MIN_ROWS vertices).  The model-level paths need none of it — a loss on
    selected rows reaches the layers as a structural RowGrad (pygcn_amd/rowgrad.py, fused.py) — so
    by default no backward pass synchronises with the host; a caller that composes bare layers
    under a row-sparse dense gradient can switch this on (never active during hipGraph capture)."""
    global _row_compaction
    _row_compaction = bool(enabled)


def _hint_will_be_used(B, nz_rows):
    """Mirror of the device-side rule (use_row_flags in gcn_spmm.hip): the product skips the
    flagged-zero rows of its dense operand B below 3/4 non-zero rows in the wide kernel (16-byte
    lanes, more than 32 of them per row) and below 1/8 in the narrow one."""
    n, F = B.shape
    v = 16 // B.element_size()
    wide = (F % v == 0 and F // v > 32 and B.stride(1) == 1 and B.data_ptr() % 16 == 0
            and (B.stride(0) * B.element_size()) % 16 == 0)
    return tuning.below(nz_rows, n, tuning.HINT_WIDE_MAX_SHARE if wide else tuning.HINT_NARROW_MAX_SHARE)


class GraphConvFunction(torch.autograd.Function):
    """The whole layer — `torch.mm(input, weight)`, `torch.spmm(adj, support)`, `+ bias`
    (reference pygcn/layers.py:33-36) and the optional fused ReLU / dropout — as ONE autograd
    node, so that what the backward pass learns about the sparsity of its tensors can be used by
    the next stage instead of being re-discovered:

        grad_pre  = mask(grad_out)                 -> row bitmap + count   (one HIP pass)
        grad_sup  = Aᵀ · grad_pre   (skips zero rows of grad_pre) -> byte flags of ITS non-zero rows
        grad_W    = inputᵀ · grad_sup,  grad_input = grad_sup · Wᵀ   on the non-zero rows only

    With a loss on few labelled vertices (`nll_loss(output[idx_train], …)`, pygcn/train.py:153)
    grad_out of the last layer is non-zero on |idx_train| rows and grad_sup on their neighbourhood:
    5 % and 16 % of the rows at bench config C4."""

    @staticmethod
    def forward(ctx, input, weight, bias, graph, relu=False, dropout_p=0.0, seed=0,
                log_softmax=False):
        if dropout_p > 0.0 and not relu:
            raise RuntimeError("fused dropout needs the fused ReLU (out > 0 encodes the mask)")
        if log_softmax and relu:
            raise RuntimeError("log_softmax cannot be combined with the fused ReLU / dropout")
        ctx.graph = graph
        ctx.relu = bool(relu)
        ctx.log_softmax = bool(log_softmax)
        ctx.scale = dropout_scale(dropout_p)
        # (a layer input that needs no gradient is the constant feature matrix: its maximum is
        #  computed once and reused as the scaled GEMM's bound)
        # (bounds exist for the scaled fp16 scheme alone: the default three-part bf16 GEMMs need none)
        bounds = gemm_needs_bounds()
        x_bound = absmax_cached(input) if (bounds and input.dtype == torch.float32 and not input.requires_grad
                                           and input.is_cuda) else None
        const_input = not input.requires_grad and input.is_cuda
        if bounds and x_bound is None and input.dtype == torch.float32 and input.is_cuda:
            x_bound = known_absmax(input)       # (left by the layer that produced this tensor)
        # REASSOCIATED for a constant input on the shape the GEMM kernel carries the epilogue for
        # (256 -> 256 fp32): out = epilogue((A·input)·W + b).  Same two kernels and bytes in
        # forward; z = A·input of this forward pass is then all the backward pass needs for
        # grad_W = zᵀ·grad_pre — no sparse product in backward (pygcn_amd/fused.py does the same
        # inside the one-node path).
        ctx.reassoc = False
        if (const_input and not log_softmax and isinstance(graph, CSRGraph)
                and layer_gemm_reassociable(input, weight, bias)):
            z = spmm_csr(graph, input)
            ctx.z_bound = y_max = None
            if x_bound is not None:
                ctx.z_bound = graph.inf_norm() * x_bound * 1.0001
                y_max = torch.zeros(1, dtype=torch.float32, device=input.device)
            out = layer_gemm(z, weight, ctx.z_bound, y_max, bias=bias, relu=relu, dropout_p=dropout_p,
                             seed=seed)
            if out is not None:
                if y_max is not None:
                    remember_absmax(out, y_max)      # the next layer's GEMM scales by it
                ctx.reassoc = True
                ctx.save_for_backward(z, weight, *([out] if relu else []))
                return out
            del z
        ctx.x_bound = x_bound
        support = _dense_forward(input, weight, x_bound)
        out = spmm_csr(graph, support, bias=bias, relu=relu, dropout_p=dropout_p, seed=seed,
                       log_softmax=log_softmax)
        if relu or log_softmax:
            ctx.save_for_backward(input, weight, out)
        else:
            ctx.save_for_backward(input, weight)
        return out

    @staticmethod
    def _backward_rows(ctx, grad):
        """The backward pass for a row-sparse gradient (pygcn_amd/rowgrad.py): `grad.rows` are the only
        rows of grad_out that can be non-zero.  Returns the gradient tuple, or None where this
        route does not apply (the caller then continues with the dense tensor)."""
        from . import fused
        from .rowgrad import RowGrad
        input, weight = ctx.saved_tensors[:2]
        out = ctx.saved_tensors[2] if (ctx.relu or ctx.log_softmax) else None
        need_in, need_w, need_b = ctx.needs_input_grad[:3]
        graph, rows, vals = ctx.graph, grad.rows, grad.values
        if not isinstance(graph, CSRGraph) or rows.numel() == 0 or vals.dtype != input.dtype:
            return None
        dev, dt = vals.device, vals.dtype
        f32 = dt == torch.float32
        if ctx.log_softmax:
            # ---- last layer: log_softmax backward on the compact rows, then Âᵀ on its block
            rs = fused.row_sets(graph, rows)
            out_rows = out.index_select(0, rs.rows_user)
            one_pass = backward_with_colsum(vals.contiguous(), out_rows, log_softmax=True) \
                if not rs.has_duplicates else None
            if one_pass is not None:
                gp, colsum, _ = one_pass
                grad_bias = colsum if need_b else None
            else:
                g32 = vals.float()
                gp = (g32 - out_rows.float().exp() * g32.sum(1, keepdim=True)).to(dt)
                grad_bias = gp.float().sum(0).to(dt) if need_b else None
            if rs.has_duplicates:
                gp = torch.zeros((rs.n_u, gp.shape[1]), dtype=dt, device=dev).index_add_(0, rs.inverse, gp)
            elif not rs.sorted_unique:
                gp = torch.empty_like(gp).index_copy_(0, rs.inverse, gp)
            if not (need_in or need_w):
                return None, None, grad_bias, None, None, None, None, None
            grad_sup = spmm_csr(rs.at_block, gp.contiguous(), tag="bwd")          # [|R2|, Fout], compact
            gs_bound = (graph.t().inf_norm() * torch.linalg.vector_norm(gp, ord=float("inf")) * 1.0001) \
                if f32 else None
            grad_w = grad_in = None
            if need_w:
                grad_w = weight_grad_rows(input, grad_sup, rs.rows2_padded, None, ctx.x_bound, gs_bound,
                                          n_list=rs.n2) if (f32 and _gemm_scheme != "exact" and
                                                            (ctx.x_bound is not None or not gemm_needs_bounds())) else None
                if grad_w is None:
                    grad_w = _weight_grad(input.index_select(0, rs.rows2), grad_sup)
            if need_in:
                wt = weight.t().contiguous()
                y_max = torch.zeros(1, dtype=torch.float32, device=dev) if f32 else None
                part = gemm_xw256(grad_sup, wt, gs_bound, y_max)
                if part is None:
                    part, y_max = gemm_bf16(grad_sup, wt), None
                if part is None:
                    part, y_max = torch.mm(grad_sup, wt), None
                grad_in = RowGrad(rs.rows2, part, input.shape[0],
                                  meta={"padded": rs.rows2_padded, "absmax": y_max})
            return grad_in, grad_w, grad_bias, None, None, None, None, None
        if ctx.reassoc and not need_in:
            # ---- a first layer evaluated as (A·x)·W: mask on the compact rows, grad_W from the saved z
            meta = grad.meta or {}
            if meta.get("padded") is None:        # (rows of unknown structure: sorted-unique lists only)
                return None
            g = vals
            bound = meta.get("absmax")
            if ctx.relu:
                g = relu_dropout_backward(vals.contiguous(), out.index_select(0, rows), ctx.scale)
                bound = bound * ctx.scale if bound is not None else None
            grad_bias = grad_w = None
            if need_b and need_w and f32 and g.is_contiguous():      # (both from ONE pass over g: weight_grad_rows)
                both = weight_grad_rows(input, g, meta["padded"], None, ctx.z_bound, bound, n_list=rows.numel(),
                                        colsum_g=True)
                if both is not None:
                    grad_w, grad_bias = both[0], both[1].to(dt)
            if need_b and grad_bias is None:
                sums = backward_with_colsum(g) if g.is_contiguous() else None
                grad_bias = sums[1] if sums is not None else g.float().sum(0).to(dt)
            if need_w and grad_w is None:
                grad_w = weight_grad_rows(input, g, meta["padded"], None, ctx.z_bound, bound,
                                          n_list=rows.numel()) if (f32 and (ctx.z_bound is not None or not gemm_needs_bounds())) else None
                if grad_w is None:
                    grad_w = _weight_grad(input.index_select(0, rows), g)
            return None, grad_w, grad_bias, None, None, None, None, None
        return None

    @staticmethod
    def backward(ctx, grad_out):
        from .rowgrad import RowGrad
        if isinstance(grad_out, RowGrad):
            res = GraphConvFunction._backward_rows(ctx, grad_out)
            if res is not None:
                return res
            grad_out = grad_out.dense()
        input, weight = ctx.saved_tensors[:2]
        out = ctx.saved_tensors[2] if (ctx.relu or ctx.log_softmax) else None
        need_in, need_w, need_b = ctx.needs_input_grad[:3]
        graph_t = ctx.graph.t()
        n = graph_t.shape[0]
        # what follows reads the host-side count of non-zero gradient rows (a stream
        # synchronisation): large graphs only, never while capturing a hipGraph
        sync_ok = (_row_compaction and n >= MIN_ROWS and (need_in or need_w)
                   and not torch.cuda.is_current_stream_capturing())
        # ... and when it may, the fused backward pass does not even write the all-zero rows of
        # grad_pre: every consumer below reads the rows flagged in the bitmap only
        grad_pre, grad_bias, hint = _grad_pre_and_bias(grad_out, out, ctx.relu, ctx.scale, need_b,
                                                       ctx.log_softmax, skip_zero_rows=sync_ok)
        if not (need_in or need_w):
            return None, None, grad_bias, None, None, None, None, None
        if ctx.reassoc:          # (input is z = A·x here; x itself needs no gradient)
            z, grad_w = input, None
            if sync_ok and hint is not None:
                nz_rows = int(hint[1].item())
                if tuning.below(nz_rows, grad_pre.shape[0], tuning.SPARSE_GEMM_MAX_SHARE):   # the listed rows only
                    rows = torch.nonzero(unpack_row_flags(hint[0], grad_pre.shape[0])).squeeze(1)
                    lst = padded_row_list(rows)
                    grad_w = weight_grad_rows(z, grad_pre, lst, lst, ctx.z_bound, None, n_list=nz_rows)
                    if grad_w is None:
                        grad_w = _weight_grad(z.index_select(0, rows), grad_pre.index_select(0, rows))
                elif ctx.relu:       # rows of grad_pre whose bit is clear were not written
                    keep = unpack_row_flags(hint[0], grad_pre.shape[0])[:, None]
                    grad_pre = torch.where(keep, grad_pre, torch.zeros_like(grad_pre[:1]))
            if grad_w is None:
                grad_pre = grad_pre.contiguous()
                g_bound = None
                if ctx.z_bound is not None:      # |mask(grad_out)| <= scale * max|grad_out|
                    g_bound = known_absmax(grad_out)
                    g_bound = g_bound * ctx.scale if g_bound is not None else \
                        torch.linalg.vector_norm(grad_pre, ord=float("inf")).reshape(1)
                grad_w = _weight_grad(z, grad_pre, ctx.z_bound, g_bound)
            return None, grad_w, grad_bias, None, None, None, None, None
        c_flags = rows = None
        compact = sync_ok and hint is not None
        unwritten = compact and (ctx.relu or ctx.log_softmax)   # grad_pre: flagged rows only
        nz_rows = int(hint[1].item()) if compact else None
        if (compact and not need_in and tuning.below(nz_rows, grad_pre.shape[0], tuning.SPARSE_GEMM_MAX_SHARE)
                and input.shape[1] <= REASSOC_MAX_WIDTH_RATIO * grad_pre.shape[1]):
            # First layer (its input needs no gradient) under a row-sparse grad_pre:
            #     grad_W = inputᵀ · (Aᵀ · grad_pre) = (A · input)ᵀ · grad_pre,
            # and only the rows of A · input that meet a non-zero row of grad_pre take part: a
            # forward product restricted to those rows (c_select = the bitmap of grad_pre) and a
            # GEMM over them replace the transpose product and the full-height GEMM.  The
            # product here gathers rows of width Fin where the transpose product gathers width
            # Fout, and z is [n_rows, Fin]: it pays (and is bounded in memory) only while Fin is
            # not much wider than Fout — a 1433 -> 16 layer keeps the transpose product.
            z = spmm_csr(ctx.graph, input, tag="bwd", c_select=hint[0],
                         out=_maybe_poisoned((ctx.graph.shape[0], input.shape[1]), input.dtype,
                                             input.device))
            rows = torch.nonzero(unpack_row_flags(hint[0], grad_pre.shape[0])).squeeze(1)
            grad_w = _weight_grad(z.index_select(0, rows), grad_pre.index_select(0, rows))
            return None, grad_w, grad_bias, None, None, None, None, None
        if unwritten and not _hint_will_be_used(grad_pre, nz_rows):
            # the product below would gather every row: give the unwritten ones their zeros
            keep = unpack_row_flags(hint[0], grad_pre.shape[0])[:, None]
            grad_pre = torch.where(keep, grad_pre, torch.zeros_like(grad_pre[:1]))
        if compact and tuning.below(nz_rows, grad_pre.shape[0], tuning.SPARSE_FLAGS_MAX_SHARE):
            c_flags = torch.zeros(n, dtype=torch.uint8, device=grad_pre.device)
        # with the flags requested, all-zero rows of the product are not even written: the GEMMs
        # below read the flagged rows only
        grad_sup = spmm_csr(graph_t, grad_pre.contiguous(), tag="bwd", b_hint=hint, c_flags=c_flags,
                            skip_zero_rows=c_flags is not None,
                            out=_maybe_poisoned((n, grad_pre.shape[1]), grad_pre.dtype, grad_pre.device)
                            if c_flags is not None else None)
        if c_flags is not None:
            rows = torch.nonzero(c_flags).squeeze(1)
            if not tuning.below(rows.numel(), n, tuning.SPARSE_GEMM_MAX_SHARE):     # too dense to pay: real zeros
                rows = None
                grad_sup = torch.where(c_flags.bool()[:, None], grad_sup, torch.zeros_like(grad_sup[:1]))
        grad_in, grad_w = _dense_grads(input, weight, grad_sup, need_in, need_w, rows)
        return grad_in, grad_w, grad_bias, None, None, None, None, None


def spmm(adj, dense, bias=None):
    """Drop-in for `torch.spmm(adj, dense)` (+ optional fused bias) with autograd: a thin caller
    of the registered operator `torch.ops.pygcn_amd.spmm_csr` (pygcn_amd/ops.py)."""
    from .ops import sparse_mm
    return sparse_mm(adj, dense, bias)
