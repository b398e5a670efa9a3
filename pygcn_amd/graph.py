"""CSRGraph — the prepared adjacency handle of the MI355X GraphConvolution path.

Replaces the torch sparse COO tensor the reference builds once at load time
(`sparse_mx_to_torch_sparse_tensor`, reference pygcn/utils.py:407-414) and passes to every
`GraphConvolution.forward(input, adj)` (pygcn/layers.py:32).  The handle owns, in HBM:

  * CSR(A): rowptr (int32, or int64 when nnz >= 2^31), col (int32), val (fp32);
  * the static launch schedule for it (`struct gcn_csr_plan`, include/gcn_spmm.h), built once by
    the native planner from the row lengths;
  * lazily, the same for CSR(A^T), which the backward product A^T·grad needs
    (reference: autograd of `torch.spmm`, pygcn/train.py:157).  The reference (PyTorch) re-derives
    the transpose on every backward call; here it is built once and cached.

Conversions from torch / scipy layouts are one-off host+device plumbing done with torch ops.
"""
import collections
import ctypes
import weakref

import numpy as np
import torch

from . import _native

# CSR arrays -> the prepared handle built on them: the registered operator
# (`torch.ops.pygcn_amd.spmm_csr`, pygcn_amd/ops.py) receives plain tensors and finds the cached
# schedule / transpose here.  Keys are the arrays' device addresses, which stay unique for as long
# as the handle (which owns the arrays) lives; handles built for arrays nobody else holds a
# CSRGraph for are kept alive by a small LRU.
_BY_ARRAYS = weakref.WeakValueDictionary()
_RECENT = collections.OrderedDict()
_RECENT_MAX = 8


def _arrays_key(rowptr, col, val, shape):
    return (rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), rowptr.dtype, int(shape[0]),
            int(shape[1]), int(col.numel()))


def graph_for_arrays(rowptr, col, val, shape):
    """The CSRGraph built on exactly these device arrays (same storage), constructing and caching
    one if there is none: schedule and transpose are then built once per adjacency, not per call."""
    key = _arrays_key(rowptr, col, val, shape)
    g = _BY_ARRAYS.get(key)
    if g is None:
        g = CSRGraph(rowptr, col, val, shape)
        key = _arrays_key(g.rowptr, g.col, g.val, g.shape)   # (.contiguous() may have copied)
    _RECENT[key] = g
    _RECENT.move_to_end(key)
    while len(_RECENT) > _RECENT_MAX:
        _RECENT.popitem(last=False)
    return g


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"pygcn_amd: {what} must live on a HIP device (got {t.device}). The MI355X path has "
            "no CPU implementation; move the tensor with .cuda().")


class CSRGraph:
    """Sparse matrix A [n_rows, n_cols] in CSR on a HIP device, with its launch schedule."""

    def __init__(self, rowptr, col, val, shape, item_cost=0, long_thresh=0, validate=True):
        """`validate` (default on; one-off, a few device reductions) checks what the kernels
        assume — monotone row pointers ending at nnz and 0 <= col < n_cols — because an
        out-of-range index would be an out-of-bounds gather on the GPU."""
        for name, t in (("rowptr", rowptr), ("col", col), ("val", val)):
            _require_cuda(t, name)
        n_rows, n_cols = int(shape[0]), int(shape[1])
        if rowptr.dtype not in (torch.int32, torch.int64) or rowptr.numel() != n_rows + 1:
            raise RuntimeError("rowptr must be int32/int64 with n_rows+1 entries")
        if col.dtype != torch.int32 or val.dtype != torch.float32 or col.numel() != val.numel():
            raise RuntimeError("col must be int32 and val float32, same length")
        if validate:
            nnz = int(col.numel())
            ok = int(rowptr[0]) == 0 and int(rowptr[-1]) == nnz
            if ok and n_rows > 0:
                ok = bool((rowptr[1:] >= rowptr[:-1]).all())
            if ok and nnz > 0:
                ok = int(col.min()) >= 0 and int(col.max()) < n_cols
            if not ok:
                raise RuntimeError("invalid CSR: rowptr must rise from 0 to nnz and every column "
                                   f"index must lie in [0, {n_cols})")
        self.rowptr = rowptr.contiguous()
        self.col = col.contiguous()
        self.val = val.contiguous()
        self.shape = (n_rows, n_cols)
        self.nnz = int(col.numel())
        self.device = val.device
        self.item_cost = int(item_cost)
        self.long_thresh = int(long_thresh)
        self._plan = None
        self._keep = None
        self._plans_extra = {}        # chunk length -> (plan, arrays): plan(dtype) for bf16 storage
        self._t = None
        self._t_val_version = None
        self._inf_norm = None
        _BY_ARRAYS[_arrays_key(self.rowptr, self.col, self.val, self.shape)] = self

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_coo(cls, row, col, val, shape, device=None, coalesced=False, reduce="sum", **kw):
        """COO triplets (any order, duplicates allowed — torch.spmm sums them) -> CSR, on the
        device by the native ingest primitive `gcn_coo_to_csr_device` (stable radix sort by
        (row, col), duplicates reduced in storage order: deterministic, no atomics).
        `reduce="max"` keeps the largest of duplicate entries instead of their sum (the
        reference's symmetrization, see from_edge_list).  `coalesced` is accepted for
        compatibility; sorted unique input simply sorts to itself.
        Limits: HIP tensors only (no host path), and fewer than 2³² − 1 triplets per call (the
        sort carries 32-bit storage positions) — both raise a RuntimeError here, before any launch.
        Larger inputs: convert in row ranges, or hand CSR arrays to the constructor."""
        row = torch.as_tensor(row)
        device = torch.device(device) if device is not None else row.device
        row = row.to(device=device, dtype=torch.int64).contiguous()
        col = torch.as_tensor(col).to(device=device, dtype=torch.int64).contiguous()
        val = torch.as_tensor(val).to(device=device, dtype=torch.float32).contiguous()
        _require_cuda(row, "COO indices")
        n_rows, n_cols = int(shape[0]), int(shape[1])
        nnz = int(row.numel())
        if col.numel() != nnz or val.numel() != nnz:
            raise RuntimeError("COO row / col / val must have the same length")
        if nnz >= 2 ** 32 - 1:
            raise RuntimeError(f"from_coo: {nnz} triplets; the device COO->CSR conversion takes fewer than "
                               "2^32 - 1 per call (convert in row ranges, or pass CSR arrays)")
        if nnz and (int(row.max()) >= n_rows or int(col.max()) >= n_cols or
                    int(row.min()) < 0 or int(col.min()) < 0):
            raise RuntimeError("COO index out of range for the given shape")
        L = _native.lib()
        is64 = int(nnz >= 2 ** 31 - 1)
        rowptr = torch.empty(n_rows + 1, dtype=torch.int64 if is64 else torch.int32, device=device)
        col_out = torch.empty(nnz, dtype=torch.int32, device=device)
        val_out = torch.empty(nnz, dtype=torch.float32, device=device)
        nnz_out = torch.zeros(1, dtype=torch.int64, device=device)
        ws_bytes = L.gcn_coo_to_csr_workspace_bytes(n_rows, n_cols, nnz)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            rc = L.gcn_coo_to_csr_device(
                row.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, n_rows, n_cols,
                {"sum": _native.GCN_REDUCE_SUM, "max": _native.GCN_REDUCE_MAX}[reduce],
                rowptr.data_ptr(), is64, col_out.data_ptr(), val_out.data_ptr(), nnz_out.data_ptr(),
                ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream)
        _native.check(rc, "gcn_coo_to_csr_device")
        k = int(nnz_out.item())                     # one 8-byte read: the number of distinct entries
        del ws
        if k < nnz:                                 # give the unused capacity back
            col_out, val_out = col_out[:k].clone(), val_out[:k].clone()
        return cls(rowptr, col_out, val_out, (n_rows, n_cols), validate=False, **kw)

    @classmethod
    def from_edge_list(cls, edges, n, device="cuda", symmetrize=True, self_loops=True,
                       normalize=True, **kw):
        """The reference's adjacency recipe (pygcn/utils.py:360-368, kept there as a comment) on
        the device, from an [E, 2] array of (source, target) vertex ids:
            adj = coo_matrix(ones, (src, dst))                  duplicates SUMMED        :360-362
            adj = adj + adj.T*(adj.T > adj) - adj*(adj.T > adj)  = max(adj, adj.T)       :365
            adj = normalize(adj + I)                            D^-1 (A + I)            :368
        Every step is the native COO->CSR reduction (sum / max) or the native row normalisation.
        Symmetrizing doubles the triplet count of the intermediate conversion: with `symmetrize` the
        input may hold fewer than 2³¹ pairs (from_coo's 2³² − 1 limit), checked before any launch."""
        dev = torch.device(device)
        e = torch.as_tensor(edges).to(device=dev, dtype=torch.int64)
        if symmetrize and 2 * e.shape[0] + (n if self_loops else 0) >= 2 ** 32 - 1:
            raise RuntimeError(f"from_edge_list: {e.shape[0]} pairs; symmetrization doubles them past the "
                               "2^32 - 1 triplets one device conversion takes")
        g = cls.from_coo(e[:, 0], e[:, 1], torch.ones(e.shape[0], device=dev), (n, n), **kw)
        if symmetrize:
            r, c, v = g.coo()
            g = cls.from_coo(torch.cat([r, c]), torch.cat([c, r]), torch.cat([v, v]), (n, n),
                             reduce="max", **kw)
        if self_loops:
            r, c, v = g.coo()
            d = torch.arange(n, device=dev, dtype=torch.int64)
            g = cls.from_coo(torch.cat([r, d]), torch.cat([c, d]),
                             torch.cat([v, torch.ones(n, device=dev)]), (n, n), **kw)
        return g.row_normalize_() if normalize else g

    def coo(self):
        """(row int64, col int64, val) of the stored entries, in CSR order."""
        deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
        row = torch.repeat_interleave(torch.arange(self.shape[0], device=self.device), deg)
        return row, self.col.to(torch.int64), self.val

    @classmethod
    def from_torch(cls, adj, device=None, **kw):
        """torch sparse COO (the reference's layout, utils.py:414) or sparse CSR tensor."""
        device = torch.device(device) if device is not None else adj.device
        if adj.layout == torch.sparse_coo:
            idx, val = adj._indices(), adj._values()
            return cls.from_coo(idx[0], idx[1], val, adj.shape, device=device,
                                coalesced=False, **kw)
        if adj.layout == torch.sparse_csr:
            rowptr = adj.crow_indices().to(device)
            if adj._nnz() < 2 ** 31 - 1:
                rowptr = rowptr.to(torch.int32)
            return cls(rowptr, adj.col_indices().to(device=device, dtype=torch.int32),
                       adj.values().to(device=device, dtype=torch.float32), adj.shape, **kw)
        raise RuntimeError(f"unsupported adjacency layout {adj.layout}")

    @classmethod
    def from_scipy(cls, m, device="cuda", **kw):
        import scipy.sparse as sp
        m = sp.csr_matrix(m)
        m.sum_duplicates()
        m.sort_indices()
        rp_dtype = np.int32 if m.nnz < 2 ** 31 - 1 else np.int64
        return cls(torch.from_numpy(m.indptr.astype(rp_dtype)).to(device),
                   torch.from_numpy(m.indices.astype(np.int32)).to(device),
                   torch.from_numpy(m.data.astype(np.float32)).to(device), m.shape, **kw)

    # ------------------------------------------------------------------ schedule
    def plan(self, dtype=None):
        """`struct gcn_csr_plan` for this matrix (built once, cached) by the DEVICE planner
        (`gcn_plan_count_device` / `gcn_plan_fill_device`): the row pointer never leaves HBM; the
        only host transfer is the 24-byte read of (n_items, n_chunks, n_long).
        `dtype`: storage type of the dense operand the plan will be used with.  A graph built
        without an explicit `long_thresh` chunks its long rows at the C-ABI default for fp32
        storage and at tuning.LONG_THRESH_BF16 for bf16 storage (a second cached schedule)."""
        if self.long_thresh <= 0 and dtype == torch.bfloat16:
            from . import tuning
            return self._plan_for(tuning.LONG_THRESH_BF16)
        if self._plan is not None:
            return self._plan
        if self._keep is None:
            self._keep = self._plan_arrays_device(self.long_thresh)
        self._plan = self._plan_struct(self._keep, self.long_thresh if self.long_thresh > 0
                                       else _native.GCN_DEFAULT_LONG_THRESH)
        return self._plan

    def _plan_for(self, thresh):
        """The schedule at another chunk length (cached beside the default one)."""
        hit = self._plans_extra.get(thresh)
        if hit is None:
            keep = self._plan_arrays_device(thresh)
            hit = self._plans_extra[thresh] = (self._plan_struct(keep, thresh), keep)
        return hit[0]

    def _plan_struct(self, keep, thresh):
        ni, nc, nl = keep["n_items"], keep["n_chunks"], keep["n_long"]
        p = _native.GcnCsrPlan()
        p.n_rows, p.n_cols, p.nnz = self.shape[0], self.shape[1], self.nnz
        p.rowptr, p.rowptr_is64 = self.rowptr.data_ptr(), int(self.rowptr.dtype == torch.int64)
        p.long_thresh = thresh
        p.col, p.val = self.col.data_ptr(), self.val.data_ptr()
        p.n_items, p.items = ni, keep["items"].data_ptr()
        p.n_chunks, p.chunk_row, p.chunk_e0 = nc, keep["chunk_row"].data_ptr(), \
            keep["chunk_e0"].data_ptr()
        p.n_long, p.long_row, p.long_chunk0 = nl, keep["long_row"].data_ptr(), \
            keep["long_chunk0"].data_ptr()
        return p

    def _plan_arrays_device(self, long_thresh):
        L = _native.lib()
        dev, n_rows = self.device, self.shape[0]
        is64 = int(self.rowptr.dtype == torch.int64)
        ws_bytes = L.gcn_plan_device_workspace_bytes(n_rows)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        counts = torch.zeros(3, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            _native.check(L.gcn_plan_count_device(self.rowptr.data_ptr(), is64, n_rows,
                                                  self.item_cost, long_thresh, ws.data_ptr(),
                                                  ws_bytes, counts.data_ptr(), stream),
                          "gcn_plan_count_device")
            ni, nc, nl = (int(v) for v in counts.tolist())
            keep = {"items": torch.empty(max(2 * ni, 1), dtype=torch.int32, device=dev),
                    "chunk_row": torch.empty(max(nc, 1), dtype=torch.int32, device=dev),
                    "chunk_e0": torch.empty(max(nc, 1), dtype=torch.int64, device=dev),
                    "long_row": torch.empty(max(nl, 1), dtype=torch.int32, device=dev),
                    "long_chunk0": torch.empty(nl + 1, dtype=torch.int32, device=dev),
                    "n_items": ni, "n_chunks": nc, "n_long": nl}
            _native.check(L.gcn_plan_fill_device(
                self.rowptr.data_ptr(), is64, n_rows, long_thresh, ws.data_ptr(), ws_bytes,
                keep["items"].data_ptr(), ni, keep["chunk_row"].data_ptr(),
                keep["chunk_e0"].data_ptr(), nc, keep["long_row"].data_ptr(),
                keep["long_chunk0"].data_ptr(), nl, stream), "gcn_plan_fill_device")
        return keep

    def plan_arrays_host_planner(self):
        """The same schedule from the HOST planner (`gcn_plan_{count,fill}_host`), as numpy arrays
        — kept for callers that hold the row pointer on the host, and as the cross-check of the
        device planner (tests/test_ingest_gpu.py: array-for-array equality)."""
        return host_schedule(self.rowptr.cpu().numpy(), self.shape[0], self.item_cost, self.long_thresh)

    def schedule_stats(self, dtype=None):
        p = self.plan(dtype)
        return {"n_items": int(p.n_items), "n_chunks": int(p.n_chunks), "n_long": int(p.n_long),
                "long_thresh": int(p.long_thresh)}

    # ------------------------------------------------------------------ transpose (for backward)
    def t(self):
        """CSR(A^T), built once on the device by the native ingest kernel
        (`gcn_csr_transpose_device`: stable radix sort by column).  Within each row of A^T the
        entries are in increasing source-row order, so backward sums are deterministic."""
        if self._t is not None and self._t_val_version != self.val._version:
            self._t = None        # the values were edited in place since the transpose was built
        if self._t is None:
            n_rows, n_cols = self.shape
            dev = self.device
            L = _native.lib()
            rowptr_t = torch.empty(n_cols + 1, dtype=self.rowptr.dtype, device=dev)
            col_t = torch.empty(self.nnz, dtype=torch.int32, device=dev)
            val_t = torch.empty(self.nnz, dtype=torch.float32, device=dev)
            ws_bytes = L.gcn_csr_transpose_workspace_bytes(n_rows, n_cols, self.nnz)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = L.gcn_csr_transpose_device(
                    self.rowptr.data_ptr(), int(self.rowptr.dtype == torch.int64),
                    self.col.data_ptr(), self.val.data_ptr(), n_rows, n_cols, self.nnz,
                    rowptr_t.data_ptr(), col_t.data_ptr(), val_t.data_ptr(), ws.data_ptr(),
                    ws_bytes, torch.cuda.current_stream().cuda_stream)
            _native.check(rc, "gcn_csr_transpose_device")
            del ws
            g = CSRGraph(rowptr_t, col_t, val_t, (n_cols, n_rows), item_cost=self.item_cost,
                         long_thresh=self.long_thresh, validate=False)
            g._t, g._t_val_version = self, val_t._version
            self._t, self._t_val_version = g, self.val._version
        return self._t

    def row_normalize_(self):
        """In place D^-1 · A on the device (`gcn_row_normalize_device`): the reference's
        `normalize(mx)` (pygcn/utils.py:390-397); rows that sum to 0 stay 0.  Invalidates the
        cached transpose."""
        with torch.cuda.device(self.device):
            rc = _native.lib().gcn_row_normalize_device(
                self.rowptr.data_ptr(), int(self.rowptr.dtype == torch.int64),
                self.val.data_ptr(), self.shape[0], torch.cuda.current_stream().cuda_stream)
        _native.check(rc, "gcn_row_normalize_device")
        self._t = None
        return self

    def inf_norm(self):
        """‖A‖∞ = the largest absolute row sum, as a DEVICE float tensor [1] (computed once): with
        it, max|A·B| <= ‖A‖∞ · max|B| bounds a product's output without a pass over it."""
        if self._inf_norm is None or self._inf_norm[0] != self.val._version:
            row = self.coo()[0]
            sums = torch.zeros(self.shape[0], dtype=torch.float32, device=self.device).index_add_(
                0, row, self.val.abs())
            top = sums.max() if sums.numel() else sums.sum()
            self._inf_norm = (self.val._version, (top * 1.00001).reshape(1))
        return self._inf_norm[1]

    # ------------------------------------------------------------------ binary cache file
    _PLAN_KEYS = ("items", "chunk_row", "chunk_e0", "long_row", "long_chunk0")

    def save(self, path, with_transpose=True):
        """Write CSR(Â), its schedule and (optionally) CSR(Âᵀ) + schedule to a versioned binary
        cache file (pygcn_amd/cache.py).  Returns the file size in bytes."""
        from . import cache
        arrays = {}

        def put(prefix, g):
            g.plan()
            k = g._keep
            arrays[prefix + "rowptr"] = g.rowptr.cpu().numpy()
            arrays[prefix + "col"] = g.col.cpu().numpy()
            arrays[prefix + "val"] = g.val.cpu().numpy()
            arrays[prefix + "items"] = k["items"][:2 * k["n_items"]].cpu().numpy()
            arrays[prefix + "chunk_row"] = k["chunk_row"][:k["n_chunks"]].cpu().numpy()
            arrays[prefix + "chunk_e0"] = k["chunk_e0"][:k["n_chunks"]].cpu().numpy()
            arrays[prefix + "long_row"] = k["long_row"][:k["n_long"]].cpu().numpy()
            arrays[prefix + "long_chunk0"] = k["long_chunk0"].cpu().numpy()
        put("a.", self)
        if with_transpose:
            put("t.", self.t())
        meta = {"n_rows": self.shape[0], "n_cols": self.shape[1], "nnz": self.nnz,
                "item_cost": self.item_cost, "long_thresh": self.long_thresh,
                "has_transpose": bool(with_transpose), "abi_version": _native.GCN_ABI_VERSION}
        return cache.write_file(path, meta, arrays)

    @classmethod
    def load(cls, path, device="cuda", verify=True):
        """Re-open a cache file written by save(): arrays go straight to the device; neither the
        planner nor the transpose runs again."""
        from . import cache
        meta, arr = cache.read_file(path, verify=verify)
        dev = torch.device(device)
        # The schedule arrays go to the kernels as they are: a file written by a build with another
        # ABI (item layout, chunking rule) would give wrong sums or out-of-range reads — refuse it.
        if meta.get("abi_version") != _native.GCN_ABI_VERSION:
            raise cache.CacheFormatError(
                f"{path}: written by ABI version {meta.get('abi_version')}, this build is "
                f"{_native.GCN_ABI_VERSION}: rebuild the cache (CSRGraph.save)")

        def dev_t(a):
            import warnings
            with warnings.catch_warnings():       # (read-only memory map: it is only copied from)
                warnings.simplefilter("ignore", UserWarning)
                return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

        def check_schedule(prefix, n_rows):
            """The stored schedule must be THE schedule of the stored row pointer: structural
            checks always; with verify=True the host planner re-derives it (0.2 s at 10^7 rows)
            and every array must match."""
            rp = arr[prefix + "rowptr"]
            sched = {k: arr[prefix + k] for k in cls._PLAN_KEYS}
            ni, nc, nl = sched["items"].size // 2, sched["chunk_row"].size, sched["long_row"].size
            bad = None
            if rp.size != n_rows + 1 or sched["items"].size % 2:
                bad = "row pointer / item array size"
            elif sched["chunk_e0"].size != nc or sched["long_chunk0"].size != nl + 1:
                bad = "chunk / long-row array sizes"
            elif nl and (int(sched["long_chunk0"][0]) != 0 or int(sched["long_chunk0"][-1]) != nc):
                bad = "long_chunk0 does not span the chunks"
            elif not nl and nc:
                bad = "chunks without long rows"
            elif ni and (int(sched["items"].min()) < 0 or int(sched["items"].max()) > n_rows):
                bad = "item row range"
            elif nc and (int(sched["chunk_row"].min()) < 0 or int(sched["chunk_row"].max()) >= n_rows
                         or int(sched["chunk_e0"].min()) < 0 or int(sched["chunk_e0"].max()) >= int(rp[-1])):
                bad = "chunk row / entry range"
            elif nl and (int(sched["long_row"].min()) < 0 or int(sched["long_row"].max()) >= n_rows):
                bad = "long row range"
            if bad is None and verify:
                want = host_schedule(rp, n_rows, meta["item_cost"], meta["long_thresh"])
                for k in cls._PLAN_KEYS:
                    if not np.array_equal(np.asarray(sched[k]), want[k]):
                        bad = f"`{k}` is not the schedule of the stored row pointer"
                        break
            if bad is not None:
                raise cache.CacheFormatError(f"{path}: inconsistent schedule ({prefix}{bad})")

        def get(prefix, shape):
            check_schedule(prefix, shape[0])
            g = cls(dev_t(arr[prefix + "rowptr"]), dev_t(arr[prefix + "col"]),
                    dev_t(arr[prefix + "val"]), shape, item_cost=meta["item_cost"],
                    long_thresh=meta["long_thresh"], validate=verify)
            ni = arr[prefix + "items"].size // 2
            nc, nl = arr[prefix + "chunk_row"].size, arr[prefix + "long_row"].size

            def padded(name, dtype):      # (plan arrays are never empty tensors: 1 spare entry)
                a = arr[prefix + name]
                return dev_t(a) if a.size else torch.empty(1, dtype=dtype, device=dev)
            g._keep = {"items": padded("items", torch.int32),
                       "chunk_row": padded("chunk_row", torch.int32),
                       "chunk_e0": padded("chunk_e0", torch.int64),
                       "long_row": padded("long_row", torch.int32),
                       "long_chunk0": dev_t(arr[prefix + "long_chunk0"]),
                       "n_items": ni, "n_chunks": nc, "n_long": nl}
            return g
        g = get("a.", (meta["n_rows"], meta["n_cols"]))
        if g.nnz != meta["nnz"]:
            raise cache.CacheFormatError(f"{path}: nnz in header and arrays differ")
        if meta.get("has_transpose"):
            gt = get("t.", (meta["n_cols"], meta["n_rows"]))
            g._t, g._t_val_version = gt, g.val._version
            gt._t, gt._t_val_version = g, gt.val._version
        return g

    # ------------------------------------------------------------------ misc
    def to_torch_csr(self):
        return torch.sparse_csr_tensor(self.rowptr.to(torch.int64), self.col.to(torch.int64),
                                       self.val, size=self.shape)

    def __repr__(self):
        return f"CSRGraph(shape={self.shape}, nnz={self.nnz}, device={self.device})"


def host_schedule(rp_host, n_rows, item_cost, long_thresh):
    """Launch schedule of a HOST row pointer (int32 / int64 numpy array) from the host planner
    `gcn_plan_{count,fill}_host`: the arrays of struct gcn_csr_plan as numpy arrays."""
    L = _native.lib()
    rp_host = np.ascontiguousarray(rp_host)
    if rp_host.dtype not in (np.int32, np.int64) or rp_host.size != n_rows + 1:
        raise RuntimeError("host_schedule: rowptr must be int32 / int64 with n_rows + 1 entries")
    is64 = int(rp_host.dtype == np.int64)
    ni, nc, nl = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    _native.check(L.gcn_plan_count_host(rp_host.ctypes.data, is64, n_rows, item_cost,
                                        long_thresh, ctypes.byref(ni), ctypes.byref(nc),
                                        ctypes.byref(nl)), "gcn_plan_count_host")
    ni, nc, nl = ni.value, nc.value, nl.value
    items = np.empty(max(2 * ni, 1), np.int32)
    chunk_row = np.empty(max(nc, 1), np.int32)
    chunk_e0 = np.empty(max(nc, 1), np.int64)
    long_row = np.empty(max(nl, 1), np.int32)
    long_chunk0 = np.empty(nl + 1, np.int32)
    _native.check(L.gcn_plan_fill_host(rp_host.ctypes.data, is64, n_rows, item_cost,
                                       long_thresh, items.ctypes.data, ni,
                                       chunk_row.ctypes.data, chunk_e0.ctypes.data, nc,
                                       long_row.ctypes.data, long_chunk0.ctypes.data, nl),
                  "gcn_plan_fill_host")
    return {"items": items[:2 * ni], "chunk_row": chunk_row[:nc], "chunk_e0": chunk_e0[:nc],
            "long_row": long_row[:nl], "long_chunk0": long_chunk0,
            "n_items": ni, "n_chunks": nc, "n_long": nl}


_ADJ_CACHE_ATTR = "_pygcn_amd_graph"


def as_graph(adj):
    """Accept what the reference's layer accepts for `adj` and return the prepared handle.
    torch sparse COO/CSR tensors are converted ONCE and the handle is cached on the tensor."""
    if isinstance(adj, CSRGraph):
        return adj
    if isinstance(adj, torch.Tensor) and adj.layout in (torch.sparse_coo, torch.sparse_csr):
        g = getattr(adj, _ADJ_CACHE_ATTR, None)
        if g is None:
            _require_cuda(adj, "adj")
            g = CSRGraph.from_torch(adj)
            try:
                setattr(adj, _ADJ_CACHE_ATTR, g)
            except AttributeError:
                pass
        return g
    raise RuntimeError(f"adj must be a CSRGraph or a torch sparse tensor, got {type(adj)}")
