"""Row-block sharding of the GraphConvolution path over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference has no multi-device code at all; this is the scaling design for the
same layer (reference pygcn/layers.py:32-38):

  * Â is cut into P contiguous row blocks with (almost) equal numbers of stored entries; rank r
    owns rows [b_r, b_r+1) of Â, of Âᵀ, of X, of every activation and gradient.  Weights are
    replicated.
  * forward   support_r = X_r · W  ->  ALL-GATHER support  ->  out_r = Â[b_r:b_r+1, :] · support
  * backward  ALL-GATHER grad_out  ->  grad_support_r = Âᵀ[b_r:b_r+1, :] · grad_out, then the
    local GEMMs, then ALL-REDUCE of grad_W / grad_b (256 KiB at F=256).
  * The all-gather lands in a padded layout [P * max_rows, F] (every rank contributes max_rows
    rows, the tail unused) and the column indices of the local blocks are remapped ONCE to that
    layout, so the gathered buffer is consumed in place: no compaction copy, no all-gather-v.
  * No float atomics, no reduce-scatter of N×F partial sums: each output row is produced by
    exactly one rank.
  * Exchange modes.  "allgather" moves every row to every rank ((P-1)/P·N·F·s bytes in per rank),
    as a direct mesh exchange (one grouped point-to-point round, every block straight to every
    peer) rather than a ring collective.
    "halo" (default) moves a row only to the ranks whose block references it: at setup every rank
    sends each owner the sorted list of that owner's rows it needs (one grouped P2P round); per
    product the owner packs those rows (index_select) and one grouped isend/irecv round lands them
    in a halo buffer [n_halo, F]; the local column ids were remapped once to [own rows | halo
    rows], and the kernels read that two-block operand in place (no copy of the own rows).  On R-MAT graphs more than half of the rows are referenced by
    no remote rank at all (self-loop-only vertices), and xGMI is point-to-point, so sending only
    what is needed, directly owner -> consumer, is the right shape for it.

  * Constant input (the feature matrix X of the first layer, `ShardedGraph.register_constant_input`):
    its halo rows are exchanged ONCE and kept — each rank then holds the feature rows its block
    references, like it holds its block of Â — and the first layer needs no exchange at all:
        forward   z_r = Â_r · [X_r ; X_halo];  out_r = epilogue(z_r · W + b)     (one GEMM over the
                                                           rank's own rows with the layer's epilogue in
                                                           its store — 256 -> 256 fp32 / bf16 128-wide;
                                                           other widths: Â_r · ([X_r ; X_halo] · W))
        backward  grad_W partial = z_rᵀ · grad_pre_r       (the z_r of the forward pass; summed by the
                                                           gradient all-reduce; X needs no gradient)
    so a 2-layer GCN epoch has 2 exchanges (layer 2 forward / backward) instead of 4.
  * Pipelined by source block: a dense halo exchange posts its transfers, runs the product over
    the entries that reference the rank's own rows while they fly, then the product over the halo
    entries (`ShardedGraph.split_block`, `product`).
  * Backward exchanges are row-sparse (`HaloExchange.exchange_sparse`): the operand is the masked
    gradient, non-zero on the labelled vertices' rows only; only those rows travel (with their
    positions in the peer's request list), and their flags + the rank's own become the operand
    hint of the local transpose product.  With the loss rows named (`ShardedGCN(x, sg, rows=...)`)
    the whole step of a rank is one autograd node and the gradient halo is static
    (pygcn_amd/sharded_fused.py).

The local product is `pygcn_amd.spmm.spmm_csr` (HIP) and the local backward pass
`pygcn_amd.spmm._grad_pre_and_bias` (HIP).  `graph_factory` / `spmm_fn` / `bwd_fn` exist so the
partition / exchange logic can be exercised on CPU with gloo in tests/, where tests/ (never this
package) supplies CPU stand-ins (the oracle) for them.
"""
import weakref

import torch
import torch.distributed as dist

from . import tuning
from .graph import CSRGraph
from .spmm import (_dense_forward, _grad_pre_and_bias, _weight_grad, dropout_scale, pack_row_flags, rows_pack,
                   rows_unpack, spmm_csr, unpack_row_flags)


def partition_rows(rowptr, world):
    """Contiguous row-block boundaries (P+1 ints) balancing stored entries per block."""
    n = rowptr.numel() - 1
    nnz = int(rowptr[-1])
    targets = torch.tensor([nnz * r // world for r in range(1, world)], dtype=rowptr.dtype,
                           device=rowptr.device)
    cuts = torch.searchsorted(rowptr, targets, right=False).clamp_(0, n).tolist()
    bounds = [0] + cuts + [n]
    for i in range(1, len(bounds)):          # monotone even for degenerate inputs
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def remap_columns(col, bounds, max_rows):
    """Global column id -> row of the padded gathered buffer [P*max_rows, F]."""
    b = torch.tensor(bounds, dtype=torch.int64, device=col.device)
    owner = torch.searchsorted(b, col.to(torch.int64), right=True) - 1
    owner.clamp_(0, len(bounds) - 2)
    return (col.to(torch.int64) - b[owner] + owner * max_rows).to(torch.int32)


def row_block(rowptr, col, val, r0, r1):
    """Rows [r0, r1) of a CSR matrix as (rebased rowptr, col, val)."""
    e0, e1 = int(rowptr[r0]), int(rowptr[r1])
    return (rowptr[r0:r1 + 1] - rowptr[r0]), col[e0:e1], val[e0:e1]


def transpose_row_block(rowptr, col, val, n_rows, r0, r1):
    """Rows [r0, r1) of Aᵀ (= columns [r0, r1) of A) as CSR with global source-row ids as columns;
    entries of a row in increasing source-row order (deterministic sums)."""
    dev = col.device
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    src = torch.repeat_interleave(torch.arange(n_rows, device=dev, dtype=torch.int32), deg)
    keep = (col >= r0) & (col < r1)
    c, s, v = col[keep].to(torch.int64) - r0, src[keep], val[keep]
    del src, keep
    _, perm = torch.sort(c, stable=True)
    counts = torch.bincount(c, minlength=r1 - r0) if c.numel() else torch.zeros(
        r1 - r0, dtype=torch.int64, device=dev)
    rp = torch.zeros(r1 - r0 + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rp[1:])
    return rp.to(rowptr.dtype), s[perm], v[perm]


def _p2p_begin(sends, recvs, group):
    """Post one grouped round of point-to-point transfers: sends = [(tensor, peer)], recvs
    likewise; returns the pending work handles.  Zero-length transfers are skipped on both sides
    (counts are known everywhere).  On RCCL the transfers run on the communicator's own stream,
    ordered after what the current stream has enqueued so far: kernels launched on the current
    stream between begin and end overlap them."""
    ops = [dist.P2POp(dist.isend, t, peer, group) for t, peer in sends if t.numel()] + \
          [dist.P2POp(dist.irecv, t, peer, group) for t, peer in recvs if t.numel()]
    return dist.batch_isend_irecv(ops) if ops else []


def _p2p_end(pending):
    """Make the current stream (the host, with gloo) wait for the posted transfers."""
    for w in pending:
        w.wait()


def _p2p_round(sends, recvs, group):
    """One grouped round of point-to-point transfers, complete on return."""
    _p2p_end(_p2p_begin(sends, recvs, group))


def _owner_of(ids, bounds_t, world):
    """Rank that owns each global row id (bounds may repeat: empty blocks own nothing)."""
    return (torch.searchsorted(bounds_t, ids, right=True) - 1).clamp_(0, world - 1)


def transpose_block_by_exchange(a_block, bounds, rank, world, group=None):
    """Rows [b_r, b_r+1) of Aᵀ from the row blocks of A held by the ranks, WITHOUT any rank ever
    holding the whole matrix: every rank buckets the stored entries of its block by the owner of
    their column and sends (column, global row, value) triplets straight to that owner (one
    grouped point-to-point round); the owner sorts what it received by (column, row).  Within each
    row of Aᵀ the entries come out in increasing source-row order — the order of
    `transpose_row_block` and of the single-GPU `gcn_csr_transpose_device`, so backward sums are
    deterministic and independent of the number of ranks.  Collective; O(nnz / P) memory."""
    rowptr, col, val = a_block
    dev = col.device
    n_global, r0, r1 = int(bounds[-1]), bounds[rank], bounds[rank + 1]
    b = torch.tensor(bounds, dtype=torch.int64, device=dev)
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    grow = torch.repeat_interleave(torch.arange(r0, r1, device=dev, dtype=torch.int64), deg)
    key = col.to(torch.int64) * n_global + grow              # sorts by (column, source row)
    key, perm = torch.sort(key)
    v = val[perm]
    del perm, grow
    cut = torch.searchsorted(key, b * n_global)               # segment of each owner
    mine = (cut[1:] - cut[:-1]).contiguous()                  # entries I hold for each owner
    M = torch.empty(world * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(M, mine, group=group)
    M = M.view(world, world).tolist()                         # M[s][r]: s holds for r
    cut = cut.tolist()
    rk = {s: torch.empty(M[s][rank], dtype=torch.int64, device=dev) for s in range(world) if s != rank}
    rv = {s: torch.empty(M[s][rank], dtype=torch.float32, device=dev) for s in range(world) if s != rank}
    sends, recvs = [], []
    for k in range(1, world):
        p = (rank + k) % world
        sends += [(key[cut[p]:cut[p + 1]], p), (v[cut[p]:cut[p + 1]], p)]
    for k in range(1, world):
        q = (rank - k) % world
        recvs += [(rk[q], q), (rv[q], q)]
    _p2p_round(sends, recvs, group)
    keys = torch.cat([key[cut[rank]:cut[rank + 1]] if s == rank else rk[s] for s in range(world)])
    vals = torch.cat([v[cut[rank]:cut[rank + 1]] if s == rank else rv[s] for s in range(world)])
    del key, v, rk, rv
    keys, perm = torch.sort(keys)
    vals = vals[perm]
    del perm
    c = keys // n_global                                       # my rows of Aᵀ (global ids)
    src = (keys - c * n_global).to(torch.int32)
    counts = torch.bincount(c - r0, minlength=r1 - r0) if c.numel() else torch.zeros(
        r1 - r0, dtype=torch.int64, device=dev)
    rp = torch.zeros(r1 - r0 + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rp[1:])
    return rp.to(rowptr.dtype), src, vals


def pack_bits(mask):
    """bool [m, F] (F a multiple of 32) -> int32 [m, F / 32]: bit j of word w = mask[:, 32w + j]."""
    m, F = mask.shape
    w = (mask.view(m, F // 32, 32).to(torch.int64) << torch.arange(32, device=mask.device)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def unpack_bits(bits, F):
    """Inverse of pack_bits: int32 [m, F / 32] -> bool [m, F]."""
    sh = torch.arange(32, device=bits.device, dtype=torch.int32)
    return ((bits[:, :, None] >> sh) & 1).bool().reshape(bits.shape[0], F)


def redistribute_rows(key, n, bounds, rank, world, group=None):
    """Sorted global entry keys (row·n + col) held by the ranks for ANY ascending contiguous row
    ranges -> the keys of this rank's final block [bounds[rank], bounds[rank+1]), by one grouped
    point-to-point round: every rank cuts its sorted keys at the final boundaries and sends each
    range to its owner; ranges arrive in source order, which is key order (the held ranges
    ascend with the rank).  Returns (keys of the final block, entries received from others)."""
    dev = key.device
    b = torch.tensor(bounds, dtype=torch.int64, device=dev)
    cut = torch.searchsorted(key, b * n)
    mine = (cut[1:] - cut[:-1]).contiguous()
    M = torch.empty(world * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(M, mine, group=group)
    M = M.view(world, world).tolist()                      # M[s][r]: s holds for r
    cut = cut.tolist()
    got = {s: torch.empty(M[s][rank], dtype=torch.int64, device=dev) for s in range(world) if s != rank}
    sends, recvs = [], []
    for k in range(1, world):
        p = (rank + k) % world
        sends.append((key[cut[p]:cut[p + 1]], p))
    for k in range(1, world):
        q = (rank - k) % world
        recvs.append((got[q], q))
    _p2p_round(sends, recvs, group)
    out = torch.cat([key[cut[rank]:cut[rank + 1]] if s == rank else got[s] for s in range(world)])
    return out, sum(int(t.numel()) for t in got.values())


class HaloExchange:
    """Which rows of every other rank's block this rank's CSR block references, and the send
    lists the other ranks asked of this rank.  Built once per (block, process group)."""

    def __init__(self, col_global, bounds, rank, world, group=None):
        dev = col_global.device
        self.rank, self.world, self.group = rank, world, group
        b = torch.tensor(bounds, dtype=torch.int64, device=dev)
        self.n_local = bounds[rank + 1] - bounds[rank]
        needed = torch.unique(col_global.to(torch.int64))            # sorted global row ids
        pos = torch.searchsorted(needed, b)                           # split by owner
        pos_l = pos.tolist()
        need = [(needed[pos_l[r]:pos_l[r + 1]] - bounds[r]).to(torch.int32) for r in range(world)]
        recv_counts = [int(need[r].numel()) if r != rank else 0 for r in range(world)]
        # counts matrix M[s][r] = rows of r that s needs
        mine = torch.tensor(recv_counts, dtype=torch.int64, device=dev)
        M = torch.empty(world * world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(M, mine, group=group)
        M = M.view(world, world).tolist()
        self.recv_counts = recv_counts
        self.send_counts = [M[s][rank] if s != rank else 0 for s in range(world)]
        # request lists: I tell every owner which of its rows I need
        asked = [torch.empty(self.send_counts[s], dtype=torch.int32, device=dev)
                 for s in range(world)]
        _p2p_round([(need[r], r) for r in range(world) if r != rank],
                   [(asked[s], s) for s in range(world) if s != rank], group)
        self.send_idx = torch.cat([asked[s] for s in range(world) if s != rank]).to(torch.int64) \
            if world > 1 else torch.empty(0, dtype=torch.int64, device=dev)
        if self.send_idx.numel():
            assert int(self.send_idx.max()) < self.n_local and int(self.send_idx.min()) >= 0
        self.n_halo = sum(recv_counts)
        self.n_send = sum(self.send_counts)
        self.last_recv_bytes = 0
        self._static = {}            # static_key -> who sends which rows (exchange_sparse)
        self.n_count_exchanges = 0   # count all-gathers run by exchange_sparse (diagnostic)
        # start of each peer's segment in send_idx / in the packed send buffer (peers in rank order)
        self.send_off, acc = [], 0
        for s in range(world):
            self.send_off.append(acc)
            acc += self.send_counts[s]
        # compact layout: [own rows | halo rows of rank 0 | rank 1 | ...]; remap the column ids
        halo_off, acc = [], 0
        for r in range(world):
            halo_off.append(acc)
            acc += recv_counts[r]
        self.halo_off = halo_off
        # global id of every halo row, in buffer order (owners ascending)
        self.halo_global = torch.cat([need[r].to(torch.int64) + bounds[r] for r in range(world) if r != rank]) \
            if world > 1 else torch.empty(0, dtype=torch.int64, device=dev)
        c = col_global.to(torch.int64)
        owner = (torch.searchsorted(b, c, right=True) - 1).clamp_(0, world - 1)
        idx = torch.searchsorted(needed, c)
        base = torch.tensor([self.n_local + halo_off[r] - pos_l[r] for r in range(world)],
                            dtype=torch.int64, device=dev)
        self.is_own = owner == rank          # per stored entry: references one of my own rows
        self.col_local = torch.where(self.is_own, c - bounds[rank], base[owner] + idx) \
            .to(torch.int32)
        self.n_buf = self.n_local + self.n_halo

    def exchange_begin(self, local):
        """Pack the rows the peers asked for and POST the grouped transfers; returns (halo buffer
        [n_halo, F], pending handles).  The buffer is defined only after exchange_end(pending);
        work enqueued in between (the product over the rank's own rows) overlaps the transfers.
        The rank's own rows are NOT copied: the product reads them in place from `local` and the
        halo rows from the returned buffer (two-block dense operand, gcn_epilogue.b2)."""
        F = local.shape[1]
        halo = torch.empty((self.n_halo, F), dtype=local.dtype, device=local.device)
        if self.world == 1:
            return halo, []
        packed = local.index_select(0, self.send_idx)
        sends, recvs = [], []
        for k in range(1, self.world):
            s = (self.rank + k) % self.world
            off = sum(self.send_counts[q] for q in range(s) if q != self.rank)
            sends.append((packed[off:off + self.send_counts[s]], s))
        for k in range(1, self.world):
            r = (self.rank - k) % self.world
            o = self.halo_off[r]
            recvs.append((halo[o:o + self.recv_counts[r]], r))
        pending = _p2p_begin(sends, recvs, self.group)
        self.last_recv_bytes = self.n_halo * F * local.element_size()
        return halo, (pending, packed)       # (the packed rows must outlive the sends)

    @staticmethod
    def exchange_end(pending):
        if pending:
            _p2p_end(pending[0])

    def exchange(self, local):
        """[n_local, F] -> the referenced remote rows [n_halo, F], complete on return."""
        halo, pending = self.exchange_begin(local)
        self.exchange_end(pending)
        return halo

    def exchange_compressed_begin(self, local):
        """Post the exchange of the requested rows of `local` [n_local, F] (F a multiple of 32) in
        COMPRESSED form — per row a bitmask of its non-zero elements (F / 8 bytes) and the non-zero
        values only — for operands that are mostly zeros (a hidden activation after ReLU and
        dropout keeps <= 25 % of its elements: ≈ 3.5 x fewer bytes than dense rows).  The message
        lengths depend on the data: one W x W all-gather of value counts and its host read precede
        the transfers.  Returns the state for exchange_compressed_end()."""
        F, dev, W = local.shape[1], local.device, self.world
        if W == 1:
            return None
        if local.is_cuda:
            # HIP path: the requested rows are read in place (no gathered copy), two passes
            bits, csum, vals = rows_pack(local, self.send_idx)
        else:
            # (CPU tensors — the gloo rehearsals of this exchange: the same format with torch ops)
            rows = local.index_select(0, self.send_idx)                   # [n_send, F]
            mask = rows != 0
            bits = pack_bits(mask)                                        # int32 [n_send, F / 32]
            vals = rows[mask]                                             # non-zero values, row-major
            per_row = mask.sum(1)
            csum = torch.zeros(self.n_send + 1, dtype=torch.int64, device=dev)
            torch.cumsum(per_row, 0, out=csum[1:])
        off = torch.tensor(self.send_off + [self.n_send], dtype=torch.int64, device=dev)
        vcut = csum[off]                                                  # value offsets per peer
        mine = (vcut[1:] - vcut[:-1]).contiguous()
        M = torch.empty(W * W, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(M, mine, group=self.group)
        M = M.view(W, W).tolist()                                         # M[s][r]: values s sends r
        vcut = vcut.tolist()
        bits_halo = torch.empty((self.n_halo, F // 32), dtype=torch.int32, device=dev)
        sends, recvs, parts = [], [], {}
        for k in range(1, W):
            p = (self.rank + k) % W
            a, b = self.send_off[p], self.send_off[p] + self.send_counts[p]
            sends += [(bits[a:b], p), (vals[vcut[p]:vcut[p + 1]], p)]
        for k in range(1, W):
            r = (self.rank - k) % W
            o = self.halo_off[r]
            parts[r] = torch.empty(M[r][self.rank], dtype=local.dtype, device=dev)
            recvs += [(bits_halo[o:o + self.recv_counts[r]], r), (parts[r], r)]
        pending = _p2p_begin(sends, recvs, self.group)
        self.last_recv_bytes = self.n_halo * (F // 8) + sum(int(t.numel()) for t in parts.values()) * local.element_size()
        return pending, (bits, vals), bits_halo, parts, F, local.dtype      # (send buffers kept alive)

    def exchange_compressed_end(self, state, like):
        """Wait for the compressed exchange and expand it: -> dense halo rows [n_halo, F]."""
        if state is None:
            return like.new_zeros((self.n_halo, like.shape[1]))
        pending, _keep, bits_halo, parts, F, dtype = state
        _p2p_end(pending)
        if not self.n_halo:
            return torch.zeros((0, F), dtype=dtype, device=bits_halo.device)
        vals = torch.cat([parts[r] for r in range(self.world) if r != self.rank])       # halo (owner) order
        if bits_halo.is_cuda:
            return rows_unpack(bits_halo, vals, F)
        halo = torch.zeros((self.n_halo, F), dtype=dtype, device=bits_halo.device)
        halo[unpack_bits(bits_halo, F)] = vals
        return halo

    def exchange_sparse(self, local, row_nonzero, static_key=None):
        """The same exchange for a ROW-SPARSE operand (gradients of a loss on few labelled vertices:
        most requested rows are entirely zero).  `row_nonzero(idx)` -> bool tensor: is row idx[i] of
        `local` non-zero.  Only the non-zero requested rows travel, each peer's message preceded by
        the positions of those rows in its request list; the receiver scatters them into a zeroed
        halo buffer.  One tiny all-gather tells every rank how many rows each peer will send (the
        only host synchronisation); every rank must call this together, like exchange().

        `static_key` (any hashable; the SAME on every rank): the caller promises that `row_nonzero`
        describes a STRUCTURAL row set that does not change between calls with this key (the loss
        rows — ShardedGCN.declare_loss_rows).  Who sends how many rows to whom is then worked out on
        the first call and kept: later calls run no count exchange and no host synchronisation.
        (Rows of the set that happen to be zero still travel: the set is structural, not data.)"""
        F, dev, W = local.shape[1], local.device, self.world
        halo = torch.zeros((self.n_halo, F), dtype=local.dtype, device=dev)
        self.last_halo_nonzero = torch.zeros(self.n_halo, dtype=torch.bool, device=dev)
        self.last_recv_bytes = 0
        if W == 1:
            return halo
        cached = self._static.get(static_key) if static_key is not None else None
        if cached is None:
            keep = row_nonzero(self.send_idx)                                   # [n_send] bool
            nz = torch.nonzero(keep).squeeze(1)                                 # sorted send positions
            off = torch.tensor(self.send_off + [self.n_send], dtype=torch.int64, device=dev)
            cut = torch.searchsorted(nz, off)                                   # split by peer
            mine = (cut[1:] - cut[:-1]).contiguous()                            # rows I send to each peer
            M = torch.empty(W * W, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(M, mine, group=self.group)
            M = M.view(W, W).tolist()                                           # M[s][r]: s sends r
            cut = cut.tolist()
            src_rows = self.send_idx.index_select(0, nz)                        # local rows to pack
            pos = (nz - off[:-1].repeat_interleave(mine)).to(torch.int32)       # position in the request
            n_nz = int(nz.numel())
            if static_key is not None:
                # (a re-declared row set bumps the version inside the key — drop the
                #  structures of the older versions instead of keeping them for ever)
                if isinstance(static_key, tuple):
                    for k in [k for k in self._static if isinstance(k, tuple) and k[:1] == static_key[:1]]:
                        del self._static[k]
                self._static[static_key] = (M, cut, src_rows, pos, n_nz)
            self.n_count_exchanges += 1
        else:
            M, cut, src_rows, pos, n_nz = cached
        rows = local.index_select(0, src_rows)                                  # packed rows
        sends, recvs, landed = [], [], []
        for k in range(1, W):
            p = (self.rank + k) % W
            sends += [(pos[cut[p]:cut[p + 1]], p), (rows[cut[p]:cut[p + 1]], p)]
        for k in range(1, W):
            r = (self.rank - k) % W
            c = M[r][self.rank]
            rp = torch.empty(c, dtype=torch.int32, device=dev)
            rr = torch.empty((c, F), dtype=local.dtype, device=dev)
            recvs += [(rp, r), (rr, r)]
            landed.append((r, rp, rr))
        _p2p_round(sends, recvs, self.group)
        self.last_halo_nonzero = torch.zeros(self.n_halo, dtype=torch.bool, device=dev)
        for r, rp, rr in landed:
            if rp.numel():
                dst = rp.to(torch.int64) + self.halo_off[r]
                halo.index_copy_(0, dst, rr)
                self.last_halo_nonzero[dst] = True
        self.last_sparse_rows = (n_nz, self.n_send)                         # sent / dense (stats)
        self.last_recv_bytes = sum(int(rp.numel()) for _, rp, _ in landed) * (F * local.element_size() + 4)
        return halo


class ShardedGraph:
    """Rank-local view of Â: the row block of Â and of Âᵀ, both with columns remapped to the
    padded all-gather layout.  Accepted as `adj` by GraphConvolution.forward."""

    def __init__(self, bounds, rank, world, a_block, at_block, group=None, exchange="halo",
                 graph_factory=CSRGraph, spmm_fn=spmm_csr, bwd_fn=_grad_pre_and_bias,
                 sparse_grad_exchange=True, overlap=True, compress_hidden=False, **plan_kw):
        if exchange not in ("halo", "allgather", "rccl-allgather"):
            raise RuntimeError("exchange must be 'halo', 'allgather' or 'rccl-allgather'")
        # "rccl-allgather" = the all-gather layout moved by the COLLECTIVE itself
        # (dist.all_gather_into_tensor: RCCL's all-gather on "nccl" — the north star's literal
        # "RCCL all-gather of activations"); "allgather" moves the same bytes as one grouped
        # point-to-point round (a direct mesh exchange)
        self._rccl_gather = exchange == "rccl-allgather"
        if self._rccl_gather:
            exchange = "allgather"
        # dense halo exchanges are pipelined by source block (own rows | halo rows), see product()
        self.overlap = bool(overlap) and exchange == "halo" and world > 1
        self._graph_factory, self._plan_kw = graph_factory, plan_kw
        self._split = {}       # transpose? -> (A_own, A_halo_plus_identity), built on first use
        # backward exchanges send only the non-zero gradient rows (halo mode; must be set
        # identically on every rank: it selects the message protocol)
        self.sparse_grad_exchange = bool(sparse_grad_exchange) and exchange == "halo"
        # hidden-layer inputs (>= 75 % zeros after ReLU + dropout) travel as bitmask + non-zero
        # values and are multiplied by the weight on arrival (product_hidden); must be set
        # identically on every rank: it selects the message protocol
        self.compress_hidden = bool(compress_hidden) and exchange == "halo"
        self._hinted_product = spmm_fn is spmm_csr     # test stand-ins take no operand hint
        self.bounds, self.rank, self.world, self.group = list(bounds), rank, world, group
        self.exchange_mode = exchange
        self.n_global = bounds[-1]
        self.r0, self.r1 = bounds[rank], bounds[rank + 1]
        self.n_local = self.r1 - self.r0
        self.max_rows = max(bounds[i + 1] - bounds[i] for i in range(world))
        self._spmm = spmm_fn
        self._bwd = bwd_fn
        self.nnz_local = int(a_block[1].numel())
        self.halo = self.halo_t = None
        blocks = []
        for rp, c, v in (a_block, at_block):
            if exchange == "halo":
                h = HaloExchange(c, bounds, rank, world, group)
                blocks.append((graph_factory(rp, h.col_local, v, (self.n_local, h.n_buf),
                                             **plan_kw), h))
            else:
                n_pad = self.world * self.max_rows
                blocks.append((graph_factory(rp, remap_columns(c, bounds, self.max_rows), v,
                                             (self.n_local, n_pad), **plan_kw), None))
        (self.A, self.halo), (self.At, self.halo_t) = blocks
        self._raw = {False: (a_block[0], a_block[2]), True: (at_block[0], at_block[2])}
        self.last_recv_bytes = {"fwd": 0, "bwd": 0}   # bytes this rank received in the last exchange
        self.timing = None    # optional list: (tag, start_event, end_event) per exchange+product
        self._const_ref = None      # weakref to the registered constant input (feature matrix)
        self._const_halo = None     # (version, halo rows) of that tensor
        self.n_const_exchanges = 0
        self.setup_stats = None     # filled by the shard-local constructors (from_rmat)
        self.static_grad_rows = None    # see declare_grad_rows
        self._grad_rows_version = 0
        # forward exchange of a halo-mode graph may be switched to an all-gather form (bench.py's
        # pre-timed A/B, set_forward_exchange); the padded block is built on first use
        self.fwd_exchange = "halo" if exchange == "halo" else ("rccl-allgather" if self._rccl_gather else "allgather")
        self._A_pad = None
        self._col_global_fwd = a_block[1] if exchange == "halo" else None

    @classmethod
    def from_global_csr(cls, rowptr, col, val, n, rank, world, device=None, group=None, **kw):
        """Every rank holds (or has generated) the same global CSR; keep this rank's blocks.
        For graphs that fit one process (tests, small inputs); see from_row_block / from_rmat for
        construction that never materialises the whole matrix on a rank."""
        bounds = partition_rows(rowptr, world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        a_block = row_block(rowptr, col, val, r0, r1)
        at_block = transpose_row_block(rowptr, col, val, n, r0, r1)
        if device is not None:
            a_block = tuple(t.to(device) for t in a_block)
            at_block = tuple(t.to(device) for t in at_block)
        return cls(bounds, rank, world, a_block, at_block, group=group, **kw)

    @classmethod
    def from_row_block(cls, bounds, rank, world, a_block, group=None, **kw):
        """Shard-local construction: this rank supplies ONLY its rows [b_r, b_r+1) of Â (rebased
        rowptr, global column ids, values); its rows of Âᵀ are assembled from the triplets the
        other ranks hold for it (transpose_block_by_exchange).  Collective.  Per-rank memory is
        O(nnz / P): a graph larger than one GPU can be ingested."""
        at_block = transpose_block_by_exchange(a_block, bounds, rank, world, group)
        return cls(bounds, rank, world, a_block, at_block, group=group, **kw)

    @classmethod
    def from_rmat(cls, n, n_edges, rank, world, device, seed=42, perm_seed=43, group=None, **kw):
        """The synthetic graph of configs C3–C5 (`utils.rmat_graph(n, n_edges, seed, perm_seed)`:
        the same matrix, entry for entry) built shard-locally in ONE pass over the edge stream:
        (1) every rank replays the stream once and keeps the (deduplicated) entries of a
        provisional uniform row block, (2) the per-row counts are all-gathered (n integers) and
        give the nnz-balanced bounds, (3) the ranks hand the rows that changed owner straight to
        their final owners (`redistribute_rows`: one grouped point-to-point round of sorted key
        ranges — a boundary moves by a fraction of a block, so little travels), (4) the Âᵀ blocks
        are assembled by the triplet exchange.  The edge stream is replayed, never stored:
        O(chunk) + O(nnz / P) memory per rank.  `setup_stats` records what moved."""
        from .utils import csr_from_keys, rmat_block_keys
        step = -(-n // world)
        p0, p1 = min(rank * step, n), min((rank + 1) * step, n)
        key = rmat_block_keys(n, n_edges, p0, p1, seed, perm_seed, device)
        deg = torch.zeros(step, dtype=torch.int64, device=device)
        deg[:p1 - p0] = torch.bincount(key // n - p0, minlength=p1 - p0)
        allc = torch.empty(world * step, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allc, deg, group=group)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
        torch.cumsum(allc[:n], 0, out=rowptr[1:])      # blocks are contiguous: padding only at the end
        bounds = partition_rows(rowptr, world)
        del rowptr, allc, deg
        key, moved = redistribute_rows(key, n, bounds, rank, world, group)
        a_block = csr_from_keys(key, n, bounds[rank], bounds[rank + 1])
        del key
        sg = cls.from_row_block(bounds, rank, world, a_block, group=group, **kw)
        sg.setup_stats = {"edge_stream_passes": 1, "entries_received_in_rebalance": moved}
        return sg

    # ---------------------------------------------------------------- exchange step
    def set_forward_exchange(self, mode):
        """Halo-mode graphs only: how the FORWARD dense exchange of a hidden layer moves its rows —
        "halo" (the constructor's: only the rows this rank's block references, pipelined by source
        block), "allgather" (every row to every rank, one grouped point-to-point round) or
        "rccl-allgather" (the same layout through RCCL's all-gather collective).  The backward
        exchanges (row-sparse / static gradient halo) and the constant-input halo are unchanged.
        Must be set identically on every rank; the padded block is built on first use."""
        if mode not in ("halo", "allgather", "rccl-allgather"):
            raise RuntimeError("forward exchange must be 'halo', 'allgather' or 'rccl-allgather'")
        if self.exchange_mode != "halo":
            raise RuntimeError("set_forward_exchange: the graph was built in all-gather mode")
        if mode != "halo" and self._A_pad is None:
            rp, val = self._raw[False]
            self._A_pad = self._graph_factory(rp, remap_columns(self._col_global_fwd, self.bounds, self.max_rows),
                                              val, (self.n_local, self.world * self.max_rows), **self._plan_kw)
        self.fwd_exchange = mode

    def all_gather_rows(self, local, rccl=None):
        """[n_local, F] on every rank -> padded [P*max_rows, F] (rows past n_local of each slot
        are never referenced by the remapped column indices).
        Default: a DIRECT (mesh) all-gather — every rank sends its block straight to each of the
        P-1 peers in one grouped point-to-point round: xGMI is a full mesh of point-to-point links,
        so all 7 transfers run concurrently at one block-time, where a ring all-gather makes 7
        sequential hops over one link (≈ 8.4 ms vs ≈ 58.6 ms for 1.28 GB blocks).
        `rccl=True` ("rccl-allgather"): ONE call of the collective, dist.all_gather_into_tensor
        (RCCL's all-gather on the "nccl" backend), every rank contributing its padded max_rows
        slot of the same buffer; which of the two is faster on a given node is what bench.py's
        pre-timed A/B measures."""
        rccl = self._rccl_gather if rccl is None else rccl
        F = local.shape[1]
        out = torch.empty((self.world * self.max_rows, F), dtype=local.dtype, device=local.device)
        slot = out[self.rank * self.max_rows:(self.rank + 1) * self.max_rows]
        slot[:self.n_local].copy_(local)
        if self.world > 1 and rccl:
            # (the slot's tail past n_local travels too — never read by anybody; on RCCL the input
            #  may alias its own slot of the output: the in-place form of the collective.  gloo — the
            #  CPU rehearsals — gets a separate input.)
            src = slot if local.is_cuda and dist.get_backend(self.group) == "nccl" else slot.clone()
            dist.all_gather_into_tensor(out, src, group=self.group)
            return out
        if self.world > 1:
            src = local if local.is_contiguous() else local.contiguous()
            sends, recvs = [], []
            for k in range(1, self.world):
                p = (self.rank + k) % self.world
                sends.append((src, p))
            for k in range(1, self.world):
                q = (self.rank - k) % self.world
                n_q = self.bounds[q + 1] - self.bounds[q]
                recvs.append((out[q * self.max_rows:q * self.max_rows + n_q], q))
            _p2p_round(sends, recvs, self.group)
        return out

    def split_block(self, transpose=False):
        """The rank's block cut by SOURCE block for the pipelined dense exchange:
            A_own  [n_local, n_local]            entries that reference the rank's own rows
            A_halo [n_local, n_halo + n_local]   entries that reference halo rows, plus one entry
                                                 (i, n_halo + i) = 1 per row
        so that   out = A_halo · [halo rows ; A_own · own rows]   equals the block's product: the
        identity entries add the partial result of the own-rows product (second block of the
        two-block dense operand) inside the same launch that applies the fused epilogue — no
        accumulate mode, no extra pass.  Built once per block on first use."""
        if transpose not in self._split:
            h = self.halo_t if transpose else self.halo
            rowptr, val = self._raw[transpose]
            dev, n_loc = val.device, self.n_local
            deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
            row = torch.repeat_interleave(torch.arange(n_loc, device=dev, dtype=torch.int64), deg)
            own = h.is_own

            def csr(rows, cols, vals, n_cols):
                rp = torch.zeros(n_loc + 1, dtype=torch.int64, device=dev)
                torch.cumsum(torch.bincount(rows, minlength=n_loc), 0, out=rp[1:])
                return self._graph_factory(rp.to(rowptr.dtype), cols.to(torch.int32), vals,
                                           (n_loc, n_cols), **self._plan_kw)
            a_own = csr(row[own], h.col_local[own], val[own], n_loc)
            ident = torch.arange(n_loc, device=dev, dtype=torch.int64)
            rows = torch.cat([row[~own], ident])
            cols = torch.cat([h.col_local[~own].to(torch.int64) - n_loc, ident + h.n_halo])
            vals = torch.cat([val[~own], torch.ones(n_loc, dtype=val.dtype, device=dev)])
            rows, perm = torch.sort(rows, stable=True)       # halo entries first, identity last
            self._split[transpose] = (a_own, csr(rows, cols[perm], vals[perm], h.n_halo + n_loc))
        return self._split[transpose]

    # ---------------------------------------------------------------- static loss rows
    def declare_grad_rows(self, mask_local):
        """`mask_local` (bool [n_local]) — the rows of this rank's block on which the gradient
        entering the LAST layer can be non-zero (the loss rows), or None to withdraw.  With it, the
        row-sparse backward exchange of that layer is STATIC: the count exchange and its host
        synchronisation happen once, not per step.  Must be called on every rank alike (the
        protocol — count exchange or not — follows from it)."""
        if mask_local is not None:
            if mask_local.dtype != torch.bool or mask_local.numel() != self.n_local:
                raise RuntimeError("declare_grad_rows: bool mask over this rank's rows expected")
            self._grad_rows_version += 1
        self.static_grad_rows = mask_local

    # ---------------------------------------------------------------- constant input
    def register_constant_input(self, t):
        """Declare `t` [n_local, F] (this rank's rows of the input feature matrix) constant across
        steps: its halo rows are exchanged once (and again only if `t` is modified in place)."""
        if self._const_ref is None or self._const_ref() is not t:
            self._const_ref, self._const_halo = weakref.ref(t), None

    def is_constant_input(self, t):
        return (self.exchange_mode == "halo" and self._const_ref is not None
                and self._const_ref() is t and not t.requires_grad)

    def constant_halo(self, t):
        """Halo rows of the registered constant input; collective on first use per version."""
        if self._const_halo is None or self._const_halo[0] != t._version:
            halo = self.halo.exchange(t.detach())
            m = torch.linalg.vector_norm(t.detach().float(), ord=float("inf")).reshape(1)
            if halo.shape[0]:
                m = torch.maximum(m, torch.linalg.vector_norm(halo.float(), ord=float("inf")).reshape(1))
            self._const_halo = (t._version, halo, m)
            self.n_const_exchanges += 1
        return self._const_halo[1]

    def constant_absmax(self, t):
        """max|[t ; halo rows of t]| (device float [1]) of the registered constant input, computed
        with the halo exchange and kept with it."""
        self.constant_halo(t)
        return self._const_halo[2]

    def _tic(self, like):
        if self.timing is None or not like.is_cuda:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        return ev

    def _toc(self, ev, tag):
        if ev is not None:
            ev[1].record()
            self.timing.append((tag, ev[0], ev[1]))

    def product(self, local, transpose=False, bias=None, relu=False, dropout_p=0.0, seed=0,
                row_nonzero=None, own_flags=None, log_softmax=False, static_key=None):
        """Exchange + local product.  `row_nonzero` (callable idx -> bool, see
        HaloExchange.exchange_sparse) marks `local` as row-sparse: only its non-zero rows travel;
        with `own_flags` (bool [n_local], the same information for all own rows) the local product
        also gets the operand hint [own flags | flags of the received rows] and skips zero rows.

        A dense halo exchange is PIPELINED BY SOURCE BLOCK (SURVEY §8e lever i): the transfers are
        posted, the product over the entries that reference the rank's own rows runs while the
        halo rows are in flight, and the product over the halo entries (which folds the first
        partial result in through its identity entries and applies the epilogue) follows the
        arrival — see split_block()."""
        ev = self._tic(local)
        # (the same seed on every rank + the block's first global row in the counter: the masks of
        #  the single-GPU run, whatever the number of ranks)
        kw = {"dropout_p": dropout_p, "seed": seed, "row_base": self.r0} if dropout_p > 0.0 else {}
        if log_softmax:          # (in the store of the launch that completes the rows)
            kw["log_softmax"] = True
        tag = "bwd_local" if transpose else "fwd_local"
        which = "bwd" if transpose else "fwd"
        gather_fwd = (self.exchange_mode == "halo" and not transpose and row_nonzero is None
                      and self.fwd_exchange != "halo")
        if gather_fwd:        # (halo-mode graph whose forward exchange was switched to an all-gather)
            gathered = self.all_gather_rows(local, rccl=self.fwd_exchange == "rccl-allgather")
            self.last_recv_bytes[which] = ((self.n_global - self.n_local) * local.shape[1]
                                           * local.element_size())
            out = self._spmm(self._A_pad, gathered, bias=bias, relu=relu, tag=tag, **kw)
        elif self.exchange_mode == "halo":
            h = self.halo_t if transpose else self.halo
            if row_nonzero is None and self.overlap:
                a_own, a_halo = self.split_block(transpose)
                halo, pending = h.exchange_begin(local)
                part = self._spmm(a_own, local, tag=tag)          # overlaps the transfers
                ev_w = self._tic(local)                           # (diagnostic: exposed transfer time)
                h.exchange_end(pending)
                self._toc(ev_w, which + "_wait")
                out = self._spmm(a_halo, halo, bias=bias, relu=relu, tag=tag, B2=part, **kw)
                self.last_recv_bytes[which] = h.last_recv_bytes
                self._toc(ev, which)
                return out
            halo = h.exchange_sparse(local, row_nonzero, static_key) if row_nonzero is not None else \
                h.exchange(local)
            self.last_recv_bytes[which] = h.last_recv_bytes
            if row_nonzero is not None and own_flags is not None and self._hinted_product:
                kw["b_hint"] = pack_row_flags(torch.cat([own_flags, h.last_halo_nonzero]))
            # dense operand = [own rows (in place) ; halo rows]
            out = self._spmm(self.At if transpose else self.A, local, bias=bias, relu=relu,
                             tag=tag, B2=halo, **kw)
        else:
            gathered = self.all_gather_rows(local)
            self.last_recv_bytes[which] = ((self.n_global - self.n_local) * local.shape[1]
                                           * local.element_size())
            out = self._spmm(self.At if transpose else self.A, gathered, bias=bias, relu=relu,
                             tag=tag, **kw)
        self._toc(ev, which)
        return out

    def product_hidden(self, h_local, weight, bias=None, relu=False, dropout_p=0.0, seed=0,
                       log_softmax=False, h_bound=None):
        """out_r = epilogue(Â_r · ([h_r ; h_halo] · W) + b) for a HIDDEN-layer input h (mostly zeros
        after ReLU + dropout): the halo rows of h travel compressed (bitmask + non-zero values,
        HaloExchange.exchange_compressed_begin), are expanded on arrival and multiplied by W here
        — the GEMM of the halo rows is recomputed locally instead of its dense result being sent.
        Pipelined like product(): the own-rows GEMM and the product over the entries that
        reference own rows run while the compressed rows are in flight.  Same result as
        product(h·W) up to fp32 summation order."""
        from .spmm import _dense_forward as gemm
        ev = self._tic(h_local)
        h = self.halo
        kw = {"dropout_p": dropout_p, "seed": seed, "row_base": self.r0} if dropout_p > 0.0 else {}
        if log_softmax:
            kw["log_softmax"] = True
        a_own, a_halo = self.split_block(False)
        state = h.exchange_compressed_begin(h_local.detach())
        h_halo = None
        if not self.overlap:                  # (--no-overlap is an A/B switch here too)
            h_halo = h.exchange_compressed_end(state, h_local)
        sup_own = gemm(h_local, weight, h_bound)
        part = self._spmm(a_own, sup_own, tag="fwd_local")            # overlaps the transfers
        ev_w = self._tic(h_local)
        if h_halo is None:
            h_halo = h.exchange_compressed_end(state, h_local)
        self._toc(ev_w, "fwd_wait")
        # (the halo rows come from OTHER ranks — this rank's bound of max|h| does not
        #  cover them, and the scaled fp16 scheme overflows silently above 2-4x its bound.  No bound:
        #  gemm_xw256 takes one absmax pass over the small halo block under "h2"; the default
        #  three-part bf16 scheme needs none.)
        sup_halo = gemm(h_halo, weight, None) if h_halo.shape[0] else \
            h_halo.new_empty((0, weight.shape[1]))
        out = self._spmm(a_halo, sup_halo, bias=bias, relu=relu, tag="fwd_local", B2=part, **kw)
        self.last_recv_bytes["fwd"] = h.last_recv_bytes
        self._toc(ev, "fwd")
        return out

    def exchange_rows(self):
        """(rows received per forward product, rows an all-gather would have received)."""
        recv = self.halo.n_halo if self.halo is not None else self.n_global - self.n_local
        return recv, self.n_global - self.n_local

    def __repr__(self):
        return (f"ShardedGraph(rank {self.rank}/{self.world}, rows [{self.r0},{self.r1}) of "
                f"{self.n_global}, nnz_local {self.nnz_local}, exchange {self.exchange_mode})")


def _check_static_rows(grad, flags):
    """Opt-in debug check (PYGCN_CHECK_STATIC_ROWS=1; a host synchronisation): a gradient handed
    to the STATIC row-sparse exchange must be zero outside the declared loss rows — a second loss on
    other rows while the declaration stands would otherwise lose those rows silently."""
    import os
    if os.environ.get("PYGCN_CHECK_STATIC_ROWS") == "1":
        stray = (grad != 0).any(1) & ~flags
        if bool(stray.any()):
            raise RuntimeError(f"static gradient rows: {int(stray.sum())} non-zero gradient rows lie outside "
                               "the rows declared by declare_loss_rows / declare_grad_rows")


class ShardedSpMMFunction(torch.autograd.Function):
    """out_r = Â_r · allgather(support);  grad_support_r = (Âᵀ)_r · allgather(grad_out)."""

    @staticmethod
    def forward(ctx, sg, support_local, bias, relu=False, dropout_p=0.0, seed=0, log_softmax=False,
                last_layer=False):
        if dropout_p > 0.0 and not relu:
            raise RuntimeError("fused dropout needs the fused ReLU (out > 0 encodes the mask)")
        if log_softmax and relu:
            raise RuntimeError("log_softmax cannot be combined with the fused ReLU / dropout")
        ctx.sg = sg
        ctx.last_layer = bool(last_layer or log_softmax)
        ctx.has_bias = bias is not None
        ctx.relu = bool(relu)
        ctx.log_softmax = bool(log_softmax)
        ctx.scale = dropout_scale(dropout_p)
        out = sg.product(support_local, transpose=False, bias=bias, relu=relu,
                         dropout_p=dropout_p, seed=seed, log_softmax=log_softmax)
        if relu or log_softmax:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        grad_support = None
        out = ctx.saved_tensors[0] if (ctx.relu or ctx.log_softmax) else None
        # grad_bias is this rank's partial sum: summed over ranks by allreduce_grads
        sg = ctx.sg
        grad_out, grad_bias, hint = sg._bwd(grad_out, out, ctx.relu, ctx.scale,
                                            ctx.has_bias and ctx.needs_input_grad[2],
                                            **({"log_softmax": True} if ctx.log_softmax else {}))
        if ctx.needs_input_grad[1]:
            grad_out = grad_out.contiguous()
            row_nonzero = flags = static_key = None
            if sg.sparse_grad_exchange:
                if ctx.last_layer and sg.static_grad_rows is not None:
                    # the loss rows were declared (declare_grad_rows): a structural row set — the
                    # count exchange ran once, this step synchronises with nobody
                    flags = sg.static_grad_rows
                    static_key = ("loss rows", sg._grad_rows_version)
                    _check_static_rows(grad_out, flags)
                else:
                    # gradients of a loss on few labelled vertices: most rows are zero and need not
                    # travel.  The fused backward pass already produced the row bitmap; without it
                    # (shapes outside that kernel) the flags are computed here
                    flags = unpack_row_flags(hint[0], grad_out.shape[0]) if hint is not None else \
                        (grad_out != 0).any(1)
                row_nonzero = lambda idx: flags[idx]
            grad_support = sg.product(grad_out, transpose=True, row_nonzero=row_nonzero,
                                      own_flags=flags if row_nonzero is not None else None,
                                      static_key=static_key)
        return None, grad_support, grad_bias, None, None, None, None, None


class ShardedHiddenLayerFunction(torch.autograd.Function):
    """The layer on a sharded graph with a HIDDEN activation as its input and the compressed
    exchange switched on (ShardedGraph(compress_hidden=True)):

        forward   out_r = epilogue(Â_r · ([h_r ; h_halo] · W) + b)      (product_hidden: rows of h travel
                                                                       as bitmask + values, the halo
                                                                       rows' GEMM runs on arrival)
        backward  as the uncompressed layer: grad_support_r = (Âᵀ)_r · exchange(grad_pre) (row-sparse
                  exchange), grad_W = h_rᵀ · grad_support_r, grad_h = grad_support_r · Wᵀ — the
                  recomputed halo GEMM is a replica of its owner's rows and owns no gradient."""

    @staticmethod
    def forward(ctx, sg, h_local, weight, bias, relu=False, dropout_p=0.0, seed=0, log_softmax=False,
                last_layer=False):
        if dropout_p > 0.0 and not relu:
            raise RuntimeError("fused dropout needs the fused ReLU (out > 0 encodes the mask)")
        ctx.sg = sg
        ctx.has_bias = bias is not None
        ctx.relu, ctx.log_softmax = bool(relu), bool(log_softmax)
        ctx.last_layer = bool(last_layer or log_softmax)
        ctx.scale = dropout_scale(dropout_p)
        from .spmm import known_absmax
        bound = known_absmax(h_local) if (h_local.is_cuda and h_local.dtype == torch.float32) else None
        out = sg.product_hidden(h_local, weight, bias=bias, relu=relu, dropout_p=dropout_p, seed=seed,
                                log_softmax=log_softmax, h_bound=bound)
        ctx.save_for_backward(h_local, weight, *([out] if (relu or log_softmax) else []))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from .spmm import _dense_grads
        sg = ctx.sg
        h_local, weight = ctx.saved_tensors[:2]
        out = ctx.saved_tensors[2] if (ctx.relu or ctx.log_softmax) else None
        need_h, need_w, need_b = ctx.needs_input_grad[1:4]
        grad_pre, grad_bias, hint = sg._bwd(grad_out, out, ctx.relu, ctx.scale, ctx.has_bias and need_b,
                                            **({"log_softmax": True} if ctx.log_softmax else {}))
        grad_h = grad_w = None
        if need_h or need_w:
            grad_pre = grad_pre.contiguous()
            row_nonzero = flags = static_key = None
            if sg.sparse_grad_exchange:
                if ctx.last_layer and sg.static_grad_rows is not None:
                    flags, static_key = sg.static_grad_rows, ("loss rows", sg._grad_rows_version)
                    _check_static_rows(grad_pre, flags)
                else:
                    flags = unpack_row_flags(hint[0], grad_pre.shape[0]) if hint is not None else \
                        (grad_pre != 0).any(1)
                row_nonzero = lambda idx: flags[idx]
            grad_support = sg.product(grad_pre, transpose=True, row_nonzero=row_nonzero,
                                      own_flags=flags if row_nonzero is not None else None,
                                      static_key=static_key)
            grad_h, grad_w = _dense_grads(h_local, weight, grad_support, need_h, need_w)
        return None, grad_h, grad_w, grad_bias, None, None, None, None, None


class ShardedInputLayerFunction(torch.autograd.Function):
    """The layer on a sharded graph when its input is the registered constant feature block (no
    gradient flows to it): no exchange in forward or backward — see the module docstring.

        out_r         = Â_r · ([X_r ; X_halo] · W) + b      (+ fused ReLU / dropout)
        grad_W (part) = (Â_r · [X_r ; X_halo])ᵀ · grad_pre_r

    Both products use this rank's forward block Â_r with the two-block dense operand; the partial
    grad_W / grad_b are summed over ranks by ShardedGCN.allreduce_grads like every other
    parameter gradient."""

    @staticmethod
    def forward(ctx, sg, x_local, x_halo, weight, bias, relu=False, dropout_p=0.0, seed=0):
        if dropout_p > 0.0 and not relu:
            raise RuntimeError("fused dropout needs the fused ReLU (out > 0 encodes the mask)")
        ctx.sg = sg
        ctx.has_bias = bias is not None
        ctx.relu = bool(relu)
        ctx.scale = dropout_scale(dropout_p)
        ev = sg._tic(x_local)
        kw = {"dropout_p": dropout_p, "seed": seed, "row_base": sg.r0} if dropout_p > 0.0 else {}
        # REASSOCIATED where the GEMM kernel can carry the epilogue (256 -> 256 fp32, HIP):
        #     out_r = epilogue((Â_r · [X_r ; X_halo]) · W + b)
        # — the same local product at the same width, then ONE GEMM over the rank's own rows (the
        # GEMM of the halo rows disappears), and z_r = Â_r·[X_r ; X_halo] of this forward pass is all
        # the backward pass needs for grad_W (no local product in backward either).
        out = z = None
        from .spmm import absmax_cached, layer_gemm, layer_gemm_reassociable
        if sg._hinted_product and x_local.is_cuda and layer_gemm_reassociable(x_local, weight, bias):
            z = sg._spmm(sg.A, x_local, tag="fwd_local", B2=x_halo)
            sg._toc(ev, "fwd")       # (the window of the product, as in the other branch)
            ev = None
            zb = None
            ctx.z_bound = None
            if x_local.dtype == torch.float32:
                xb = sg.constant_absmax(x_local) if sg.is_constant_input(x_local) else (
                    torch.maximum(absmax_cached(x_local), torch.linalg.vector_norm(
                        x_halo, ord=float("inf")).reshape(1)) if x_halo.shape[0] else absmax_cached(x_local))
                zb = ctx.z_bound = sg.A.inf_norm() * xb * 1.0001
            y_max = torch.zeros(1, dtype=torch.float32, device=z.device) if zb is not None else None
            out = layer_gemm(z, weight, zb, y_max, bias=bias, relu=relu, **kw)
            if out is None:
                z = None
            elif y_max is not None:
                from .spmm import remember_absmax
                remember_absmax(out, y_max)          # the next layer's GEMM scales by it
        ctx.reassoc = out is not None
        if out is None:
            sup_own = _dense_forward(x_local, weight)
            sup_halo = _dense_forward(x_halo, weight) if x_halo.shape[0] else \
                x_halo.new_empty((0, weight.shape[1]))
            ev = sg._tic(sup_own)
            out = sg._spmm(sg.A, sup_own, bias=bias, relu=relu, tag="fwd_local", B2=sup_halo, **kw)
            sg._toc(ev, "fwd")
        ctx.save_for_backward(z if ctx.reassoc else x_local, x_halo, weight, *([out] if relu else []))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        sg = ctx.sg
        x_local, x_halo = ctx.saved_tensors[:2]
        out = ctx.saved_tensors[3] if ctx.relu else None
        grad_pre, grad_bias, hint = sg._bwd(grad_out, out, ctx.relu, ctx.scale,
                                            ctx.has_bias and ctx.needs_input_grad[4])
        grad_w = None
        if ctx.needs_input_grad[3] and ctx.reassoc:
            # (x_local is z_r = Â_r·X here.)  With both maxima known — ‖Â‖∞·max|X| and the maximum the
            # GEMM above this layer reported for grad_out — the gather-fused MFMA kernel runs
            from .spmm import known_absmax
            g_bound = known_absmax(grad_out) if ctx.z_bound is not None else None
            grad_w = _weight_grad(x_local, grad_pre.contiguous(), ctx.z_bound,
                                  g_bound * ctx.scale if g_bound is not None else None)
        elif ctx.needs_input_grad[3]:
            grad_pre = grad_pre.contiguous()
            # rows of Â_r · X that meet an all-zero row of grad_pre add nothing: with the bitmap
            # of the fused backward pass the product computes only the others (c_select) and the
            # GEMM runs over them (a purely local decision: no collective depends on it)
            sparse = (hint is not None and sg._hinted_product
                      and tuning.below(int(hint[1].item()), grad_pre.shape[0], tuning.SPARSE_GEMM_MAX_SHARE))
            ev = sg._tic(grad_pre)
            z = sg._spmm(sg.A, x_local, tag="bwd_local", B2=x_halo,   # this rank's rows of Â · X
                         **({"c_select": hint[0]} if sparse else {}))
            sg._toc(ev, "bwd")
            if sparse:
                rows = torch.nonzero(unpack_row_flags(hint[0], grad_pre.shape[0])).squeeze(1)
                grad_w = _weight_grad(z.index_select(0, rows), grad_pre.index_select(0, rows))
            else:
                grad_w = _weight_grad(z, grad_pre)
        return None, None, None, grad_w, grad_bias, None, None, None


class ShardedGCN(torch.nn.Module):
    """A replicated GCN driven on row-block shards: forward(x_local, sharded_adj)."""

    def __init__(self, model, sg):
        super().__init__()
        self.model, self.sg = model, sg
        self._flat = None
        self._row_sets = {}

    def forward(self, x_local, sg=None, rows=None):
        """`rows` (extension, like GCN.forward): local row ids the loss reads — returns
        `output[rows]` and, where the shapes allow, runs the whole step of this rank as one autograd
        node with a static halo of gradient rows (pygcn_amd/sharded_fused.py).  EVERY rank must
        pass its rows on the same call (the first call per row set builds the structure
        collectively)."""
        sg = sg if sg is not None else self.sg
        if sg.exchange_mode == "halo" and not x_local.requires_grad:
            sg.register_constant_input(x_local)    # the feature rows: exchanged once, then kept
        if rows is None:
            return self.model(x_local, sg)
        from . import sharded_fused as sf
        from .spmm import dropout_seed_for
        if not sf.fusable(sg, self.model, x_local):
            return self.model(x_local, sg)[rows.rows_user if isinstance(rows, sf.ShardedRowSets) else rows]
        if isinstance(rows, sf.ShardedRowSets):
            rs = rows                              # prepared handle: nothing collective, no lookup
        else:
            # A tensor: look it up by identity — and make the build decision SYMMETRIC.  The
            # constructor is collective; if one rank re-created its index tensor (a miss) while
            # the others still hit, the collectives would not match and the job would hang.  One
            # tiny all-reduce of a "must rebuild" flag per call (a host synchronisation: training
            # loops should pass the handle of prepare_rows() instead — bench.py does).
            from .fused import rows_key
            key = rows_key(rows)
            hit = self._row_sets.get(key)
            miss = torch.tensor([0 if hit is not None else 1], dtype=torch.int32, device=x_local.device)
            dist.all_reduce(miss, op=dist.ReduceOp.MAX, group=sg.group)
            if int(miss.item()):
                if len(self._row_sets) >= 4:
                    self._row_sets.clear()
                hit = self._row_sets[key] = (sf.ShardedRowSets(sg, rows), rows)  # (keeps `rows` alive)
            rs = hit[0]
        m = self.model
        p = m.dropout if m.training else 0.0
        seed = dropout_seed_for(x_local) if p > 0.0 else 0
        return sf.ShardedGCN2RowsFunction.apply(sg, rs, x_local, sg.constant_halo(x_local),
                                                m.gc1.weight, m.gc1.bias, m.gc2.weight, m.gc2.bias,
                                                float(p), seed)

    def declare_loss_rows(self, idx_local):
        """Layer-by-layer path: tell the sharded graph which local rows the loss reads (None to
        withdraw), so the row-sparse gradient exchange of the last layer becomes static
        (ShardedGraph.declare_grad_rows).  Every rank must call it alike."""
        if idx_local is None:
            self.sg.declare_grad_rows(None)
            return
        mask = torch.zeros(self.sg.n_local, dtype=torch.bool, device=idx_local.device)
        mask[idx_local.long()] = True
        self.sg.declare_grad_rows(mask)

    def prepare_rows(self, rows_local):
        """COLLECTIVE: the static backward structure for the loss rows of every rank (local row
        ids of the caller's block; may be empty) — pygcn_amd/sharded_fused.py.  Pass the returned
        handle as `rows=`: the forward call then involves no cache lookup, no flag exchange and
        no host synchronisation."""
        from . import sharded_fused as sf
        return sf.ShardedRowSets(self.sg, rows_local)

    def nll_loss(self, logp_local, labels_local, idx_local=None):
        """This rank's share of the global-mean NLL over the (optionally index-selected) nodes of
        all ranks: local sum / global count.  Backpropagating it on every rank and summing the
        parameter gradients (allreduce_grads) gives the gradient of the global mean."""
        if idx_local is not None:
            logp_local, labels_local = logp_local[idx_local], labels_local[idx_local]
        cnt = torch.tensor([float(labels_local.numel())], device=logp_local.device)
        dist.all_reduce(cnt, group=self.sg.group)
        s = torch.nn.functional.nll_loss(logp_local, labels_local, reduction="sum") \
            if labels_local.numel() else logp_local.sum() * 0.0
        return s / cnt[0]

    def global_loss(self, loss_share):
        """Sum of the ranks' shares = the global-mean loss (for reporting)."""
        t = loss_share.detach().clone().reshape(1)
        dist.all_reduce(t, group=self.sg.group)
        return float(t.item())

    def allreduce_grads(self):
        """Sum parameter gradients over ranks (one flat bucket; 2·F² + 2·F floats)."""
        params = [p for p in self.model.parameters() if p.grad is not None]
        if not params:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        dist.all_reduce(flat, group=self.sg.group)
        off = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
