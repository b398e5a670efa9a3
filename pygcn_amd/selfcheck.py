"""Self-validation of the row-block sharded path on the hardware it actually runs on.

The reference has no multi-device code (`pygcn/train.py:30` pins one GPU), and the
builder of this repository never sees more than one GPU: the first run between GPUs is somebody
else's.  So that first run validates ITSELF, outside any timed region (bench.py calls these at
world > 1; tests/test_sharded_cpu.py runs them over gloo with a deliberately corrupted exchange to
prove that they fire):

  * `overlap_selftest`    the pipelined dense exchange (transfers posted, own-rows product under
                          them, halo product after the wait — ShardedGraph.product) relies on the
                          backend ordering the transfers against the compute stream.  The same
                          operand goes through the pipelined and the unpipelined form; on a
                          mismatch the graph falls back to the unpipelined form and says so.
  * `sharded_grad_check`  one training step of the sharded model — all-reduced parameter gradients —
                          against the same step of the unsharded model on rank 0 (same seed: the
                          sharded path draws the single-GPU run's dropout masks).  This is the
                          BACKWARD exchange's check: a wrong static gradient halo would leave the
                          forward `loss_check` untouched and produce a plausible timing line.
  * `link_rate`           one large point-to-point transfer, timed: the per-link rate that decides
                          between the exchange forms.

Every function is collective (all ranks call it alike) and returns a small dict for the bench line.
"""
import time

import torch
import torch.distributed as dist


def _normwise(a, b):
    """max|a - b| / max|b| as a python float (0 for empty tensors)."""
    if b.numel() == 0:
        return 0.0
    scale = float(b.detach().abs().max())
    err = float((a.detach().double() - b.detach().double()).abs().max())
    if err != err:                         # NaN anywhere: never a pass
        return float("inf")
    return err / scale if scale > 0 else (0.0 if err == 0 else float("inf"))


def _all_max(value, device, group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def overlap_selftest(sg, operands, tol=1e-5):
    """Pipelined vs unpipelined dense halo exchange on the same operands.

    `operands`: a list of [n_local, F] tensors (a few DIFFERENT ones: stale halo rows of an
    earlier, identical exchange must not pass for fresh ones).  Leaves `sg.overlap` True if every
    rank agrees on every operand within `tol` (normwise; the two forms differ by fp32 summation
    order only), else switches the pipelining off on every rank.  Returns
    {"max_err", "agrees", "overlap_in_use"}."""
    wanted = bool(sg.overlap)
    if not wanted or sg.exchange_mode != "halo" or sg.world == 1:
        return {"max_err": None, "agrees": None, "overlap_in_use": bool(sg.overlap),
                "note": "pipelined exchange not in use"}
    worst = 0.0
    for t in operands:
        sg.overlap = True
        piped = sg.product(t)
        sg.overlap = False
        plain = sg.product(t)
        worst = max(worst, _normwise(piped.float(), plain.float()))
    worst = _all_max(worst, operands[0].device, sg.group)
    ok = worst <= tol
    sg.overlap = bool(ok)
    out = {"max_err": worst, "agrees": bool(ok), "overlap_in_use": bool(ok), "tolerance": tol}
    if not ok:
        out["note"] = ("the pipelined exchange (transfers under the own-rows product) did NOT reproduce "
                       "the unpipelined result on this backend: fell back to exchange-then-product")
    return out


def sharded_grad_check(params, sharded_step, reference_step, group=None, tol=5e-5):
    """All-reduced gradients of one sharded training step vs the unsharded step on rank 0.

    params          the (replicated) model parameters
    sharded_step    callable(): zeroes the gradients, runs one sharded forward/backward and the
                    gradient all-reduce — collective, every rank calls it
    reference_step  callable() -> list of gradient tensors in `params` order, or None: the SAME step
                    on the whole graph with the SAME parameters and seed — called on rank 0 only
                    (it may build the single-GPU graph; the other ranks wait at the broadcast)
    Returns {"max_err", "per_param", "ok", "tolerance"}; max_err is the largest normwise error over
    the parameters and over the ranks (every rank compares its own all-reduced copy)."""
    rank = dist.get_rank(group)
    sharded_step()
    got = [p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    dev = got[0].device
    flat = torch.zeros(sum(g.numel() for g in got) + 1, dtype=torch.float32, device=dev)
    if rank == 0:
        ref = reference_step()
        if ref is not None:
            flat[:-1].copy_(torch.cat([g.detach().reshape(-1).float() for g in ref]))
            flat[-1] = 1.0
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(flat, src=src, group=group)
    if float(flat[-1]) != 1.0:
        return {"max_err": None, "ok": None, "note": "rank 0 could not run the unsharded reference step"}
    per, off = [], 0
    for g in got:
        r = flat[off:off + g.numel()].view_as(g)
        off += g.numel()
        per.append(_normwise(g.float(), r))
    worst = _all_max(max(per), dev, group)
    per = [_all_max(v, dev, group) for v in per]
    return {"max_err": worst, "per_param": per, "ok": bool(worst <= tol), "tolerance": tol}


def link_rate(device, nbytes, group=None, backend_is_nccl=True):
    """One point-to-point transfer of `nbytes` from rank 0 to rank 1 (after a small warm-up that
    opens the connection), timed on the receiver: GB/s of ONE link in ONE direction.  Collective
    (ranks >= 2 only take part in the closing barrier).  Returns {"gb_per_s", "bytes", "ms"}."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world < 2:
        return {"gb_per_s": None, "note": "one rank"}
    n = max(1, int(nbytes) // 4)
    buf = torch.empty(n, dtype=torch.float32, device=device) if rank < 2 else None
    ms = 0.0

    def once(t):
        if rank == 0:
            dist.send(t, dst=dist.get_global_rank(group, 1) if group is not None else 1, group=group)
        elif rank == 1:
            dist.recv(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if rank < 2:
        buf[: min(n, 1 << 20)].fill_(1.0)
        once(buf[: min(n, 1 << 20)])            # warm-up: connection set-up is not link time
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        once(buf)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        ms = (time.perf_counter() - t0) * 1e3
    ms = _all_max(ms, device, group)
    return {"gb_per_s": (n * 4 / (ms * 1e-3) / 1e9) if ms > 0 else None, "bytes": n * 4, "ms": ms,
            "what": "one dist.send / dist.recv pair rank 0 -> rank 1, host-timed around a device "
                    "synchronisation on both ends (max of the two), after a 4 MB warm-up transfer"}


def forward_exchange_ab(sg, h_local, weight, bias, modes, reps=3, log_softmax=True):
    """Pre-timed A/B of the hidden layer's FORWARD exchange forms on the node the job runs on:
    for every mode in `modes` — "halo" (pipelined per sg.overlap), "allgather" (grouped
    point-to-point), "rccl-allgather" (the collective), "compress-hidden" (bitmask + values, the
    weight applied on arrival) — `reps` evaluations of the layer  epilogue(Â_r·exchange(h·W) + b)
    after one warm-up, each bracketed by a barrier; the figure of a mode is the slowest rank's
    mean.  Leaves the graph in the FASTEST mode (the same on every rank: the times are
    all-reduced) and returns {"ms": {mode: ms}, "chosen": mode}.  Collective."""
    from .spmm import _dense_forward
    dev = h_local.device
    cuda = dev.type == "cuda"
    kw = {"log_softmax": True} if log_softmax else {}

    def run(mode):
        if mode == "compress-hidden":
            sg.set_forward_exchange("halo")
            return sg.product_hidden(h_local, weight, bias=bias, **kw)
        sg.set_forward_exchange(mode)
        return sg.product(_dense_forward(h_local, weight), bias=bias, **kw)
    ms, ref, agree = {}, None, {}
    for mode in modes:
        out = run(mode)                                    # warm-up (builds the padded block / split once)
        if ref is None:
            ref = out.float()
        else:                                              # every form must compute the same layer
            agree[mode] = _all_max(_normwise(out.float(), ref), dev, sg.group)
        del out
        if cuda:
            torch.cuda.synchronize(dev)
        dist.barrier(group=sg.group)
        t0 = time.perf_counter()
        for _ in range(reps):
            run(mode)
        if cuda:
            torch.cuda.synchronize(dev)
        ms[mode] = _all_max((time.perf_counter() - t0) / reps * 1e3, dev, sg.group)
    chosen = min(ms, key=ms.get)
    compress_before = sg.compress_hidden
    sg.compress_hidden = chosen == "compress-hidden"
    sg.set_forward_exchange("halo" if chosen == "compress-hidden" else chosen)
    return {"ms": {k: round(v, 4) for k, v in ms.items()}, "chosen": chosen,
            "max_err_vs_first_mode": {k: v for k, v in agree.items()},
            "compress_hidden_before": bool(compress_before)}
