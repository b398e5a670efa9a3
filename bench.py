#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: "SpMM GEdge/s + fwd+bwd ms/epoch, 10M-node
synthetic CSR, feat_dim=256".

    python bench.py --gpus N --steps K --warmup W          (N > 1: under torch.distributed.run, or
                                                            bare — it then starts its N ranks itself)

A *step* is one full-graph training epoch of the upstream 2-layer GCN (256 -> 256 -> 256) on the
synthetic R-MAT graph of config C4: 2 GEMM + 2 SpMM (over all rows) + bias + ReLU +
dropout + log-softmax + NLL forward; backward = the transpose SpMM of layer 2 (restricted to the
rows that can be non-zero) + 3 GEMMs (layer 1 is evaluated as (A·X)·W1, so its weight gradient
reuses this step's forward product A·X and needs no sparse product); Adam step.  Inputs are
generated on the device and are resident in HBM before the timed region starts.

The timed workload is STATIONARY: the parameters and the Adam state are snapshotted after the
warm-up epochs and restored (a 0.8 MB device copy, inside the timed region) at the start of every
timed epoch, so every timed epoch is "epoch W+1" and `ms_per_step` does not depend on --steps.
(Free-running training on the bench's random labels drives the hidden layer of hub vertices dead
within ~25 epochs, which makes the layer-1 backward product 4x cheaper — a property of the random
labels, not of the kernels; the free-running figures are reported beside the graded one.)

Printed JSON (rank 0, one line):
  value        = fwd SpMM throughput, nnz(A_hat) / mean duration of the forward `gcn_spmm_csr`
                 launches inside the timed region, in GEdge/s.  At N > 1: all ranks' stored
                 entries / the slowest rank's mean forward window, where the window INCLUDES the
                 exchange step the product depends on (halo P2P or all-gather; for layer 1 in
                 halo mode the GEMM of the held feature halo rows + the local product, see
                 pygcn_amd/sharded.py); `spmm_only_gedges` is the same without the exchange.
  ms_per_step  = fwd+bwd ms/epoch (wall, max over ranks)
  roofline     = algorithmic bytes of one forward SpMM launch / its mean duration vs 8 TB/s HBM
  cpu_baseline = the reference's literal call `torch.spmm(adj, support)` (pygcn/layers.py:34) on
                 the COO layout the reference builds (pygcn/utils.py:407-414), timed on the box's
                 host cores on a row block of the same graph; the CSR/MKL form of the same call
                 and the oracle's OpenMP port are reported beside it.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0    # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

CONFIGS = {
    # name: (nodes, sampled directed edges, feat, dtype)
    "c4": (10_000_000, 100_000_000, 256, "f32"),   # the configuration the metric is quoted on
    "c3": (1_000_000, 10_000_000, 256, "f32"),
    "c5": (50_000_000, 1_000_000_000, 128, "bf16"),  # BASELINE config 5 (fits one GPU: 120 GB)
    "tiny": (50_000, 500_000, 256, "f32"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--edges", type=int, default=0)
    ap.add_argument("--feat", type=int, default=0)
    ap.add_argument("--dtype", default="", choices=["", "f32", "bf16"],
                    help="storage type of features / activations / parameters (values and "
                         "accumulation stay fp32); default: the config's")
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--reference-call", action="store_true",
                    help="profiling aid: the timed epoch uses upstream's unchanged lines "
                         "(model(features, adj); nll_loss(output[idx_train], ...)) instead of rows=")
    ap.add_argument("--dense-loss", action="store_true",
                    help="the timed epoch takes the NLL over ALL vertices (every gradient row is "
                         "non-zero: the fork's live case, a reduction over all nodes — reference "
                         "pygcn/train.py:151-155) instead of upstream's idx_train share")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed side measurements (free-running and dense-loss epochs)")
    ap.add_argument("--spmm-only", action="store_true",
                    help="time only forward SpMM launches (profiling aid; not the graded mode)")
    ap.add_argument("--item-cost", type=int, default=0)
    ap.add_argument("--long-thresh", type=int, default=0)
    ap.add_argument("--exchange", default="auto", choices=["auto", "halo", "allgather", "rccl-allgather"],
                    help="multi-GPU exchange step of the hidden layer's forward product: 'halo' = the rows a "
                         "rank references only (grouped point-to-point, pipelined); 'allgather' = every row, "
                         "one grouped point-to-point round; 'rccl-allgather' = every row through RCCL's "
                         "all-gather collective (dist.all_gather_into_tensor); 'auto' (default) = a "
                         "pre-timed A/B of all of them (and --compress-hidden) before the timed region, "
                         "the fastest is timed (`exchange_ab_ms`)")
    ap.add_argument("--gemm-scheme", default="bf16x3", choices=["bf16x3", "h2", "exact"],
                    help="the fp32 256x256 GEMMs of the epoch: 'bf16x3' (default) = three bf16 parts, six "
                         "MFMAs per product, fp32-EQUIVALENT (24-bit significand); 'h2' = two scaled fp16 "
                         "parts (22 bits, half the matrix work); 'exact' = hipBLASLt fp32")
    ap.add_argument("--graph", default="rmat", choices=["rmat", "uniform"],
                    help="'uniform' (single GPU): the cache-hostile graph — every vertex draws 10 neighbours "
                         "uniformly (+ I, row-normalized): no hubs, every gather an HBM miss")
    ap.add_argument("--no-selfcheck", action="store_true",
                    help="multi-GPU: skip the pre-timed self-validation (overlap self-test, sharded "
                         "gradient check against the single-GPU step, link rate)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="multi-GPU: exchange, then one product (no source-block pipelining)")
    ap.add_argument("--compress-hidden", action="store_true",
                    help="multi-GPU A/B switch: layer 2's input rows (>= 75 %% zeros after ReLU + "
                         "dropout) travel as bitmask + non-zero values and meet W2 on arrival, instead "
                         "of dense rows of h1 x W2 (pygcn_amd/sharded.py: product_hidden)")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)   # launcher test
    ap.add_argument("--rehearsal", action="store_true",
                    help="N>1 code-path rehearsal on ONE GPU: all ranks share cuda:0, collectives "
                         "go over gloo staged through the host. Not a measurement.")
    return ap.parse_args()


def algorithmic_bytes(nnz, n_rows, F, s=4, rowptr_bytes=4):
    """gather model, no cache-reuse credit:
    nnz*(F*s + 4 + 4) + n_rows*(F*s + p)."""
    return nnz * (F * s + 8) + n_rows * (F * s + rowptr_bytes)


def host_cpu_info():
    """(model string, physical cores, logical cpus) from /proc/cpuinfo — what `lscpu` prints."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) or logical), logical


def cpu_baseline_child(path):
    """Runs in a CHILD process (a crash in a CPU library must not lose the GPU result): times
    the reference-side CPU products on the sample saved by the parent; prints JSON lines (the
    last complete one wins).  Legs, SURVEY §8(d): the reference's call `torch.spmm(adj, support)`
    (pygcn/layers.py:34) forward, and forward + backward (`out.backward(G)`: autograd's
    `adj.t() @ grad` of pygcn/train.py:157), on (i) the COO int64 layout the reference builds
    (pygcn/utils.py:407-414) and (ii) CSR (MKL — the best CPU case); plus the oracle's OpenMP port."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    z = np.load(path)
    rp, c, v, n, F = z["rowptr"], z["col"], z["val"], int(z["n"]), int(z["F"])
    budget_rows, nnz_s = len(rp) - 1, int(rp[-1])
    whole = bool(z["whole"])
    model, phys, logical = host_cpu_info()
    threads = max(1, min(phys, len(os.sched_getaffinity(0))))
    torch.set_num_threads(threads)
    # dense operand: a 65 536-row random block tiled to n rows (values do not matter for timing;
    # torch.randn of 2.56e9 elements alone would take longer than the whole baseline)
    blk = torch.randn(65536, F, generator=torch.Generator().manual_seed(44))
    Bt = blk.repeat((n + 65535) // 65536, 1)[:n].contiguous()
    Gt = blk.repeat((budget_rows + 65535) // 65536, 1)[:budget_rows].contiguous()

    def best(fn, reps=3, budget_s=25.0):
        """min / median over up to `reps` runs after one warm-up, stopping early once the leg has
        used its time budget (the whole baseline must stay within a few minutes)."""
        t0 = time.perf_counter()
        fn()
        ts = []
        for _ in range(reps):
            t1 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t1)
            if time.perf_counter() - t0 > budget_s:
                break
        return min(ts), float(np.median(ts)), len(ts)

    sample = ((f"the WHOLE normalized adjacency ({budget_rows} rows, {nnz_s} stored entries)" if whole else
               f"first {budget_rows} rows ({nnz_s} stored entries: a 1/{max(1, round(n / budget_rows))} row "
               f"block) of the same normalized adjacency") +
              f" x full dense operand [{n},{F}] fp32 (a 65 536-row random block tiled to {n} rows — "
              f"values do not affect the timing); backward seed G [{budget_rows},{F}] likewise; min of up "
              f"to 3 runs after 1 warm-up per leg")
    # (i) the reference's literal call on its own layout: torch.spmm(COO int64, dense)
    rows = np.repeat(np.arange(budget_rows, dtype=np.int64), np.diff(rp))
    coo = torch.sparse_coo_tensor(np.vstack([rows, c.astype(np.int64)]), v, (budget_rows, n))
    t_coo, t_coo_med, _ = best(lambda: torch.spmm(coo, Bt))
    out = {"value": round(nnz_s / t_coo / 1e9, 5), "unit": "GEdge/s", "cores": threads,
           "kind": "reference",
           "what": "torch.spmm(adj, dense) — the reference's call (pygcn/layers.py:34) on the COO "
                   "int64/fp32 layout it builds (pygcn/utils.py:407-414); the arithmetic lives in "
                   "PyTorch, the reference has no other implementation of it",
           "sample": sample, "median_gedges": round(nnz_s / t_coo_med / 1e9, 5),
           "cpu_model": model, "physical_cores": phys, "logical_cpus": logical,
           "torch_threads": torch.get_num_threads(), "torch_version": torch.__version__}
    print(json.dumps(out), flush=True)

    def fwd_bwd(A, parts_b, parts_g):
        for bp, gp in zip(parts_b, parts_g):
            bp.grad = None
            torch.spmm(A, bp).backward(gp)
    # forward + backward through autograd, as `loss.backward()` runs it (pygcn/train.py:157):
    # grad_support = adj.t() @ grad_output.  bwd_gedges = entries / (t_fwd+bwd - t_fwd).
    Bg = Bt.clone().requires_grad_(True)
    t_fb, _, k = best(lambda: fwd_bwd(coo, [Bg], [Gt]), reps=2, budget_s=40.0)
    out["fwd_plus_bwd_s"] = round(t_fb, 4)
    out["fwd_s"] = round(t_coo, 4)
    out["bwd_gedges"] = round(nnz_s / max(t_fb - t_coo, 1e-9) / 1e9, 5)
    out["fwd_plus_bwd_gedges"] = round(2 * nnz_s / t_fb / 1e9, 5)
    out["bwd_note"] = ("out = torch.spmm(adj_coo, B.requires_grad_()); out.backward(G) — autograd's "
                       f"transpose product (pygcn/train.py:157); min of {k} run(s) after 1 warm-up; "
                       "bwd_gedges = entries / (t(fwd+bwd) - t(fwd))")
    del Bg
    print(json.dumps(out), flush=True)
    # (ii) the oracle's OpenMP CSR port (test infrastructure; timed here as the "port" baseline)
    import gcn_oracle
    B = Bt.numpy()
    t_port, _, _ = best(lambda: gcn_oracle.spmm_csr(rp, c, v, B))
    out["oracle_port_csr_gedges"] = round(nnz_s / t_port / 1e9, 5)
    out["oracle_port_threads"] = gcn_oracle.lib().oracle_num_threads()
    print(json.dumps(out), flush=True)
    # (iii) the same torch call on CSR (MKL sparse BLAS — the best CPU case).  MKL's 32-bit
    # interface crashes when a dense operand has >= 2^31 elements (seen: SIGSEGV at n*F =
    # 2.56e9), so the operand is multiplied in column panels of < 2^31 elements each
    csr = torch.sparse_csr_tensor(torch.from_numpy(rp), torch.from_numpy(c.astype(np.int64)),
                                  torch.from_numpy(v), size=(budget_rows, n))
    panels = 1
    while n * (F // panels) >= 2 ** 31 and panels < F:
        panels *= 2
    w = F // panels
    parts = [Bt[:, i * w:(i + 1) * w].contiguous() for i in range(panels)]
    t_csr, _, _ = best(lambda: [torch.spmm(csr, p) for p in parts], reps=2)
    out["torch_spmm_csr_gedges"] = round(nnz_s / t_csr / 1e9, 5)
    out["torch_spmm_csr_note"] = f"MKL CSR, dense operand in {panels} column panel(s) of {w}"
    print(json.dumps(out), flush=True)
    parts_b = [p.clone().requires_grad_(True) for p in parts]
    parts_g = [Gt[:, i * w:(i + 1) * w].contiguous() for i in range(panels)]
    t_csr_fb, _, _ = best(lambda: fwd_bwd(csr, parts_b, parts_g), reps=2, budget_s=40.0)
    out["torch_spmm_csr_bwd_gedges"] = round(nnz_s / max(t_csr_fb - t_csr, 1e-9) / 1e9, 5)
    out["torch_spmm_csr_fwd_plus_bwd_s"] = round(t_csr_fb, 4)
    print(json.dumps(out), flush=True)


def cpu_baseline(rowptr, col, val, n, F, budget_rows):
    """Reference-side CPU product on a bounded row block of the same graph (rank 0, N=1)."""
    import subprocess
    import tempfile
    rp = rowptr[:budget_rows + 1].cpu().numpy().astype(np.int64)
    nnz_s = int(rp[-1])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sample.npz")
        np.savez(path, rowptr=rp, col=col[:nnz_s].cpu().numpy(), val=val[:nnz_s].cpu().numpy(),
                 n=n, F=F, whole=bool(budget_rows >= rowptr.numel() - 1))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", path],
                           capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        return {"error": f"cpu baseline child failed rc={r.returncode}: {r.stderr[-300:]}"}
    out = json.loads(lines[-1])
    if r.returncode != 0:
        out["note"] = f"a later leg crashed in the child (rc={r.returncode}): {r.stderr[-200:]}"
    return out


def measured_traffic(config, dt):
    """HBM-side bytes per forward launch from the committed PMC passes — only if they were taken
    on the kernel source that is running now (the file is stamped with the source hash)."""
    tfile = os.path.join(ROOT, "profiles", f"traffic_{config}.json")
    if not os.path.exists(tfile):
        return None, "no PMC pass committed for this configuration"
    try:
        t = json.load(open(tfile))
        if t.get("dtype", "f32") != dt:
            return None, "the committed PMC pass was taken at another storage type"
        src = open(os.path.join(ROOT, "pygcn_amd", "csrc", "gcn_spmm.hip"), "rb").read()
        if t.get("kernel_source_sha256") != hashlib.sha256(src).hexdigest():
            return None, ("stale: profiles/traffic_%s.json was taken on another version of "
                          "gcn_spmm.hip (commit %s)" % (config, t.get("commit", "?")))
        return t.get("hbm_bytes_per_launch"), ("PMC FETCH_SIZE (x%.4f calibrated) + WRITE_SIZE, "
                                               "separate passes, commit %s"
                                               % (t.get("fetch_size_calibration_factor", 0),
                                                  t.get("commit", "?")))
    except Exception as ex:          # a malformed side file must not lose the bench line
        return None, f"unreadable: {ex!r}"


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1 at a free port) BEFORE this
    process has touched the GPU, let rank 0's JSON line through on the inherited stdout, and
    return the launcher's exit status (non-zero as soon as one rank dies: torchrun then stops
    the others).  Never exec-replaces the process."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    def on_term(signum, frame):      # the driver's timeout: leave through the except below
        raise SystemExit(128 + signum)
    signal.signal(signal.SIGTERM, on_term)
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait()
    except BaseException:            # Ctrl-C / SIGTERM on the parent: take the whole group down
        try:
            os.killpg(proc.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        proc.wait()
        raise


def main():
    args = parse()
    if args.spmm_only and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("--spmm-only is a single-GPU profiling aid")
    if args.cpu_baseline_child:
        cpu_baseline_child(args.cpu_baseline_child)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))      # (no GPU call has happened in this process)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    if world > 1 and args.rehearsal:
        dist.init_process_group("gloo")
        from tools.rehearsal import install_host_staging
        install_host_staging()
    elif world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        # (a mismatched or lost collective must end the run, not hang it until the driver's clock does)
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=8))

    if args.fail_rank == rank:       # (tests/test_bench_launch.py: a rank that dies after the rendezvous)
        os._exit(3)
    from pygcn_amd import GCN, CSRGraph, _native
    from pygcn_amd import spmm as spmm_mod
    from pygcn_amd.functional import nll_loss      # (= F.nll_loss, mean reduction; gather / scatter)
    from pygcn_amd.utils import rmat_graph, uniform_graph
    _native.lib()
    spmm_mod.set_gemm_scheme(args.gemm_scheme)
    if args.graph == "uniform" and world > 1:
        raise SystemExit("--graph uniform is a single-GPU measurement")

    n, e, feat, dt = CONFIGS[args.config]
    n, e, feat, dt = args.nodes or n, args.edges or e, args.feat or feat, args.dtype or dt
    tdtype = torch.bfloat16 if dt == "bf16" else torch.float32
    esize = 2 if dt == "bf16" else 4

    # ---------------------------------------------------------------- inputs (HBM resident)
    t0 = time.perf_counter()
    gen = torch.Generator(device=dev)
    gen.manual_seed(44)
    kw = dict(item_cost=args.item_cost, long_thresh=args.long_thresh)
    peak_setup = None

    def global_inputs(lo, hi):
        """Rows [lo, hi) of the feature matrix and label vector of the WHOLE graph (seeds 44 / 45):
        every world size trains on the same problem, so losses are comparable across N.  A rank
        draws the full stream and keeps its slice (the generator has no skip-ahead by rows); the
        temporary is released before the timed region."""
        g = torch.Generator(device=dev)
        g.manual_seed(44)
        full = torch.randn(n, feat, generator=g, device=dev)
        xs = full[lo:hi].to(tdtype).clone() if (lo, hi) != (0, n) else full.to(tdtype)
        del full
        lab = torch.randint(0, feat, (n,), device=dev,
                            generator=torch.Generator(device=dev).manual_seed(45))[lo:hi].clone()
        torch.cuda.empty_cache()
        return xs, lab
    if world == 1:
        if args.graph == "uniform":
            rowptr, col, val = uniform_graph(n, max(1, e // n), seed=46, device=dev)
        else:
            rowptr, col, val = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
        nnz = int(col.numel())
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        graph = CSRGraph(rowptr, col, val, (n, n), **kw)
        graph.plan()
        graph.t().plan()
        x, labels = global_inputs(0, n)
        torch.manual_seed(42)
        model = GCN(feat, feat, feat, dropout=args.dropout).to(dev).to(tdtype)
        adj = graph
        n_local, nnz_local = n, nnz
        r0 = 0
        fwd_model = model
    else:
        # shard-local construction: a rank generates only its own rows of the SAME graph the
        # single-GPU run builds; its rows of the transpose arrive as triplets from their holders.
        # Per-rank memory is O(nnz / N) (+ one generator chunk), never the whole matrix.
        from pygcn_amd.sharded import ShardedGraph, ShardedGCN
        torch.cuda.reset_peak_memory_stats(dev)
        # (always built in halo mode — the constant-input halo and the static gradient halo of the
        #  backward pass need it — the hidden layer's FORWARD exchange form follows --exchange:
        #  named explicitly, or chosen by the pre-timed A/B below)
        adj = ShardedGraph.from_rmat(n, e, rank, world, dev, seed=42, perm_seed=43,
                                     exchange="halo",
                                     overlap=not args.no_overlap,
                                     compress_hidden=args.compress_hidden, **kw)
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        peak_setup = torch.cuda.max_memory_allocated(dev)
        n_local, nnz_local = adj.n_local, adj.nnz_local
        r0 = adj.r0
        # the rank's rows of the SAME feature matrix / labels the single-GPU run draws
        x, labels = global_inputs(adj.r0, adj.r1)
        torch.manual_seed(42)
        model = GCN(feat, feat, feat, dropout=args.dropout).to(dev).to(tdtype)
        fwd_model = ShardedGCN(model, adj)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    # upstream epoch: loss on the labelled training nodes only (train.py:140-157 comments,
    # utils.py:370 idx_train = range(140) of 2708 -> the same 5.17 % head of the (seeded-permuted)
    # vertex list here)
    # (sharded: the global list cut by row block — a rank past the head holds no labelled vertex)
    n_train_global = max(1, int(n * 140 / 2708))
    n_train = max(0, min(n_train_global, r0 + n_local) - r0)
    idx_train = torch.arange(n_train, device=dev)
    torch.cuda.synchronize()

    labels_train = labels[idx_train]

    def epoch(dense_loss=False, reference_call=False):
        model.train()
        opt.zero_grad(set_to_none=True)
        if reference_call:      # upstream's literal lines (train.py:140-157): no `rows=` hint
            out = fwd_model(x, adj)
            loss = F.nll_loss(out[idx_train], labels_train)
        elif world > 1 and dense_loss:
            out = fwd_model(x, adj)
            loss = fwd_model.nll_loss(out.float(), labels, None)
        elif world > 1 and sharded_rows_note is not None:
            loss = fwd_model.nll_loss(fwd_model(x, adj).float(), labels, idx_train)
        elif world > 1:
            # every rank names the rows of its block the loss reads: one autograd node per rank,
            # a static halo of gradient rows (pygcn_amd/sharded_fused.py)
            loss = fwd_model.nll_loss(fwd_model(x, adj, rows=rows_handle).float(), labels_train)
        elif dense_loss:
            # a loss over ALL vertices (the fork's live loss reduces over every node, reference
            # pygcn/train.py:151-155): pygcn_amd.functional.nll_loss = F.nll_loss (mean), whose
            # gradient reaches the model's backward pass in structural form (one non-zero per row)
            loss = nll_loss(fwd_model(x, adj), labels)
        else:
            # upstream: F.nll_loss(output[idx_train], labels[idx_train]) (train.py:153) — the model is
            # told which rows the loss reads, so the backward pass runs on the rows that can be
            # non-zero (pygcn_amd/fused.py); the forward pass is the full one
            loss = nll_loss(fwd_model(x, adj, rows=idx_train).float(), labels_train)
        loss.backward()
        if world > 1:
            fwd_model.allreduce_grads()
        opt.step()

    snapshot = None

    def take_snapshot():
        st = opt.state_dict()["state"]
        return ([p.detach().clone() for p in model.parameters()],
                {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                 for k, v in st.items()})

    def restore_snapshot():
        params, state = snapshot
        with torch.no_grad():
            for p, q in zip(model.parameters(), params):
                p.copy_(q)
            live = opt.state_dict()["state"]
            for k, v in state.items():
                for kk, vv in v.items():
                    if torch.is_tensor(vv):
                        live[k][kk].copy_(vv)

    def step():
        if args.spmm_only:
            with torch.no_grad():
                spmm_mod.spmm_csr(graph, x)
            return
        if snapshot is not None:
            restore_snapshot()
        epoch(dense_loss=args.dense_loss, reference_call=args.reference_call)

    def timed(k, fn):
        """K calls of fn bracketed by barrier + synchronize; returns (wall seconds, per-call ms from
        HIP events recorded at the call boundaries — no synchronisation inside)."""
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(k):
            fn()
            marks[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        return wall, [marks[i].elapsed_time(marks[i + 1]) for i in range(k)]

    # structural setup that belongs to the (graph, idx_train) pair, not to an epoch: the row sets /
    # the transpose block (and, sharded, the static gradient halo) — built here, collectively,
    # whatever --warmup says
    sharded_rows_note = None
    rows_handle = idx_train
    if not args.spmm_only and not args.reference_call and not args.dense_loss:
        try:
            if world > 1:      # collective, once: the handle makes the per-epoch call lookup-free
                rows_handle = fwd_model.prepare_rows(idx_train)
            with torch.no_grad():
                fwd_model(x, adj, rows=rows_handle)
            torch.cuda.synchronize()
        except Exception as ex:
            # (N > 1 only: a deterministic failure of the one-node path's setup raises on every
            #  rank alike — fall back to the layer-by-layer sharded path rather than lose the run)
            if world == 1:
                raise
            sharded_rows_note = f"one-node path unavailable, layer-by-layer path timed instead: {ex!r}"
    # ---------------------------------------------------------------- N > 1: the run validates itself
    # (outside the timed region; pygcn_amd/selfcheck.py).  Nobody who builds this code sees more
    # than one GPU: the first run between GPUs checks its own exchanges before it times them.
    selfcheck = {}
    if world > 1 and not args.spmm_only:
        from pygcn_amd import selfcheck as sc
        selfcheck["ranks"] = dist.get_world_size()
        selfcheck["backend"] = dist.get_backend()
        names = [None] * world
        dist.all_gather_object(names, f"{torch.cuda.get_device_name(dev)} (cuda:{local_rank})")
        selfcheck["devices"] = names
        tol_fwd = 1e-5 if dt == "f32" else 2.0 ** -6
        if not args.no_selfcheck:
            try:
                # (the size must be the same number on both ends: max_rows, not this rank's n_local)
                selfcheck["link_rate"] = sc.link_rate(dev, min(1_280_000_000, max(1, adj.max_rows) * feat * esize))
            except Exception as ex:
                selfcheck["link_rate"] = {"error": repr(ex)}
            # (b) the pipelined exchange against the unpipelined one, on three different operands
            ops = [(x.float() * (k + 1)).to(tdtype) for k in range(3)]
            selfcheck["overlap_selftest"] = sc.overlap_selftest(adj, ops, tol=tol_fwd)
            del ops
        # (3) the forward exchange forms, timed on THIS node: halo P2P, P2P all-gather, RCCL's
        # all-gather collective, compressed hidden rows; the fastest is what the epoch is timed with
        modes = ["halo", "allgather", "rccl-allgather", "compress-hidden"]
        if args.exchange != "auto":
            adj.set_forward_exchange(args.exchange)
            selfcheck["exchange_chosen"] = args.exchange + (" + compress-hidden" if args.compress_hidden else "")
        elif feat % 32:
            selfcheck["exchange_chosen"] = "halo (A/B skipped: width not a multiple of 32)"
        else:
            model.train()
            with torch.no_grad():
                torch.manual_seed(4242)
                h1 = model.gc1(x, adj, relu=True, dropout=args.dropout)
                ab = sc.forward_exchange_ab(adj, h1, model.gc2.weight, model.gc2.bias, modes,
                                            reps=3, log_softmax=True)
            del h1
            selfcheck["exchange_ab_ms"] = ab["ms"]
            selfcheck["exchange_ab_max_err_vs_halo"] = ab["max_err_vs_first_mode"]
            selfcheck["exchange_chosen"] = ab["chosen"]
            selfcheck["exchange_ab_note"] = (
                "layer-2 forward (GEMM + exchange + local product + log_softmax) per exchange form, 3 "
                "evaluations after a warm-up, slowest rank's mean; the timed epoch uses the fastest")
        if not args.no_selfcheck and sharded_rows_note is None and not args.dense_loss \
                and not args.reference_call:
            # (a) all-reduced gradients of one sharded step vs the same step on ONE GPU (rank 0
            # builds the whole graph next to its shard): the check of the BACKWARD exchange
            n_tg = n_train_global

            def sharded_once():
                torch.manual_seed(777)
                model.train()
                model.zero_grad(set_to_none=True)
                loss = fwd_model.nll_loss(fwd_model(x, adj, rows=rows_handle).float(), labels_train)
                loss.backward()
                fwd_model.allreduce_grads()

            def reference_once():
                try:
                    rp_, c_, v_ = rmat_graph(n, e, seed=42, perm_seed=43, device=dev)
                    g1 = CSRGraph(rp_, c_, v_, (n, n), **kw)
                    x1, lab1 = global_inputs(0, n)
                    idx1 = torch.arange(n_tg, device=dev)
                    torch.manual_seed(777)
                    model.train()
                    model.zero_grad(set_to_none=True)
                    nll_loss(model(x1, g1, rows=idx1).float(), lab1[idx1]).backward()
                    return [p.grad.detach().clone() for p in model.parameters()]
                except Exception as ex:          # (e.g. the whole graph does not fit next to the shard)
                    selfcheck["sharded_grad_check_error"] = repr(ex)
                    return None
            res = sc.sharded_grad_check(list(model.parameters()), sharded_once, reference_once,
                                        tol=5e-5 if dt == "f32" else 2.0 ** -4)
            model.zero_grad(set_to_none=True)
            torch.cuda.empty_cache()
            selfcheck["sharded_grad_check"] = res
            selfcheck["sharded_grad_check_note"] = (
                "max normwise error of the all-reduced parameter gradients of one sharded training step "
                "(dropout on, seed 777) against the same step of the unsharded model on rank 0 — same "
                "graph, features, labels, parameters, dropout masks; bench.py exits non-zero above the "
                "tolerance")
    # the same number at every world size (same graph, features, labels, initial parameters; no
    # dropout in eval mode): a sharded run that computes something else shows up here
    loss_check = None
    if not args.spmm_only:
        model.eval()
        with torch.no_grad():
            out0 = fwd_model(x, adj)
            if world > 1:
                loss_check = fwd_model.global_loss(fwd_model.nll_loss(out0.float(), labels, idx_train))
            else:
                loss_check = float(F.nll_loss(out0[idx_train].float(), labels_train))
        del out0
        model.train()
    for _ in range(args.warmup):
        step()
    if not args.spmm_only and args.warmup > 0:
        torch.cuda.synchronize()
        snapshot = take_snapshot()
    records = []
    spmm_mod.set_timing_records(records)
    if world > 1:
        adj.timing = []
    elapsed, per_step = timed(args.steps, step)
    spmm_mod.set_timing_records(None)

    local_ms = {"fwd": 0.0, "bwd": 0.0}
    if world > 1:   # launch window = exchange + local SpMM (the exchange step is part of it)
        for tag, a, b, _ in records:
            local_ms["fwd" if tag.startswith("fwd") else "bwd"] += a.elapsed_time(b)
        wait_ms = [a.elapsed_time(b) for tag, a, b in adj.timing if tag == "fwd_wait"]
        records = [(tag, a, b, None) for tag, a, b in adj.timing]
        adj.timing = None
    fwd_ms = [a.elapsed_time(b) for tag, a, b, _ in records if tag == "fwd"]
    bwd_ms = [a.elapsed_time(b) for tag, a, b, _ in records if tag.startswith("bwd")]
    t_fwd = float(np.mean(fwd_ms)) if fwd_ms else float("nan")
    t_bwd = float(np.mean(bwd_ms)) if bwd_ms else float("nan")
    l2 = [a.elapsed_time(b) for tag, a, b, _ in records if tag == "bwd_l2"]
    l1 = [a.elapsed_time(b) for tag, a, b, _ in records if tag == "bwd_l1"]
    plain = [a.elapsed_time(b) for tag, a, b, _ in records if tag == "bwd"]
    if l2 or l1:      # one-node path: launches are tagged by layer (layer 1 may need none at all)
        bwd_l2 = float(np.mean(l2)) if l2 else 0.0
        bwd_l1 = float(np.sum(l1)) / max(1, len(l2)) if l1 else 0.0
    elif plain and len(plain) == args.steps:   # sharded, layer 1 reassociated: layer 2 only
        bwd_l2, bwd_l1 = float(np.mean(plain)), 0.0
    else:             # layer-by-layer / sharded path: layer 2 first, then layer 1, every epoch
        bwd_l2 = float(np.mean(plain[0::2])) if len(plain) >= 2 else float("nan")
        bwd_l1 = float(np.mean(plain[1::2])) if len(plain) >= 2 else float("nan")
    n_fwd = max(1, len(fwd_ms))
    t_fwd_local = local_ms["fwd"] / n_fwd if world > 1 else t_fwd
    recv = [0, 0]
    if world > 1:
        recv = [adj.last_recv_bytes["fwd"], adj.last_recv_bytes["bwd"]]
    stats = torch.tensor([elapsed, t_fwd, t_bwd if bwd_ms else 0.0, t_fwd_local, recv[0], recv[1]],
                         device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        tot = torch.tensor([nnz_local, n_local, peak_setup or 0], device=dev, dtype=torch.float64)
        mx = tot.clone()
        dist.all_reduce(tot)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        nnz_total, n_total = int(tot[0].item()), int(tot[1].item())
        peak_setup = int(mx[2].item())
    else:
        nnz_total, n_total = nnz, n
    elapsed, t_fwd, t_bwd, t_fwd_local_max = [float(v) for v in stats.tolist()[:4]]
    recv_max = [int(v) for v in stats.tolist()[4:]]

    # ---------------------------------------------------------------- untimed side measurements
    def count_host_syncs(fn):
        """MEASURED: host synchronisations of one call of `fn`, by two detectors at once — torch's
        sync debug mode (warns on synchronising calls; a prototype that misses some) and hooks on
        the tensor methods that read device memory back (`.item()`, `.tolist()`, `.cpu()`,
        `torch.nonzero`); the larger count is reported.  The detectors are checked against a
        deliberate `.item()` first: a counter that cannot see that one reports "unmeasured"."""
        import warnings
        hooks = []

        def run(call):
            seen = []
            for name in ("item", "tolist", "cpu"):
                real = getattr(torch.Tensor, name)
                hooks.append((torch.Tensor, name, real))
                setattr(torch.Tensor, name, (lambda r, nm: lambda t, *a, **k: (
                    seen.append(nm) if t.is_cuda else None, r(t, *a, **k))[1])(real, name))
            real_nz = torch.nonzero
            hooks.append((torch, "nonzero", real_nz))
            torch.nonzero = lambda *a, **k: (seen.append("nonzero"), real_nz(*a, **k))[1]
            warned = 0
            try:
                torch.cuda.set_sync_debug_mode("warn")
                with warnings.catch_warnings(record=True) as caught:
                    warnings.simplefilter("always")
                    call()
                warned = sum("synchroniz" in str(w.message).lower() for w in caught)
            finally:
                torch.cuda.set_sync_debug_mode("default")
                while hooks:
                    obj, name, real = hooks.pop()
                    setattr(obj, name, real)
            return max(warned, len(seen))
        torch.cuda.synchronize()
        try:
            probe = torch.ones(1, device=dev)
            if run(lambda: probe.item()) < 1:
                return "unmeasured: the detectors did not see a deliberate .item()"
            n_sync = run(fn)
        except Exception as ex:                  # (the counter must never cost the bench line)
            return f"unmeasured: {ex!r}"
        torch.cuda.synchronize()
        return n_sync

    def dense_bwd_figures(l2_ms, l1_ms, n_steps):
        """The transpose product at FULL height (dense gradient): ms, GEdge/s and its roofline
        fraction (same algorithmic byte model as the forward product; Âᵀ has the same nnz)."""
        if not l2_ms:
            return {}
        t2 = float(np.mean(l2_ms))
        t1 = float(np.sum(l1_ms)) / max(1, n_steps) if l1_ms else 0.0
        rp_b = 4 if nnz_local < 2 ** 31 - 1 else 8
        alg_t = algorithmic_bytes(nnz_local, n_local, feat, esize, rp_b)
        return {"spmm_bwd_dense_ms_layer2_layer1": [round(t2, 4), round(t1, 4)],
                "spmm_bwd_dense_gedges": round(nnz_total / (t2 * 1e-3) / 1e9, 4),
                "spmm_bwd_dense_roofline_frac": round(alg_t / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}

    extras = {}
    syncs = None
    if not args.spmm_only:
        syncs = count_host_syncs(step)
    if not args.spmm_only and not args.no_extras and snapshot is not None:
        snapshot_keep, snapshot = snapshot, None            # free-running: no restore
        _, free = timed(max(args.steps, 8), step)
        extras["ms_per_step_free_running"] = {
            "first": round(free[0], 3), "last": round(free[-1], 3), "steps": len(free),
            "note": "same epochs WITHOUT the snapshot restore, continuing from the last timed "
                    "state: on random labels the hidden units of hub vertices die and the "
                    "layer-1 backward product shrinks — why the graded figure is taken at fixed "
                    "parameters"}
        snapshot = snapshot_keep
        if world == 1:
            try:
                rec2 = []
                restore_snapshot()
                epoch(dense_loss=True)                      # warm-up of the dense-gradient shape
                spmm_mod.set_timing_records(rec2)

                def dense_step():
                    restore_snapshot()
                    epoch(dense_loss=True)
                wall, _ = timed(3, dense_step)
                spmm_mod.set_timing_records(None)
                bd2 = [a.elapsed_time(b) for tag, a, b, _ in rec2 if tag in ("bwd", "bwd_l2")]
                bd1 = [a.elapsed_time(b) for tag, a, b, _ in rec2 if tag == "bwd_l1"]
                extras["ms_per_step_dense_loss"] = round(wall / 3 * 1e3, 3)
                extras.update(dense_bwd_figures(bd2, bd1, 3))
                extras["host_syncs_per_step_dense_loss"] = count_host_syncs(dense_step)
                extras["dense_loss_note"] = ("NLL over ALL rows instead of the idx_train share (the fork's "
                                             "live loss reduces over every node, pygcn/train.py:151-155): "
                                             "every gradient row is non-zero, no product can skip operand "
                                             "rows; `python bench.py --dense-loss` times this epoch as the "
                                             "main figure")
            except Exception as ex:                         # e.g. out of memory on a small device
                spmm_mod.set_timing_records(None)
                extras["ms_per_step_dense_loss"] = None
                extras["dense_loss_note"] = f"failed: {ex!r}"
            try:           # opt-in: z = A x X (two constants) computed once, not every epoch
                from pygcn_amd import fused as fused_mod
                fused_mod.set_input_product_cache(True)
                restore_snapshot()
                epoch()

                def cached_step():
                    restore_snapshot()
                    epoch()
                wall, _ = timed(5, cached_step)
                extras["ms_per_step_cached_input_product"] = round(wall / 5 * 1e3, 3)
                extras["cached_input_product_note"] = (
                    "NOT the graded figure: with pygcn_amd.fused.set_input_product_cache(True) the layer-1 "
                    "product z = A x X — a product of two constants of the run, since layer 1 is evaluated as "
                    "(A x X) x W1 — is computed once and reused, so an epoch holds ONE forward sparse product; "
                    "`ms_per_step` keeps both inside every timed epoch as SURVEY 8(d) defines it")
            except Exception as ex:
                extras["ms_per_step_cached_input_product"] = None
                extras["cached_input_product_note"] = f"failed: {ex!r}"
            finally:
                fused_mod.set_input_product_cache(False)
                if hasattr(adj, "_input_product"):
                    del adj._input_product
            if dt == "f32":
                # the same graded epoch (and the dense-gradient epoch) under the OTHER GEMM schemes:
                # "h2" (two scaled fp16 parts: 22 bits, half the matrix work) and hipBLASLt's fp32
                for other, key in (("h2" if args.gemm_scheme != "h2" else "bf16x3", None), ("exact", "hipblaslt")):
                    key = key or other
                    try:
                        spmm_mod.set_gemm_scheme(other)
                        restore_snapshot()
                        epoch()

                        def other_step():
                            restore_snapshot()
                            epoch()
                        wall, _ = timed(3, other_step)
                        extras[f"ms_per_step_{key}_gemm"] = round(wall / 3 * 1e3, 3)
                        if other != "exact":
                            restore_snapshot()
                            epoch(dense_loss=True)

                            def other_dense():
                                restore_snapshot()
                                epoch(dense_loss=True)
                            wall, _ = timed(3, other_dense)
                            extras[f"ms_per_step_dense_loss_{key}_gemm"] = round(wall / 3 * 1e3, 3)
                    except Exception as ex:
                        extras[f"ms_per_step_{key}_gemm"] = None
                        extras[f"{key}_gemm_note"] = f"failed: {ex!r}"
                    finally:
                        spmm_mod.set_gemm_scheme(args.gemm_scheme)
            try:
                restore_snapshot()
                epoch(reference_call=True)

                def ref_step():
                    restore_snapshot()
                    epoch(reference_call=True)
                wall, _ = timed(3, ref_step)
                extras["ms_per_step_reference_call"] = round(wall / 3 * 1e3, 3)
                extras["host_syncs_per_step_reference_call"] = count_host_syncs(ref_step)
                extras["reference_call_note"] = ("upstream's lines unchanged: output = model(features, adj); "
                                                 "F.nll_loss(output[idx_train], labels[idx_train]) — the "
                                                 "selection output[idx_train] hands the layers' backward "
                                                 "passes a row-sparse gradient (pygcn_amd/rowgrad.py), so "
                                                 "they run on the same transpose block and compact rows "
                                                 "as the rows= path, one autograd node per layer")
            except Exception as ex:
                extras["ms_per_step_reference_call"] = None
                extras["reference_call_note"] = f"failed: {ex!r}"

    # the forward product on the CACHE-HOSTILE graph (uniform degree 10 + I: no hubs, every gather an
    # HBM miss), next to the R-MAT figure: is the schedule good, or is R-MAT kind?  
    if world == 1 and args.graph == "rmat" and not args.no_extras and not args.spmm_only and dt == "f32":
        try:
            del model, opt, fwd_model
            snapshot = None
            torch.cuda.empty_cache()
            rp_u, c_u, v_u = uniform_graph(n, max(1, e // n), seed=46, device=dev)
            g_u = CSRGraph(rp_u, c_u, v_u, (n, n), **kw)
            g_u.plan()
            nnz_u = int(c_u.numel())
            with torch.no_grad():
                for _ in range(2):
                    spmm_mod.spmm_csr(g_u, x)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
                for i in range(5):
                    ev[i].record()
                    spmm_mod.spmm_csr(g_u, x)
                ev[5].record()
                torch.cuda.synchronize()
            t_u = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(5)]))
            alg_u = algorithmic_bytes(nnz_u, n, feat, esize, 4 if nnz_u < 2 ** 31 - 1 else 8)
            tr_u, tr_note = measured_traffic(args.config + "_uniform", dt)
            extras["roofline_uniform"] = {
                "graph": f"uniform: {n} vertices x {max(1, e // n)} random neighbours (seed 46, duplicates "
                         f"removed) + I, row-normalized -> nnz {nnz_u}; max degree "
                         f"{int((rp_u[1:] - rp_u[:-1]).max())}",
                "spmm_fwd_ms": round(t_u, 4), "value_gedges": round(nnz_u / (t_u * 1e-3) / 1e9, 4),
                "bound": "hbm", "achieved": round(alg_u / (t_u * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(alg_u / (t_u * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "algorithmic_bytes_per_launch": alg_u, "traffic": tr_u, "traffic_source": tr_note,
                "note": "same kernel, same byte model, 5 launches after 2 warm-ups (HIP events); the guide's "
                        "measured ceiling for random whole-row gathers from HBM is 5.5-5.8 TB/s "
                        "(MI355X_MICROARCH.md) = 0.69-0.72 of the 8 TB/s spec"}
            del g_u, rp_u, c_u, v_u
        except Exception as ex:
            extras["roofline_uniform"] = {"error": repr(ex)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gedges = nnz_total / (t_fwd * 1e-3) / 1e9
        rp_bytes = 4 if nnz_local < 2 ** 31 - 1 else 8
        # roofline of the dominant kernel (forward SpMM launch) on THIS rank's shard
        alg = algorithmic_bytes(nnz_local, n_local, feat, esize, rp_bytes)
        kernel_ms = t_fwd_local
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_note = measured_traffic(args.config + ("_uniform" if args.graph == "uniform" else ""),
                                                 dt) if world == 1 else (None, "single-GPU figure only")
        line = {
            "metric": "SpMM GEdge/s + fwd+bwd ms/epoch, 10M-node synthetic CSR, feat_dim=256",
            "value": round(gedges, 4), "unit": "GEdge/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong",
            **({"rehearsal": "one GPU shared by all ranks, gloo staged through the host: "
                             "code-path check only, not a measurement"} if args.rehearsal else {}),
            "vs_baseline": None, "dtype": dt, "data": "synthetic",
            "config": {"workload": f"{args.config}: " + ("R-MAT(0.57,0.19,0.19,0.05)" if args.graph == "rmat" else
                                                         "UNIFORM random neighbours (cache-hostile, --graph uniform)")
                                   + f" {n_total} nodes / "
                                   f"{e} sampled edges -> nnz {nnz_total} (dedupe + I, "
                                   f"row-normalized), feat_dim {feat}, 2-layer GCN "
                                   f"{feat}->{feat}->{feat}, fwd+bwd+Adam, dropout {args.dropout}, "
                                   + ("NLL over ALL vertices (--dense-loss)" if args.dense_loss else
                                      "NLL on the first 140/2708 of the vertices (upstream idx_train share)"),
                       "nodes": n_total, "nnz": nnz_total, "feat_dim": feat,
                       "parallelism": (f"row-block x{world} (the fixed {args.config} graph cut into "
                                       f"nnz-balanced row blocks, built shard-locally), "
                                       f"{selfcheck.get('exchange_chosen', args.exchange)} forward exchange "
                                       "(backward: static row-sparse gradient halo)"
                                       + (", pipelined by source block (own rows | halo rows)"
                                          if adj.overlap else "")
                                       + f", rank0 receives {adj.exchange_rows()[0]} of "
                                       f"{adj.exchange_rows()[1]} remote rows per dense exchange"
                                       + ("; the halo rows of the constant feature matrix are "
                                          "exchanged once before the timed region and held (like "
                                          "the adjacency block), so layer 1 is the local product "
                                          "A_r x [X_r ; X_halo] followed by one GEMM over the "
                                          "rank's own rows, without an exchange: 2 exchanges per epoch "
                                          "(layer 2 forward dense, layer 2 backward non-zero rows only), not 4"
                                          ))
                       if world > 1 else "single GPU",
                       "mode": "spmm-only" if args.spmm_only else
                               ("train-epoch (upstream's unchanged lines)" if args.reference_call else
                                "train-epoch (loss over all vertices)" if args.dense_loss else "train-epoch")},
            "stationary": ("parameters + Adam state restored from the post-warm-up snapshot at the "
                           "start of every timed epoch (inside the timed region): every timed "
                           "epoch is epoch warmup+1") if snapshot is not None else None,
            "ms_per_step_min_max": [round(min(per_step), 3), round(max(per_step), 3)],
            "ms_per_step_median_hip_events": round(float(np.median(per_step)), 3),
            "tolerance_note": "parity contract 1e-5 relative (normwise) is per step (one forward/backward), "
                              "against the reference's fp32 CPU arithmetic; gradients that are fp32 reductions "
                              "over the graph's vertices are gated at 1e-5 against a float64 evaluation of the "
                              "same step and at 1e-5 + the reference's own measured rounding error against the "
                              "fp32 reference (tests/conftest.py assert_parity); two fp32 routes against each "
                              "other 2e-5; a 200-epoch Adam trajectory 1e-3 (chained fp32 steps)",
            "gemm_scheme": ({"bf16x3": "bf16x3: the dense 256x256 products of the epoch (not the SpMM, which is "
                                       "plain fp32 FMA — `value` and `roofline` do not depend on this) run on the "
                                       "repo's own fp32-EQUIVALENT MFMA kernels: both operands in three bf16 parts "
                                       "(24-bit significand), six MFMAs per product, fp32 accumulation "
                                       "(pygcn_amd/csrc/gcn_gemm.hip, gcn_gemm_xw256_f32_b3 / gcn_gemm_atg256_f32_b3); "
                                       "`ms_per_step_h2_gemm` = the same epoch on the 22-bit two-part fp16 scheme, "
                                       "`ms_per_step_hipblaslt_gemm` = on hipBLASLt's fp32",
                             "h2": "h2: power-of-two-scaled two-part fp16 MFMA emulation of fp32 (22-bit "
                                   "significand; opt-in, NOT fp32-equivalent); `ms_per_step_bf16x3_gemm` = the "
                                   "same epoch on the fp32-equivalent three-part scheme",
                             "exact": "exact: every dense product on hipBLASLt's fp32 path (torch.mm)"}
                            [args.gemm_scheme]) if dt == "f32" else "bf16 MFMA, fp32 accumulate",
            "host_syncs_per_step": syncs,
            "host_syncs_note": "MEASURED on one extra epoch of the timed kind after the timed region "
                               "(torch.cuda.set_sync_debug_mode warnings and hooks on .item() / .tolist() / "
                               ".cpu() / nonzero, validated against a deliberate .item() first); the "
                               "dense-loss and upstream-lines epochs carry their own counts",
            "spmm_fwd_ms": round(t_fwd, 4), "spmm_bwd_ms": round(t_bwd, 4),
            "spmm_bwd_gedges": round(nnz_total / (t_bwd * 1e-3) / 1e9, 4) if bwd_ms else None,
            "spmm_bwd_ms_layer2_layer1": [round(bwd_l2, 4), round(bwd_l1, 4)],
            "spmm_bwd_note": ("layer 2: the transpose product on the block of A^T with columns idx_train "
                              "(5 % of the rows: the only non-zero rows of the operand) and rows = the "
                              "vertices that have a neighbour in idx_train (16 %: the only rows of the "
                              "result that can be non-zero), cut once per (graph, idx_train), compact "
                              "operand in, compact result out; layer 1 (input needs no gradient) is evaluated as (A x X) x W1, "
                              "so grad_W1 = (A x X)^T x grad_pre reuses the A x X of the SAME step's "
                              "forward pass and needs no sparse product in backward (0 ms); `value` "
                              "and `roofline` are the unrestricted forward products; the "
                              "dense-gradient epoch is reported beside it") if world == 1 else
                             "layer 2: a STATIC halo of the labelled vertices' gradient rows (who sends what is fixed once per (graph, idx_train): one grouped P2P round of known sizes, no count exchange) + the product on the rank's [R2_r, R] block of A^T; layer 1: none (grad_W1 = (A_r x X)^T x grad_pre reuses the forward pass's A_r x X) (pygcn_amd/sharded_fused.py)",
            "spmm_launches_timed": len(fwd_ms) + len(bwd_ms),
            "spmm_local_fwd_ms_rank0": round(kernel_ms, 4),
            "graph_gen_s": round(t_gen, 2),
            **(dense_bwd_figures(l2 or plain, l1, args.steps) if args.dense_loss and world == 1 else {}),
            "loss_check": {"value": loss_check,
                           "what": "eval-mode (no dropout) mean NLL on idx_train at the seeded initial "
                                   "parameters, before the warm-up: the same graph, features, labels "
                                   "and parameters at every --gpus N, so this number must agree across "
                                   "world sizes to fp32 summation order"},
            **extras,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_note,
                         "achieved_is": "ALGORITHMIC bytes (gather model, no cache credit) / HIP-"
                                        "event duration; part of the gather is served by the "
                                        "256 MiB Infinity Cache, so it is not literal HBM traffic",
                         "kernel": ("spmm_wide_kernel<float,4>" if dt == "f32" and feat > 128 else
                                    "spmm_narrow_kernel") + " (forward gcn_spmm_csr launch)",
                         "algorithmic_bytes_per_launch": alg},
        }
        if world > 1:
            line["selfcheck"] = selfcheck
            for k in ("sharded_grad_check", "exchange_ab_ms", "exchange_chosen"):     # (top level too)
                if k in selfcheck:
                    line[k] = selfcheck[k]["max_err"] if k == "sharded_grad_check" else selfcheck[k]
            line["rccl_ranks"] = selfcheck.get("ranks")
            if sharded_rows_note is not None:
                line["sharded_path_note"] = sharded_rows_note
            line["spmm_plus_exchange_gedges"] = round(gedges, 4)
            line["spmm_only_gedges"] = round(nnz_total / (t_fwd_local_max * 1e-3) / 1e9, 4)
            line["spmm_only_note"] = ("all ranks' stored entries / the slowest rank's mean LOCAL "
                                      "product time per forward product (both launches of the "
                                      "pipelined form summed), exchange excluded")
            line["exchange_bytes_received_per_rank_max"] = {"fwd_dense": recv_max[0],
                                                            "bwd_sparse": recv_max[1]}
            line["setup_peak_bytes_per_rank_max"] = peak_setup
            line["setup_s_rank0"] = round(t_gen, 2)
            line["setup_stats_rank0"] = adj.setup_stats
            line["hidden_exchange"] = ("compressed: bitmask + non-zero values, W2 applied on arrival "
                                       "(--compress-hidden)") if adj.compress_hidden else \
                "dense rows of h1 x W2 (default; --compress-hidden is the A/B switch)"
            line["exchange_exposed_wait_ms_rank0"] = round(float(np.mean(wait_ms)), 4) if wait_ms else None
            line["exchange_exposed_wait_note"] = ("time the compute stream waits for the halo rows "
                                                  "AFTER the own-rows product has run under the "
                                                  "transfers (pipelined dense exchange, rank 0)")
        if world == 1 and not args.no_cpu_baseline:
            try:
                # the whole graph where it runs in seconds (C3-sized and below), a
                # 1/8 row block at C4, 1/64 at C5 (bounded CPU work; the sample is named in the line)
                frac = 1 if nnz <= 20_000_000 else (8 if nnz <= 200_000_000 else 64)
                line["cpu_baseline"] = cpu_baseline(rowptr, col, val, n, feat,
                                                    budget_rows=max(1, n // frac))
            except Exception as ex:
                line["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    gc = selfcheck.get("sharded_grad_check") if world > 1 else None
    if gc is not None and gc.get("ok") is False:
        # the sharded backward pass does not reproduce the single-GPU gradients: the line above is
        # not a measurement of the reference's training step
        sys.stderr.write(f"bench.py: sharded_grad_check FAILED: {gc}\n")
        raise SystemExit(4)


if __name__ == "__main__":
    main()
