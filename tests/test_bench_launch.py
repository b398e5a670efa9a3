"""`python bench.py --gpus N` with NO launcher (the shape of the driver's single-GPU command with
another N) must start its own ranks, print ONE JSON line from rank 0 and pass a rank's failure on
as a non-zero exit status — never hang, never exec-replace a process that has touched the GPU."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    return r, time.time() - t0


def test_launcher_passes_a_failing_rank_on_cpu():
    """No GPU here: every rank stops at `bench.py needs MI355X GPUs` — the launcher must come back
    with a non-zero status (and must not have tried the GPU itself)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check")
    r, dt = _run(["--gpus", "2", "--rehearsal", "--config", "tiny", "--steps", "1", "--warmup", "0"], 300)
    assert r.returncode != 0
    assert "needs MI355X" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert dt < 240


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    r, _ = _run(["--gpus", "2", "--rehearsal", "--config", "c3", "--steps", "2", "--warmup", "1",
                 "--no-extras"], 900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    for key in ("spmm_only_gedges", "spmm_plus_exchange_gedges", "exchange_bytes_received_per_rank_max",
                "roofline", "loss_check"):
        assert key in line, key
    assert line["exchange_bytes_received_per_rank_max"]["fwd_dense"] > 0
    # round 4: the N > 1 run validates itself before it times anything (pygcn_amd/selfcheck.py) —
    # the one-GPU rehearsal carries every field of the real run
    sc = line["selfcheck"]
    assert line["rccl_ranks"] == 2 and len(sc["devices"]) == 2 and sc["backend"] == "gloo"
    assert sc["overlap_selftest"]["agrees"] is True and sc["overlap_selftest"]["max_err"] <= 1e-5
    assert sc["sharded_grad_check"]["ok"] is True and line["sharded_grad_check"] <= 5e-5
    assert set(line["exchange_ab_ms"]) == {"halo", "allgather", "rccl-allgather", "compress-hidden"}
    assert line["exchange_chosen"] in line["exchange_ab_ms"]
    assert all(v <= 1e-5 for v in sc["exchange_ab_max_err_vs_halo"].values())
    assert sc["link_rate"]["gb_per_s"] > 0
    # the same problem at every world size: the single-GPU run must print the same eval loss
    r1, _ = _run(["--gpus", "1", "--config", "c3", "--steps", "1", "--warmup", "0", "--no-extras",
                  "--no-cpu-baseline"], 900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    a, b = line["loss_check"]["value"], one["loss_check"]["value"]
    assert abs(a - b) <= 2e-5 * abs(b), (a, b)
    assert line["config"]["nnz"] == one["config"]["nnz"]


@pytest.mark.gpu
def test_bench_launcher_reports_a_dead_rank():
    r, dt = _run(["--gpus", "2", "--rehearsal", "--config", "tiny", "--steps", "1", "--warmup", "0",
                  "--fail-rank", "1"], 600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert dt < 500


@pytest.mark.gpu
def test_bench_dense_loss_mode_and_measured_fields():
    """`bench.py --dense-loss` (the loss over all vertices as the timed epoch) and the fields the
    line must carry: measured host synchronisations, the full-height transpose product as a
    GEdge/s + roofline pair, the GEMM scheme, the cross-world-size loss check."""
    r, _ = _run(["--config", "tiny", "--dense-loss", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                 "--no-extras"], 600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["mode"] == "train-epoch (loss over all vertices)"
    assert line["host_syncs_per_step"] == 0
    assert line["spmm_bwd_dense_gedges"] > 0 and line["spmm_bwd_dense_roofline_frac"] > 0
    assert line["gemm_scheme"].startswith("bf16x3") and line["loss_check"]["value"] > 0
    assert line["roofline"]["bound"] == "hbm" and line["unit"] == "GEdge/s"
    r2, _ = _run(["--config", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], 600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    full = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    for key in ("ms_per_step_dense_loss", "host_syncs_per_step_dense_loss", "ms_per_step_reference_call",
                "host_syncs_per_step_reference_call", "ms_per_step_h2_gemm", "ms_per_step_dense_loss_h2_gemm",
                "ms_per_step_hipblaslt_gemm", "spmm_bwd_dense_gedges", "roofline_uniform"):
        assert key in full, key
    assert full["host_syncs_per_step"] == 0 and full["host_syncs_per_step_dense_loss"] == 0
    assert full["host_syncs_per_step_reference_call"] == 0
    assert full["roofline_uniform"]["frac"] > 0 and full["roofline_uniform"]["value_gedges"] > 0


@pytest.mark.gpu
def test_bench_explicit_exchange_and_scheme_flags():
    """`--exchange rccl-allgather` (the north star's literal collective for the activations), `--gemm-scheme
    h2` and `--graph uniform` are accepted and named in the line."""
    r, _ = _run(["--gpus", "2", "--rehearsal", "--config", "tiny", "--steps", "1", "--warmup", "1",
                 "--no-extras", "--exchange", "rccl-allgather", "--gemm-scheme", "h2"], 900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["exchange_chosen"] == "rccl-allgather" and "exchange_ab_ms" not in line
    assert line["gemm_scheme"].startswith("h2") and line["selfcheck"]["sharded_grad_check"]["ok"] is True
    assert "rccl-allgather forward exchange" in line["config"]["parallelism"]
    r, _ = _run(["--config", "tiny", "--graph", "uniform", "--steps", "1", "--warmup", "1", "--no-extras",
                 "--no-cpu-baseline"], 600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "UNIFORM" in line["config"]["workload"] and line["value"] > 0
