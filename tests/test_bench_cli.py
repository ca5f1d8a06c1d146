"""bench.py's launch contract on a machine without GPUs: a bare `python bench.py --gpus 2` must spawn its own two ranks
(one process per GPU through torch.distributed.run on 127.0.0.1) instead of refusing, and each rank must fail loudly --
"no GPU visible", exit code 3 -- because the engine has no CPU fallback."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check of the spawn path")
def test_bare_multi_gpu_launch_spawns_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and p.returncode != 2
    assert p.stderr.count("no GPU visible; the engine has no CPU fallback") == 2, p.stderr[-2000:]
    assert p.stdout.strip() == ""


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_single_gpu_without_device_fails_loudly():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 3 and "no CPU fallback" in p.stderr and p.stdout.strip() == ""


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check of the spawn path")
def test_config_arguments_reach_every_spawned_rank():
    """`python bench.py --config 3 --gpus 2 ...`: the self-spawn forwards the whole command line, so both ranks parse
    `--config 3` (a rank that lost it would still die on "no GPU", so the argument echo is checked through an unknown flag:
    argparse's error names it on every rank)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and p.stderr.count("no GPU visible; the engine has no CPU fallback") == 2, p.stderr[-2000:]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--substeps", "7", "--bogus-flag"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--bogus-flag" in p.stderr  # rejected before any spawn: one parser, one command line
    assert p.stdout.strip() == ""


@pytest.mark.gpu
def test_forced_single_rank_exchange_line_fields():
    """On the GPU box: the RCCL path of bench.py with ONE rank (MJB_BENCH_FORCE_GATHER=1) -- init_process_group("nccl"),
    ExternalStream interop, the side-stream all-gather / all-reduces and the final barrier all run; the line must say so."""
    import json
    env = dict(os.environ, MJB_BENCH_FORCE_GATHER="1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--substeps", "100",
                        "--no-cpu-baseline", "--no-other-configs"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["metrics"]["env_steps"] == 4096 * 100 * 4 and line["metrics"]["nenv"] == 4096  # reduced over the (one) rank
    assert "forced single-rank gather: sensordata round trip ok" in p.stderr


# ---- the N-rank control flow without GPUs: `--dry-ranks N` = rendezvous with a timeout, shard ranges, the real
# OverlappedExchange on thread streams, barrier fences, per-rank times gathered + MAX over ranks, the rank-0 line, exit codes ----
def _run_bench(*argv, timeout=300):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


@pytest.mark.parametrize("ranks", [1, 2, 4, 8])
def test_dry_ranks_print_one_line_from_rank_zero(ranks):
    p, lines = _run_bench("--dry-ranks", str(ranks), "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1 and p.stdout.count("\n") == 1      # ONE line on stdout (gloo's banners went to stderr)
    line = lines[0]
    assert line["n_gpus"] == ranks and line["dry_ranks"] == ranks and line["steps"] == 3 and line["warmup"] == 1
    assert line["exchange_ok"] is True                        # every rank's shard of the LAST launch arrived, metrics SUM / MAX right
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and "DRY RUN" in line["data"]
    assert 0 < line["rank_ms_per_step"]["min"] <= line["rank_ms_per_step"]["max"] <= line["ms_per_step"] * 1.0000001
    assert line["exchange_ms"] is not None and line["exchange_ms"] >= 0
    assert abs(line["value"] - ranks * 64 * 10 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]   # whole-job aggregate
    assert line["config"]["gloo_ranks"] == ranks              # what the process group says, not the environment variable
    assert line["cpu_baseline_leg"] is (ranks == 1)           # the CPU leg runs on rank 0 at N = 1 only


def test_world_size_mismatch_fails_the_line():
    """A launcher that started fewer ranks than --gpus / --dry-ranks asks for must not produce a line that looks like an N-rank one."""
    import json
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-ranks", "2", "--steps", "2", "--warmup", "1", "--dist-timeout", "20"],
                       env=env, capture_output=True, text=True, timeout=120)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode != 0 and len(lines) == 1 and "error" in lines[0] and "process group has 1 rank" in lines[0]["error"]


@pytest.mark.parametrize("bad", [0, 1])
def test_a_failing_rank_ends_in_one_error_line_not_a_hang(bad):
    p, lines = _run_bench("--dry-ranks", "2", "--steps", "3", "--warmup", "1", "--dry-fail-rank", str(bad), "--dist-timeout", "20", timeout=120)
    assert p.returncode != 0
    assert len(lines) == 1 and "error" in lines[0] and lines[0]["rank"] == 0
    assert f"injected failure on rank {bad}" in lines[0]["error"]
    assert "value" not in lines[0]
