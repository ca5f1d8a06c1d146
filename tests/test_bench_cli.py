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
