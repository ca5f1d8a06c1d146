"""N > 1 path on CPU: world_size-2 gloo processes each step their env shard (CPU oracle as the stepper -- test
harness only) with the global Philox env offset, all-gather sensordata, and must reproduce the single-process
rollout of all envs bit for bit.  Exercises mujoco_ros_pkgs_amd.sharding exactly as bench.py uses it."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, E, K, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import random_franka_state
    from mujoco_ros_pkgs_amd import mjcf, sharding
    from oracle import pyoracle
    model = mjcf.load_asset("franka_like")
    qpos, qvel = random_franka_state(model, world * E, seed=77)
    lo, hi = sharding.shard_range(rank, world, E)
    _, _, sens = pyoracle.rollout(model, qpos[lo:hi], qvel[lo:hi], K, noise_std=20.0, noise_rate=0.1, seed=12345,
                                  env_offset=lo)
    gathered = sharding.gather_sensordata(torch.from_numpy(np.ascontiguousarray(sens)))
    # the 16-double metrics vector: [0:8] summed, [8:16] maximised over the ranks (what mjb_metrics hands to the all-reduce)
    met = torch.zeros(16, dtype=torch.float64)
    met[0], met[1], met[6] = E * K, rank, E
    met[8], met[9], met[10] = float(np.abs(sens).max()), 10.0 + rank, K * 0.002
    sharding.reduce_metrics(met)
    if rank == 0:
        q.put((gathered.numpy().copy(), met.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process(oracle_built, franka):
    from conftest import random_franka_state
    from mujoco_ros_pkgs_amd import sharding
    world, E, K = 2, 6, 15
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, met = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    qpos, qvel = random_franka_state(franka, world * E, seed=77)
    _, _, ref = oracle_built.rollout(franka, qpos, qvel, K, noise_std=20.0, noise_rate=0.1, seed=12345, env_offset=0)
    assert got.shape == (world * E, franka["nsensordata"])
    assert np.array_equal(got, ref)
    assert met[0] == world * E * K and met[1] == 0 + 1 and met[6] == world * E
    assert met[8] == np.abs(ref).max() and met[9] == 11.0 and abs(met[10] - K * 0.002) < 1e-15
    one = torch.arange(16, dtype=torch.float64)
    assert sharding.reduce_metrics(one) is one and one[5] == 5   # world size 1: no collective
    with pytest.raises(ValueError):
        sharding.reduce_metrics(torch.zeros(8, dtype=torch.float64))
    assert sharding.shard_range(1, 2, 4096) == (4096, 8192)
    with pytest.raises(ValueError):
        sharding.shard_range(2, 2, 4096)
    t = torch.zeros(3, 2, dtype=torch.float64)
    assert sharding.gather_sensordata(t) is t  # world size 1: no collective
