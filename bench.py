#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched step engine on BASELINE.json's configs[1]
("Franka Panda 9-DoF no-contact, 4096 envs, 1xMI355X fp64"), weak-scaled to N GPUs of one node.

One bench "step" = ONE fused kernel launch that advances every env of the rank's batch by
``--substeps`` physics steps (rollouts of K = 1000 steps), driven by the reference's
Ornstein-Uhlenbeck ctrl-noise injector generated on device (counter-based Philox, seed 12345), followed
-- when N > 1 -- by the RCCL all-gather of ``sensordata`` over the node.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--substeps S] [--envs E] [--lanes G]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = {"franka_like": 712, "franka_table": 1072, "shadow_hand_like": 2136}  # 8*(2nq+2nv+2nv+nu+S)+8
# per-model workload: (BASELINE config label, OU ctrl-noise std [= 0.5 * ctrlrange of the big actuators], default envs per GPU)
WORKLOADS = {
    "franka_like": ("BASELINE configs[1]: Franka-Panda-like 9-DoF arm, no contacts", 0.5 * 87.0, 4096),
    "franka_table": ("BASELINE configs[2]: Franka-like arm + table + cube contacts", 0.5 * 87.0, 4096),
    "shadow_hand_like": ("BASELINE configs[4]: Shadow-Hand-like 24-DoF hand + in-hand cube (Newton, elliptic cones)", 0.1, 1024),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)
FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet fp64 vector peak; replaced by the measured fma rate when the profile holds one


def synthetic_state(model, nenv, seed):
    """SURVEY.md §8d inputs: mid-range pose + U(-0.1,0.1) rad, fingers U(0,0.04) m, qvel U(-0.1,0.1)."""
    rng = np.random.default_rng(seed)
    rngj = np.asarray(model["jnt_range"])
    mid = 0.5 * (rngj[:, 0] + rngj[:, 1])
    qpos = np.tile(np.asarray(model["qpos0"]), (nenv, 1))
    qvel = np.zeros((nenv, model["nv"]))
    for j in range(model["njnt"]):
        t, qa, da = model["jnt_type"][j], model["jnt_qposadr"][j], model["jnt_dofadr"][j]
        if t == 3:
            qpos[:, qa] = mid[j] + rng.uniform(-0.1, 0.1, nenv)
            qvel[:, da] = rng.uniform(-0.1, 0.1, nenv)
        elif t == 2:
            qpos[:, qa] = rng.uniform(rngj[j, 0], rngj[j, 1], nenv)
            qvel[:, da] = rng.uniform(-0.1, 0.1, nenv)
        elif t == 0:  # free object: at rest 1 cm above contact, identity quat + yaw U(-pi, pi)
            yaw = rng.uniform(-np.pi, np.pi, nenv)
            qpos[:, qa + 3] = np.cos(yaw / 2)
            qpos[:, qa + 4:qa + 6] = 0
            qpos[:, qa + 6] = np.sin(yaw / 2)
    return qpos, qvel


class DevArray:
    """Zero-copy view of an engine HBM array for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def initial_state(name, model, nenv, seed):
    if name == "shadow_hand_like":
        from mujoco_ros_pkgs_amd import workloads
        return workloads.hand_grasp_states(model, nenv, seed)
    return synthetic_state(model, nenv, seed)


def cpu_baseline(name, model, noise_std, nsteps_total_target_s=12.0):
    """Time the CPU oracle ("port": from-scratch restatement, -O3 -march=native) on a bounded sample of the
    same workload, all host cores, one env per thread at a time."""
    from oracle import pyoracle
    pyoracle.build()
    cores = os.cpu_count() or 1
    nenv, nsteps = 4 * cores, 200
    qpos, qvel = initial_state(name, model, nenv, seed=999)
    kw = dict(noise_std=noise_std, noise_rate=0.1, seed=12345, nthreads=cores, fast=True)
    t0 = time.perf_counter()
    pyoracle.rollout(model, qpos, qvel, nsteps, **kw)
    dt = time.perf_counter() - t0
    rate = nenv * nsteps / dt
    # scale the sample to ~target seconds
    nsteps2 = int(min(max(nsteps, rate * nsteps_total_target_s / nenv), 200000))
    t0 = time.perf_counter()
    pyoracle.rollout(model, qpos, qvel, nsteps2, **kw)
    dt = time.perf_counter() - t0
    return {"value": nenv * nsteps2 / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{nenv} envs x {nsteps2} steps of the same {name} workload (OU ctrl noise), oracle/libmjo_fast.so "
                      f"(gcc -O3 -march=native), {cores} threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--substeps", type=int, default=1000, help="physics steps fused into one launch")
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (weak scaling); 0 = the config's own (4096; 1024 for the hand)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per env (0 = engine default)")
    ap.add_argument("--epb", type=int, default=0, help="envs per workgroup (0 = engine default)")
    ap.add_argument("--model", default="franka_like")
    ap.add_argument("--solver", default="", choices=["", "PGS", "Newton"], help="override the model's constraint solver")
    ap.add_argument("--nefcmax", type=int, default=0, help="override the model's constraint-row capacity")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the engine has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    force_gather = os.environ.get("MJB_BENCH_FORCE_GATHER", "0") == "1"  # exercise the RCCL path on one GPU
    # The contract is ONE JSON line on stdout: RCCL prints a version banner through C stdio (flushed at exit, i.e. after
    # anything Python prints), so everything written to fd 1 from here on goes to stderr and the line is written to the
    # saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mujoco_ros_pkgs_amd import engine, mjcf, sharding

    model = mjcf.load_asset(args.model, **({"nefcmax": args.nefcmax} if args.nefcmax else {}))
    if args.solver:
        model = mjcf.Model(dict(model))
        model["solver"] = {"PGS": 0, "Newton": 2}[args.solver]
    cm = engine.CompiledModel(model)
    label, noise_std, default_envs = WORKLOADS.get(args.model, (args.model, 1.0, 4096))
    E, S = (args.envs or default_envs), args.substeps
    batch = engine.Batch(cm, E, local_rank)
    batch.set_launch(args.lanes, args.epb)
    qpos, qvel = initial_state(args.model, model, E, seed=1000 + rank)
    batch.set("qpos", qpos)
    batch.set("qvel", qvel)
    # reference injector: tau = 0.1 s, std = 0.5 * ctrlrange (87 N m on the big joints), seed 12345
    env_lo, _ = sharding.shard_range(rank, world, E)
    batch.set_ctrl_noise(noise_std, 0.1, 12345, env_lo)
    batch.synchronize()

    nsd = model["nsensordata"]
    sens_local = torch.as_tensor(DevArray(batch.device_ptr("sensordata"), (E, nsd)), device=f"cuda:{local_rank}")
    gather = world > 1 or force_gather
    sens_all = torch.empty((world * E, nsd), dtype=torch.float64, device=f"cuda:{local_rank}") if gather else None

    def one_step():
        batch.step(S)
        if gather:
            batch.synchronize()  # kernel ran on the engine's stream; the gather runs on torch's
            if world > 1:
                sharding.gather_sensordata(sens_local, sens_all)
            else:
                dist.all_gather_into_tensor(sens_all, sens_local)

    def fence():
        if gather:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity: state still finite (non-finite envs are auto-reset and counted by the engine)
    finite = bool(np.all(np.isfinite(batch.get("qpos"))))
    resets = batch.warning_count()  # mj_check* auto-resets since the batch was made (must be 0 on config 2)

    # dominant-kernel duration measured with HIP events on the engine's own stream
    kern_ms = batch.time_steps(S, max(1, min(args.steps, 5)))

    if rank == 0:
        traffic, fp64 = None, None
        try:  # HBM bytes and executed fp64 flops per launch from the committed rocprofv3 PMC passes (same kernel, same config)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            if (E, S, args.model) == (4096, 1000, "franka_like"):
                traffic = (pmc["FETCH_SIZE"]["mean_per_dispatch"] + pmc["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
                if "fp64_executed_flops_per_dispatch" in pmc:
                    flops = pmc["fp64_executed_flops_per_dispatch"]
                    peak = pmc.get("fp64_peak_measured", {}).get("fp64_fma_tflops", FP64_PEAK_TFLOPS)
                    ach = flops / (kern_ms * 1e-3) / 1e12
                    fp64 = {"bound": "fp64-valu", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                            "executed_flops_per_launch": flops,
                            "note": "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes (profiles/r01_pmc_summary.json); "
                                    "peak = tools/ubench/fp64_peak on the same box"}
        except Exception:
            traffic, fp64 = None, None
        value = world * E * S * args.steps / elapsed
        bytes_per_launch = ALGO_BYTES_PER_ENV_STEP.get(args.model, 712) * E * S
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{label}, {E} envs per GPU, fp64, Euler dt={model['timestep'][0]}",
                       "envs_per_gpu": E, "physics_steps_per_launch": S, "model": args.model,
                       "solver": {0: "PGS", 1: "CG", 2: "Newton"}[int(model["solver"])] if model["nefcmax"] else "none",
                       "ctrl": f"on-device OU noise (Philox seed 12345, tau 0.1 s, std {noise_std:g})",
                       "parallelism": f"env-sharded x{world}, RCCL all-gather of sensordata per launch" if world > 1
                       else "single GPU", "state_finite": finite, "auto_resets": resets},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mjb_step_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if fp64:
            out["roofline"]["fp64"] = fp64  # second view: with K fused steps the kernel is VALU / latency bound
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, model, noise_std)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if gather:
        if force_gather and rank == 0:
            batch.synchronize()
            dist.all_gather_into_tensor(sens_all, sens_local)
            torch.cuda.synchronize()
            host = torch.from_numpy(batch.get("sensordata")).to(sens_all.device)
            ok = bool(torch.equal(sens_all[:E], host)) and bool(torch.isfinite(sens_all).all())
            print(f"forced single-rank gather: sensordata round trip {'ok' if ok else 'MISMATCH'}", file=sys.stderr)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
