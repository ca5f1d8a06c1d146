#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched step engine on BASELINE.json's configs[1]
("Franka Panda 9-DoF no-contact, 4096 envs, 1xMI355X fp64"), weak-scaled to N GPUs of one node.

One bench "step" = ONE fused kernel launch that advances every env of the rank's batch by ``--substeps`` physics
steps (rollouts of K = 1000 steps), driven by the reference's Ornstein-Uhlenbeck ctrl-noise injector
generated on device (counter-based Philox, seed 12345), followed by the engine's metrics reduction and -- when
N > 1 -- by the RCCL all-gather of ``sensordata`` + the 16-double metrics all-reduce over the node,
issued on a side stream so that they overlap the next launch.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|5] [--substeps S] [--envs E] [--lanes G]

With N > 1 and no torch.distributed environment the script spawns its own ranks (one process per GPU through
``python -m torch.distributed.run`` on 127.0.0.1); under torchrun it uses the environment it is given.
``--config`` selects the BASELINE workload: 2 = configs[1] (default, the headline metric), 3 = configs[2]
(Franka + table + cube, PGS), 5 = configs[4] (Shadow-Hand-like, Newton + elliptic cones, the high-contact POWER GRASP workload since
round 5; per-GPU shard of 1024 envs).  ``--model shadow_hand_like`` is the round 1 - 4 config-5 workload (cube resting in the half-open
palm, ~5 contacts / ~20 rows), reported as ``5_light``.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = {"franka_like": 712, "franka_table": 1072, "shadow_hand_like": 2136, "shadow_hand_grasp": 2136}  # 8*(2nq+2nv+2nv+nu+S)+8
# per-model workload: (BASELINE config label, OU ctrl-noise std [= 0.5 * ctrlrange of the big actuators], default envs per GPU,
#                      default physics steps per launch, BASELINE config number)
WORKLOADS = {
    "franka_like": ("BASELINE configs[1]: Franka-Panda-like 9-DoF arm, no contacts", 0.5 * 87.0, 4096, 1000, 2),
    # (1000 steps per launch since r03, like configs[1]: a fused launch ends with its SLOWEST env's serial chain of steps -- envs
    #  pressing into the table run PGS to its 100-sweep cap for a while, ~300 us per step -- and at 200 steps that tail set the
    #  launch time; per-launch series and the 200 / 500 / 1000 sweep: profiles/r03_cfg3_launch_length.txt, tools/launch_series.py)
    "franka_table": ("BASELINE configs[2]: Franka-like arm + table + cube contacts", 0.5 * 87.0, 4096, 1000, 3),
    # (1000 steps per launch like the other configs since the four-envs-per-CU frame: 1024 envs = 1024 slots, every env has a slot of
    #  its own and a launch lasts as long as its slowest env's chain -- over 1000 steps the chains' sums even out: 4.7 M at 100 steps
    #  per launch, 5.0 M at 1000.  Slots x launch length x chunking: profiles/r03_cfg5_residency.txt)
    # (round 5, VERDICT r04 #1: configs[4] says "high contact count" -- the power grasp of tools/gen_hand_model.py: fingers servoed
    #  closed over the cube, mean ncon ~21, mean nefc ~87, p99 ~115, ~85 % of the env-steps beyond 64 rows, i.e. on the 4-rows-per-lane
    #  Newton solver over the env's HBM row block; every constrained line carries the measured ncon / nefc statistics)
    "shadow_hand_grasp": ("BASELINE configs[4]: Shadow-Hand-like 24-DoF hand, cube held in a power grasp (high contact count; Newton, elliptic cones)", 0.1, 1024, 1000, 5),
    # (the round 1 - 4 workload of config 5: the cube RESTS in the half-open palm, mean ncon 4 - 5, ~20 rows -- kept as "5_light")
    "shadow_hand_like": ("BASELINE configs[4] LIGHT: Shadow-Hand-like 24-DoF hand, cube resting in the half-open palm (Newton, elliptic cones)", 0.1, 1024, 1000, 5),
}
CONFIG_MODEL = {2: "franka_like", 3: "franka_table", 4: "franka_table", 5: "shadow_hand_grasp"}
# key of a workload in `other_configs` and in the profiles/ file names (rNN_cfg<tag>_*)
WORKLOAD_TAG = {"franka_like": "2", "franka_table": "3", "shadow_hand_grasp": "5", "shadow_hand_like": "5_light"}
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
    if name == "shadow_hand_grasp":
        from mujoco_ros_pkgs_amd import workloads
        return workloads.hand_power_grasp_states(model, nenv, seed)
    return synthetic_state(model, nenv, seed)


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_oracle(name, model, noise_std, nenv, threads, target_s, pin=False):
    """Env-steps/s of oracle/libmjo_fast.so on `threads` host threads over ~target_s seconds of the same workload."""
    from oracle import pyoracle
    qpos, qvel = initial_state(name, model, nenv, seed=999)
    kw = dict(noise_std=noise_std, noise_rate=0.1, seed=12345, nthreads=threads, fast=True)
    old_aff = None
    if pin and hasattr(os, "sched_getaffinity"):
        old_aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {sorted(old_aff)[0]})
    try:
        nsteps = 100
        t0 = time.perf_counter()
        pyoracle.rollout(model, qpos, qvel, nsteps, **kw)
        rate = nenv * nsteps / (time.perf_counter() - t0)
        nsteps2 = int(min(max(nsteps, rate * target_s / nenv), 200000))
        t0 = time.perf_counter()
        pyoracle.rollout(model, qpos, qvel, nsteps2, **kw)
        dt = time.perf_counter() - t0
    finally:
        if old_aff is not None:
            os.sched_setaffinity(0, old_aff)
    return nenv * nsteps2 / dt, nsteps2, dt


def cpu_baseline(name, model, noise_std, with_mujoco=True):
    """CPU figures beside the GPU number, all on a bounded sample of the same workload:
    the oracle ("port": from-scratch restatement, gcc -O3 -march=native) on all host threads and on ONE pinned thread,
    and real MuJoCo through $MUJOCO_DIR when that library exists on the box (else the literal NOT MEASURED)."""
    from oracle import pyoracle
    pyoracle.build()
    cores = os.cpu_count() or 1
    v_all, n_all, dt_all = _time_oracle(name, model, noise_std, 4 * cores, cores, 10.0)
    v_one, n_one, dt_one = _time_oracle(name, model, noise_std, 4, 1, 6.0, pin=True)
    out = {"value": v_all, "unit": "env-steps/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_string(),
           "sample": f"{4 * cores} envs x {n_all} steps of the same {name} workload (OU ctrl noise), oracle/libmjo_fast.so "
                     f"(gcc -O3 -march=native), {cores} threads, {dt_all:.1f} s",
           "single_thread": {"value": v_one, "unit": "env-steps/s", "cores": 1,
                             "sample": f"4 envs x {n_one} steps, one thread pinned to one core, {dt_one:.1f} s"}}
    if not with_mujoco:
        return out
    try:
        from oracle import mujoco_ref
        out["mujoco"] = mujoco_ref.time_reference(name, noise_std)
    except Exception as exc:  # the leg is optional by construction: never let it take the bench line down
        out["mujoco"] = f"NOT MEASURED ({type(exc).__name__}: {exc})"
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _load_json(*names):
    for n in names:
        p = os.path.join(ROOT, "profiles", n)
        if os.path.exists(p):
            try:
                return json.load(open(p)), n
            except Exception:
                pass
    return None, None


def roofline_block(name, tag, solver_tag, E, S, samples, kernel="mjb_step_kernel"):
    """`roofline` object of one workload: algorithmic bytes per launch / median kernel time against the HBM peak, the
    counter-measured HBM traffic and executed fp64 work when profiles/ holds a PMC summary collected on THESE kernel sources
    (fingerprint check: mujoco_ros_pkgs_amd/provenance.py) at this (envs, substeps), and the useful fp64 rate from the
    oracle's operation count."""
    from mujoco_ros_pkgs_amd import provenance
    kern_ms = samples[len(samples) // 2]
    traffic, fp64, source = None, None, None
    # (the newest round's counter summary of this config; one collected on THESE kernel sources wins)
    import glob
    cands = sorted((os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_cfg{tag}{solver_tag}_pmc_summary.json"))), reverse=True)
    pmc, pmc_file = _load_json(*cands) if cands else (None, None)
    for n in cands:
        cand, _ = _load_json(n)
        if cand is not None and cand.get("csrc_sha") == provenance.csrc_sha():
            pmc, pmc_file = cand, n
            break
    flops, flops_file = None, None
    for fn in ("r05_oracle_flops.json", "r04_oracle_flops.json", "r03_oracle_flops.json", "r02_oracle_flops.json"):  # newest file that has this workload
        cand, _ = _load_json(fn)
        if cand and name in cand:
            flops, flops_file = cand, fn
            break
    if pmc is not None:
        if pmc.get("csrc_sha") != provenance.csrc_sha():
            source = f"STALE: profiles/{pmc_file} was collected on other kernel sources ({pmc.get('csrc_sha')} != {provenance.csrc_sha()}); not reported"
        elif (E, S) != (pmc.get("envs"), pmc.get("substeps")):
            source = f"profiles/{pmc_file} holds ({pmc.get('envs')} envs, {pmc.get('substeps')} steps per launch), not this run's; not reported"
        else:
            try:
                traffic = (pmc["FETCH_SIZE"]["mean_per_dispatch"] + pmc["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
                source = f"profiles/{pmc_file}"
                if "fp64_executed_flops_per_dispatch" in pmc:
                    fl = pmc["fp64_executed_flops_per_dispatch"]
                    peak = pmc.get("fp64_peak_measured", {}).get("fp64_fma_tflops", FP64_PEAK_TFLOPS)
                    ach = fl / (kern_ms * 1e-3) / 1e12
                    fp64 = {"bound": "fp64-valu", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                            "executed_flops_per_launch": fl,
                            "note": f"SQ_INSTS_VALU_{{ADD,MUL,FMA,TRANS}}_F64 x 64 lanes, idle lanes included (profiles/{pmc_file}); "
                                    "peak = tools/ubench/fp64_peak on the same box"}
            except Exception:
                traffic, fp64 = None, None
    if flops and name in flops:  # the oracle's instrumented operation count (THE flop figure)
        per = float(flops[name]["flops_per_env_step"])
        ach = per * E * S / (kern_ms * 1e-3) / 1e12
        fp64 = fp64 or {"bound": "fp64-valu", "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s"}
        fp64.update(algorithmic_flops_per_env_step=per, useful_achieved=ach, useful_frac=ach / fp64["peak"],
                    useful_note=f"oracle op counter, add / mul / div / sqrt = 1, fma = 2 (profiles/{flops_file})")
    bytes_per_launch = ALGO_BYTES_PER_ENV_STEP.get(name, 712) * E * S
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_source": source, "kernel": kernel, "kernel_ms": kern_ms,
           "kernel_ms_samples": {"n": len(samples), "median": kern_ms, "min": samples[0], "max": samples[-1]},
           "algorithmic_bytes_per_launch": bytes_per_launch}
    if fp64:
        out["fp64"] = fp64  # second view: with K fused steps the kernel is VALU / latency bound
    return out


def workload_stats(batch):
    """ncon / nefc / solver-iteration statistics of the env-steps the batch ran since set_stats(True) (device-side counters,
    mjb_set_stats; VERDICT r04 #1: a reader must be able to tell what the contact benchmark asked of the solver)."""
    st = batch.stats()
    return {"env_steps_counted": st["evaluations"], "ncon_mean": round(st["ncon_mean"], 3), "ncon_p50": st["ncon_p50"], "ncon_p99": st["ncon_p99"],
            "ncon_max": st["ncon_max"], "nefc_mean": round(st["nefc_mean"], 3), "nefc_p50": st["nefc_p50"], "nefc_p99": st["nefc_p99"],
            "nefc_max": st["nefc_max"], "rows_gt64_share": round(st["rows_gt64_share"], 5), "solver_iters_mean": round(st["solver_iter_mean"], 3)}


def kernel_form(batch):
    """Which kernel the batch's fused launches ran: the lane = env form (one env per lane, csrc/mjb_lane_env.hip) or the generic one."""
    if not batch.lane_env_info()[1]:
        return "mjb_step_kernel", "generic (G lanes per env, frame in LDS)"
    form = batch.lane_env_last_form()
    kname = ("mjb_lane_env_kernel", "mjb_lane_env_duo_kernel", "mjb_lane_env_duo2_kernel", "mjb_lane_env_trio_kernel")[form if form in (0, 1, 2, 3) else 0]
    how = ("one wavefront per 64 envs", "two wavefronts per 64 envs: position half | velocity half",
           "two wavefronts per 64 envs, pipelined body by body through LDS: poses | cinert, cdof, velocities, forces",
           "three wavefronts per 64 envs, pipelined body by body through LDS: poses | cinert, cdof, inertias, factors, solves, Euler | velocities, forces")[form if form in (0, 1, 2, 3) else 0]
    return kname, f"lane = env (one env per lane, {how}; sensordata evaluated at the last step of a launch)"


def measure_other_config(name, device, launches=5, with_cpu=True, envs=None, substeps=None, lane_env=None, tag=None, sens_every=False, split=None):
    """One of the other BASELINE workloads (franka_table = configs[2], shadow_hand_grasp = configs[4]'s per-GPU shard, shadow_hand_like =
    its light predecessor), measured in this process after the headline timing so that the DRIVER's record carries it (VERDICT r02
    #3a): `launches` timed fused launches bracketed by synchronisation (five, like the standalone `--config N` run: config 3's first
    launches after the synthetic start are its heaviest, three of them read 2 % low), then five single-launch kernel samples; the
    device-side ncon / nefc / iteration counters run over exactly the timed launches; the CPU oracle legs on the same workload."""
    from mujoco_ros_pkgs_amd import engine, mjcf
    label, noise_std, E, S, cfgno = WORKLOADS[name]
    E, S = envs or E, substeps or S
    model = mjcf.Model(dict(mjcf.load_asset(name)))
    model["enableflags"] = int(model["enableflags"]) | 2
    cm = engine.CompiledModel(model)
    batch = engine.Batch(cm, E, device)
    if lane_env is not None:
        batch.set_lane_env(lane_env)
    if sens_every:
        batch.set_sensors_every_step(True)
    if split is not None:
        batch.set_split_step(split)
    qpos, qvel = initial_state(name, model, E, seed=1000)
    batch.set("qpos", qpos)
    batch.set("qvel", qvel)
    batch.set_ctrl_noise(noise_std, 0.1, 12345, 0)
    batch.step(S)
    batch.synchronize()
    constrained = int(model["nefcmax"]) > 0
    if constrained:
        batch.set_stats(True)
    t0 = time.perf_counter()
    for _ in range(launches):
        batch.step(S)
    batch.synchronize()
    elapsed = time.perf_counter() - t0
    stats = workload_stats(batch) if constrained else None
    if constrained:
        batch.set_stats(False)
    noise_mode = batch.noise_mode()
    finite = bool(np.all(np.isfinite(batch.get("qpos"))))
    samples = sorted(batch.time_steps(S, 1) for _ in range(5))
    kname, kform = kernel_form(batch)
    if sens_every and batch.lane_env_info()[1]:
        kform = kform.replace("sensordata evaluated at the last step of a launch", "sensordata evaluated at EVERY step: mjb_set_sensors_every_step")
    if batch.split_step_info()[1]:
        kname = "mjb_cstep_kernel"
        kform = (f"split step: smooth stages one env per lane (mjb_smooth_kernel) + constraint stages one env per wavefront (mjb_cstep_kernel), a kernel pair "
                 f"per step on {batch.split_step_info()[2]} env slices; kernel_ms is the whole launch")
    out = {"metric": "env_steps_per_sec", "value": E * S * launches / elapsed, "unit": "env-steps/s", "n_gpus": 1,
           "steps": launches, "warmup": 1, "ms_per_step": 1e3 * elapsed / launches,
           "config": {"workload": f"{label}, {E} envs per GPU, fp64, Euler dt={model['timestep'][0]}", "baseline_config": cfgno,
                      "envs_per_gpu": E, "physics_steps_per_launch": S, "model": name,
                      "solver": {0: "PGS", 1: "CG", 2: "Newton"}[int(model["solver"])] if constrained else "none",
                      "nconmax": int(model["nconmax"]), "nefcmax": int(model["nefcmax"]), "kernel_form": kform,
                      "fused_frame_bytes": batch.fused_frame()[1], "fused_frame": ("default", "default", "wide (up to 128 rows in LDS, two per lane; as many as let three frames share a CU)")[batch.fused_frame()[0]],
                      "noise_pregen": noise_mode,
                      "state_finite": finite, "auto_resets": batch.warning_count(),
                      # (mjWARN_CONTACTFULL / mjWARN_CNSTRFULL events of every env-step this batch ran, warm-up included, and their rate:
                      #  config 3's 16-contact capacity -- SURVEY.md §8's table, what keeps eight lean frames per CU -- is exceeded
                      #  by a few env-steps per million; the oracle truncates identically.  tools/overflow_rate.py,
                      #  profiles/r04_cfg3_overflow.txt; bounded in tests/test_gpu_full_size.py)
                      "contactfull": batch.warning("contactfull"), "cnstrfull": batch.warning("cnstrfull"),
                      "overflow_per_env_step": (batch.warning("contactfull") + batch.warning("cnstrfull")) / float(E * S * (launches + 6))},
           "roofline": roofline_block(name, tag or WORKLOAD_TAG[name], "", E, S, samples, kname)}
    if stats is not None:
        out["workload_stats"] = stats
    batch.close()
    cm.close()
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(name, model, noise_std, with_mujoco=False)
    return out


# config 2 beside its headline line: the generic 16-lanes-per-env kernel on the same 4096 envs, and the lane = env kernel on a batch that
# fills the chip with 64-env wavefronts (report both) -- (tag, envs, steps per launch, mjb_set_lane_env mode)
CONFIG2_EXTRAS = (("2_generic_kernel", 4096, 1000, 0), ("2_lane_env_65536", 65536, 200, 1))
# ... the headline kernel with the sensor stages evaluated at every step (A15's per-step cost on that kernel as a number), and config 3's
# model on ONE GPU at 32 768 envs, fused kernel against the split step (round 6: smooth stages one env per lane, profiles/r06_split_step.txt)
R06_EXTRAS = (("2_sensors_every_step", "franka_like", dict(envs=4096, substeps=1000, sens_every=True)),
              ("3_32768_fused", "franka_table", dict(envs=32768, substeps=500, launches=3, split=0)),
              ("3_32768_split", "franka_table", dict(envs=32768, substeps=500, launches=3, split=1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="BASELINE config number (2 = configs[1], the headline; 3; 5)")
    ap.add_argument("--substeps", type=int, default=0, help="physics steps fused into one launch (0 = the config's own)")
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (weak scaling); 0 = the config's own (4096; 1024 for the hand)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per env (0 = engine default)")
    ap.add_argument("--lane-env", choices=("auto", "on", "off"), default="auto",
                    help="the lane = env form of the unconstrained fused step (mjb_set_lane_env): auto = the engine's rule (batches of >= 4096 envs "
                         "of a model whose topology is compiled in)")
    ap.add_argument("--epb", type=int, default=0, help="envs per workgroup (0 = engine default)")
    ap.add_argument("--model", default="")
    ap.add_argument("--solver", default="", choices=["", "PGS", "Newton"], help="override the model's constraint solver")
    ap.add_argument("--nefcmax", type=int, default=0, help="override the model's constraint-row capacity")
    ap.add_argument("--nconmax", type=int, default=0, help="override the model's contact capacity")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs 3 / 5 lines a default single-GPU run appends")
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="seconds a collective may wait for a missing rank before the run fails with an error line")
    ap.add_argument("--dry-ranks", type=int, default=0, help="CI: run the N-rank control flow (rendezvous, overlapped exchange, fence, max-over-ranks "
                    "timing, the rank-0 line) under gloo with a stub batch and thread streams -- no GPU, the value means nothing")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help="(with --dry-ranks) this rank raises inside the timed region: exercises the error line")
    args = ap.parse_args()
    if args.dry_ranks:
        args.gpus = args.dry_ranks
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    name = args.model or CONFIG_MODEL[args.config or 2]

    # ---- one process per GPU: spawn the ranks ourselves when nobody did (bare `python bench.py --gpus N`)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # (also --dry-ranks N)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    if args.dry_ranks:
        sys.exit(guarded(dry_run, args))
    sys.exit(guarded(gpu_run, args, name))


def emit_line(fd, obj):
    os.write(fd, (json.dumps(obj) + "\n").encode())


def guarded(fn, *a):
    """Run one rank's body; whatever goes wrong -- this rank's own exception, a collective that timed out on a rank that never
    arrived, the launcher's SIGTERM after another rank died -- ends in ONE JSON line with an "error" key on the real stdout of the
    job's speaker, rank 0 (a failing rank > 0 leaves its message in a note rank 0 quotes), and a non-zero exit code, never in a hang
    inside dist.barrier()."""
    import signal
    rank = int(os.environ.get("RANK", "0"))
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # (RCCL / gloo print banners through C stdio: everything but the line goes to stderr)
    state = {"done": False}

    note = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mjb_bench_error_%s.txt" % os.environ.get("MASTER_PORT", "0"))  # a failing rank > 0 leaves its message here
    if rank == 0 and os.path.exists(note):
        os.remove(note)

    def peer_note():
        try:
            time.sleep(0.3)
            return open(note).read()[:2000]
        except OSError:
            return ""

    def on_term(signum, frame):
        if not state["done"] and rank == 0:
            emit_line(real_stdout, {"error": ("terminated by the launcher (signal %d): another rank failed or the job was cancelled. " % signum) + peer_note(), "rank": rank})
        os._exit(1)
    signal.signal(signal.SIGTERM, on_term)
    try:
        rc = fn(args_with_stdout(a[0], real_stdout), *a[1:])
        state["done"] = True
        return rc or 0
    except SystemExit as e:
        state["done"] = True
        return e.code
    except BaseException as exc:  # noqa: BLE001 -- the contract is a line, not a traceback alone
        import traceback
        traceback.print_exc(file=sys.stderr)
        collective = type(exc).__name__ in ("DistBackendError", "DistNetworkError", "DistStoreError") or "imeout" in str(exc) or "Connection closed by peer" in str(exc)
        if rank == 0:
            emit_line(real_stdout, {"error": (f"{type(exc).__name__}: {exc} " + (peer_note() if collective else ""))[:2000], "rank": rank,
                                    "n_gpus": int(os.environ.get("WORLD_SIZE", "1"))})
        elif not collective:  # (rank 0 speaks for the job: it reads this when its collective fails or the launcher stops it)
            try:
                with open(note, "w") as fh:
                    fh.write(f"rank {rank}: {type(exc).__name__}: {exc}")
            except OSError:
                pass
        state["done"] = True
        os._exit(1)   # (not sys.exit: a process group half torn down can hang in its destructor)


def args_with_stdout(args, fd):
    args.real_stdout = fd
    return args


def timed_region(args, one_step, fence, dist, world, make_tensor):
    """The contract's timing: W untimed steps, fence, exactly K steps, fence; the MAX over ranks is the job's time, and every rank's
    own time is gathered so that a weak-scaling loss can be attributed (straggler vs exchange)."""
    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step()
    fence()
    mine = time.perf_counter() - t0
    elapsed, per_rank = mine, [mine]
    if world > 1:
        t = make_tensor([mine])
        gathered = [make_tensor([0.0]) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [float(g.item()) for g in gathered]
        tmax = make_tensor([mine])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed, per_rank


def dry_run(args):
    """--dry-ranks N: every line of the N-rank control flow that does not need a GPU -- rendezvous with a timeout, shard ranges, the
    real OverlappedExchange (thread streams), barrier + synchronise fences, all-gather of the per-rank times, MAX over ranks, the
    rank-0 line -- with a stub batch whose "launch" writes rank- and launch-dependent sensordata."""
    import datetime
    import torch
    import torch.distributed as dist
    from mujoco_ros_pkgs_amd import sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.dist_timeout))
    if dist.get_world_size() != args.gpus:  # (the same rule as gpu_run: the line reports the process group, and fails when it is not the one asked for)
        raise RuntimeError(f"--dry-ranks {args.gpus} but the process group has {dist.get_world_size()} rank(s) (WORLD_SIZE={world})")
    world = dist.get_world_size()
    E, S, nsd = 64, args.substeps or 10, 8
    env_lo, _ = sharding.shard_range(rank, world, E)
    rt = sharding.ThreadStreams()
    sens = torch.zeros(E, nsd, dtype=torch.float64)
    met = torch.zeros(16, dtype=torch.float64)
    xch = sharding.OverlappedExchange(sens, met, None, torch.device("cpu"), force=True, streams=rt)
    launches = {"n": 0}

    def launch():
        launches["n"] += 1
        k = launches["n"]
        time.sleep(0.002)
        sens.copy_(torch.full((E, nsd), float(1000 * k + rank)))
        met[0] = float(E * S * k)   # env_steps so far (SUM over ranks)
        met[8] = float(rank)        # a MAX entry

    def one_step():
        if rank == args.dry_fail_rank and launches["n"] >= args.warmup:
            raise RuntimeError(f"dry run: injected failure on rank {rank}")
        rt.enqueue(rt.eng, launch)
        xch.issue()

    def fence():
        dist.barrier()
        rt.synchronize(rt.eng)
        rt.synchronize(rt.side)

    elapsed, per_rank = timed_region(args, one_step, fence, dist, world, lambda v: torch.tensor(v, dtype=torch.float64))
    sens_all, m = xch.finish()
    total = args.warmup + args.steps
    ok = all(bool((sens_all[r * E:(r + 1) * E] == float(1000 * total + r)).all()) for r in range(world))
    ok = ok and float(m[0]) == float(world * E * S * total) and float(m[8]) == float(world - 1)
    if rank == 0:
        emit_line(args.real_stdout, {
            "metric": "env_steps_per_sec", "value": world * E * S * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "DRY RUN: stub batch, gloo, thread streams -- no GPU, the value means nothing",
            "dry_ranks": world, "exchange_ok": ok, "rank_ms_per_step": {"min": 1e3 * min(per_rank) / args.steps, "max": 1e3 * max(per_rank) / args.steps},
            "exchange_ms": xch.last_ms(), "cpu_baseline_leg": world == 1 and not args.no_cpu_baseline,  # (gpu_run's guard: rank 0 at N = 1 only)
            "config": {"workload": "dry run", "rccl_ranks": 0, "gloo_ranks": dist.get_world_size(), "shard": [env_lo, env_lo + E]}})
    dist.barrier()
    rt.close()
    dist.destroy_process_group()
    return 0 if ok else 4


def gpu_run(args, name):
    import datetime
    import torch
    import torch.distributed as dist

    real_stdout = args.real_stdout
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running with the {world} rank(s) of the environment",
              file=sys.stderr)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the engine has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if local_rank >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} are visible", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_gather = os.environ.get("MJB_BENCH_FORCE_GATHER", "0") == "1"  # exercise the RCCL path on one GPU
    # The contract is ONE JSON line on stdout: RCCL prints a version banner through C stdio (flushed at exit, i.e. after
    # anything Python prints), so everything written to fd 1 from here on goes to stderr and the line is written to the
    # saved descriptor.
    sys.stdout.flush()
    os.dup2(2, 1)
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # (a rank that never arrives fails the collectives of the others after --dist-timeout instead of hanging them: guarded())
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.dist_timeout))
        # (the line's n_gpus / rccl_ranks are what the process group SAYS, not what the environment promised: a launcher that started
        #  fewer ranks than --gpus must not produce a line that looks like an N-GPU measurement)
        if dist.get_world_size() != args.gpus and not force_gather:
            raise RuntimeError(f"--gpus {args.gpus} but the RCCL process group has {dist.get_world_size()} rank(s) (WORLD_SIZE={world})")
        world = dist.get_world_size()

    from mujoco_ros_pkgs_amd import binding, engine, mjcf, sharding

    model = mjcf.load_asset(name, **({"nefcmax": args.nefcmax} if args.nefcmax else {}), **({"nconmax": args.nconmax} if args.nconmax else {}))
    model = mjcf.Model(dict(model))
    model["enableflags"] = int(model["enableflags"]) | 2  # mjENBL_ENERGY: the metrics vector carries the energies
    if args.solver:
        model["solver"] = {"PGS": 0, "Newton": 2}[args.solver]
    cm = engine.CompiledModel(model)
    label, noise_std, default_envs, default_sub, cfgno = WORKLOADS.get(name, (name, 1.0, 4096, 200, 0))
    E, S = (args.envs or default_envs), (args.substeps or default_sub)
    batch = engine.Batch(cm, E, local_rank)
    batch.set_launch(args.lanes, args.epb)
    batch.set_lane_env({"auto": -1, "on": 1, "off": 0}[args.lane_env])
    qpos, qvel = initial_state(name, model, E, seed=1000 + rank)
    batch.set("qpos", qpos)
    batch.set("qvel", qvel)
    # reference injector: tau = 0.1 s, std = 0.5 * ctrlrange (87 N m on the big joints), seed 12345
    env_lo, _ = sharding.shard_range(rank, world, E)
    batch.set_ctrl_noise(noise_std, 0.1, 12345, env_lo)
    batch.synchronize()

    nsd = model["nsensordata"]
    sens_local = torch.as_tensor(DevArray(batch.device_ptr("sensordata"), (E, nsd)), device=dev)
    metrics_local = torch.as_tensor(DevArray(batch.metrics_device_ptr(), (16,)), device=dev)
    xch = sharding.OverlappedExchange(sens_local, metrics_local, batch.stream, dev, force=force_gather)

    def one_step():
        batch.step(S)                # K fused physics steps, asynchronous on the engine's stream
        batch.metrics_device_ptr()   # the 16-double metrics reduction, same stream
        xch.issue()                  # staging copy (engine stream) + all-gather / all-reduce (side stream): overlaps the next launch

    def fence():
        if xch.active:
            dist.barrier()
        torch.cuda.synchronize()

    constrained = int(model["nefcmax"]) > 0
    if constrained:
        batch.set_stats(True)    # device-side ncon / nefc / iteration counters over everything this rank runs from here on (warm-up included)
    elapsed, per_rank = timed_region(args, one_step, fence, dist, world, lambda v: torch.tensor(v, dtype=torch.float64, device=dev))
    sens_all, met = xch.finish()
    noise_mode = batch.noise_mode()
    noise_modes = [noise_mode]
    if world > 1:  # every rank's generator mode: eight side-stream generators + RCCL kernels compete for the CUs the step kernel fills
        code = torch.tensor([("in-kernel", "same-stream", "side-stream").index(noise_mode)], dtype=torch.int32, device=dev)
        codes = [torch.zeros_like(code) for _ in range(world)]
        dist.all_gather(codes, code)
        noise_modes = [("in-kernel", "same-stream", "side-stream")[int(c.item())] for c in codes]
    stats = None
    if constrained:
        stats = workload_stats(batch)
        batch.set_stats(False)
    met = met.cpu().numpy()
    metrics = dict(zip(binding.METRIC_NAMES, (float(x) for x in met)))

    # sanity: state still finite (non-finite envs are auto-reset and counted by the engine)
    finite = bool(np.all(np.isfinite(batch.get("qpos")))) and bool(torch.isfinite(sens_all).all())
    resets = batch.warning_count()  # mj_check* auto-resets of THIS rank (must be 0 on config 2)

    # dominant-kernel duration measured with HIP events on the engine's own stream: 5 single-launch samples
    samples = sorted(batch.time_steps(S, 1) for _ in range(5))
    kern_ms = samples[len(samples) // 2]
    kname, kform = kernel_form(batch)

    if rank == 0:
        solver_tag = "" if not args.solver else "_" + args.solver.lower()  # (the counters belong to ONE kernel variant)
        value = world * E * S * args.steps / elapsed
        out = {
            "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{label}, {E} envs per GPU, fp64, Euler dt={model['timestep'][0]}",
                       "baseline_config": cfgno, "envs_per_gpu": E, "physics_steps_per_launch": S, "model": name,
                       "solver": {0: "PGS", 1: "CG", 2: "Newton"}[int(model["solver"])] if model["nefcmax"] else "none",
                       "nconmax": int(model["nconmax"]), "nefcmax": int(model["nefcmax"]), "kernel_form": kform,
                       "ctrl": f"on-device OU noise (Philox seed 12345, tau 0.1 s, std {noise_std:g})",
                       "parallelism": f"env-sharded x{world}, RCCL all-gather of sensordata + 16-double metrics all-reduce per launch, "
                                      "side stream (overlaps the next launch)" if world > 1 else "single GPU",
                       "rccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1) if xch.active else 0,
                       "noise_pregen": noise_mode if len(set(noise_modes)) == 1 else noise_modes, "state_finite": finite, "auto_resets": resets},
            # (attribution of a weak-scaling loss: the slowest and the fastest rank's own time per step, and the duration of the last
            #  exchange -- staging wait + all-gather + two all-reduces -- measured with an event pair on the side stream)
            "rank_ms_per_step": {"min": 1e3 * min(per_rank) / args.steps, "max": 1e3 * max(per_rank) / args.steps},
            "exchange_ms": xch.last_ms(),
            "metrics": metrics,
            "roofline": roofline_block(name, WORKLOAD_TAG.get(name, str(cfgno)), solver_tag, E, S, samples, kname),
        }
        if stats is not None:
            out["workload_stats"] = stats
            out["config"]["contactfull"], out["config"]["cnstrfull"] = batch.warning("contactfull"), batch.warning("cnstrfull")
        out["config"]["fused_frame_bytes"] = batch.fused_frame()[1]
        out["config"]["fused_frame"] = ("default", "default", "wide (up to 128 rows in LDS, two per lane; as many as let three frames share a CU)")[batch.fused_frame()[0]]
        default_run = world == 1 and not (args.config or args.model or args.solver or args.nefcmax or args.nconmax or args.envs or
                                          args.substeps or args.lanes or args.epb or args.lane_env != "auto")
        if default_run and not args.no_other_configs:
            # the other BASELINE workloads, same process, same box (configs[2] and the per-GPU shard of configs[4])
            out["other_configs"] = {}
            for xtag, xe, xs, xmode in CONFIG2_EXTRAS:
                try:
                    out["other_configs"][xtag] = measure_other_config("franka_like", local_rank, with_cpu=False, envs=xe, substeps=xs, lane_env=xmode, tag=xtag)
                except Exception as exc:
                    out["other_configs"][xtag] = {"error": f"{type(exc).__name__}: {exc}"}
            for xtag, xname, kw in R06_EXTRAS:
                try:
                    out["other_configs"][xtag] = measure_other_config(xname, local_rank, with_cpu=False, tag=xtag, **kw)
                except Exception as exc:
                    out["other_configs"][xtag] = {"error": f"{type(exc).__name__}: {exc}"}
            for other in ("franka_table", "shadow_hand_grasp", "shadow_hand_like"):
                try:
                    out["other_configs"][WORKLOAD_TAG[other]] = measure_other_config(other, local_rank, with_cpu=not args.no_cpu_baseline)
                except Exception as exc:  # never let an extra take the headline line down
                    out["other_configs"][WORKLOAD_TAG[other]] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name, model, noise_std)
        emit_line(real_stdout, out)
    if xch.active:
        if force_gather and rank == 0:
            host = torch.from_numpy(batch.get("sensordata")).to(dev)
            one_step()
            sa, _ = xch.finish()
            torch.cuda.synchronize()
            host = torch.from_numpy(batch.get("sensordata")).to(dev)
            ok = bool(torch.equal(sa[:E], host)) and bool(torch.isfinite(sa).all())
            print(f"forced single-rank gather: sensordata round trip {'ok' if ok else 'MISMATCH'}", file=sys.stderr)
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    main()
