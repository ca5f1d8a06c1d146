#!/usr/bin/env python3
"""stage_marginal.py -- marginal-cost profile of a SHIPPED constrained kernel.

A build with -DMJB_DOUBLE_STAGE=<id> (csrc/mjb_step.hip, MJB_REP) runs one stage twice -- every stage recomputes its outputs from its inputs, so the
rollout is unchanged -- and the launch-time difference to the production build is what that stage costs in THROUGHPUT: on the production register
allocation (the doubled stage is a two-trip loop around the same code), with the partner wavefront on the SIMD, over the whole batch.  The Gauss-Seidel
sweeps are not idempotent: id 23 appends MJB_EXTRA_SWEEPS sweeps behind the stop test (results move by ~1e-9), difference / that = one sweep, times the
workload's mean sweep count.  The windowed cycle probes (tools/profile_stages.py) measure ONE wavefront's latency and shift the 256-register allocation;
rocprofv3's PC sampling is not supported on this pool.

  python tools/stage_marginal.py build [config]      # on the build host: libmjb_xd<id>.so for every id (parallel hipcc)
  python tools/stage_marginal.py run [config]        # on the GPU box: bench line per build -> the table (JSON on the last line)
"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mujoco_ros_pkgs_amd", "csrc")
STAGES = [(0, "kinematics"), (1, "com_pos"), (2, "crb + factorM (+ M + hB)"), (16, "collision"), (17, "make_constraint + reference"), (4, "transmission + sens_pos"),
          (5, "com_vel"), (6, "passive"), (7, "rne"), (9, "actuation"), (10, "acceleration"), (20, "pgs: rows of J M^-1 (tri_build + tri_solve)"),
          (21, "pgs: b, A_ii, warmstart forces"), (22, "pgs: AR build"), (23, "pgs: Gauss-Seidel sweeps (from %d extra)"), (24, "pgs: J'f"), (25, "pgs: presolve (M^-1 J'f + Euler's solve)")]
EXTRA = 8
GROUP = {3: 5, 5: 3}   # bench config -> kernel slice (variant 9 / variant 4)


def build(cfg):
    def one(sid):
        cmd = [os.path.join(ROOT, "tools", "build_variant.sh"), f"d{sid}", str(GROUP[cfg]), f"-DMJB_DOUBLE_STAGE={sid}", f"-DMJB_EXTRA_SWEEPS={EXTRA}"]
        return sid, subprocess.run(cmd, capture_output=True, text=True).returncode
    with ThreadPoolExecutor(6) as ex:
        for sid, rc in ex.map(one, [s for s, _ in STAGES] + [99]):
            print("built", sid, "rc", rc, flush=True)


def bench(cfg, lib=None):
    env = dict(os.environ)
    if lib:
        env["MJB_LIBRARY"] = lib
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(cfg), "--no-cpu-baseline", "--no-other-configs"], env=env, capture_output=True, text=True)
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    return d["roofline"]["kernel_ms"], d["value"], d.get("workload_stats", {})


def run(cfg):
    base_ms, base_v, st = bench(cfg)
    base2, _, _ = bench(cfg)
    sweeps = st.get("solver_iters_mean", 0)
    rows, total = [], 0.0
    print(f"config {cfg}: production build {base_ms:.2f} ms per launch (repeat {base2:.2f}), {base_v / 1e6:.2f} M env-steps/s, mean sweeps {sweeps}")
    for sid, name in STAGES:
        lib = os.path.join(CSRC, f"libmjb_xd{sid}.so")
        if not os.path.exists(lib):
            continue
        ms, _, _ = bench(cfg, lib)
        d = ms - base_ms
        if sid == 23:
            per = d / EXTRA
            d = per * sweeps
            name = name % EXTRA + f": {per:.3f} ms per sweep x {sweeps:.1f}"
        total += d
        rows.append({"id": sid, "stage": name, "ms": round(d, 3), "share": round(d / base_ms, 4)})
        print(f"  {name:<70s} {d:8.2f} ms  {100 * d / base_ms:5.1f} %", flush=True)
    lib = os.path.join(CSRC, "libmjb_xd99.so")
    if os.path.exists(lib):   # the step loop WITHOUT its forward pass: ctrl noise, mj_check*, Euler's integration, the work queue, state load / store
        ms, _, _ = bench(cfg, lib)
        total += ms
        rows.append({"id": 99, "stage": "everything outside the forward pass (measured alone)", "ms": round(ms, 3), "share": round(ms / base_ms, 4)})
        print(f"  {'everything outside the forward pass (a build without it)':<70s} {ms:8.2f} ms  {100 * ms / base_ms:5.1f} %", flush=True)
    print(f"  {'sum of the marginal costs':<70s} {total:8.2f} ms  {100 * total / base_ms:5.1f} %")
    print(json.dumps({"config": cfg, "production_ms": base_ms, "production_repeat_ms": base2, "sum_ms": round(total, 3), "rows": rows}))


if __name__ == "__main__":
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    (build if sys.argv[1] == "build" else run)(cfg)
