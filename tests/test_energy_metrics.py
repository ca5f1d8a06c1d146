"""mjData.energy (mjENBL_ENERGY) and the 16-double metrics vector of mjb_metrics (steps, resets, max |qacc|,
energy -- what the RCCL all-reduce carries)."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

PENDULUM = """
<mujoco><compiler angle="radian"/><option timestep="0.001" gravity="0 0 -9.81"><flag energy="enable"/></option>
<worldbody><body pos="0 0 1"><joint name="h" type="hinge" axis="0 1 0" stiffness="3" springref="0.2"/>
<inertial pos="0 0 -0.5" mass="2" diaginertia="0.1 0.1 0.1"/></body></worldbody></mujoco>
"""


def test_oracle_energy_closed_form(oracle_built):
    model = mjcf.compile_xml_string(PENDULUM)
    assert model["enableflags"] & 2
    d = oracle_built.OracleData(model)
    th, w = 0.7, 1.3
    d.qpos[0] = th
    d.qvel[0] = w
    d.forward()
    # com at 1 - 0.5 cos(th) above the ground; rotation about y by th: z = 1 - 0.5 cos(th)
    pe = 2 * 9.81 * (1 - 0.5 * np.cos(th)) + 0.5 * 3 * (th - 0.2) ** 2
    ke = 0.5 * (0.1 + 2 * 0.25) * w * w
    assert abs(d.energy[0] - pe) < 1e-12 and abs(d.energy[1] - ke) < 1e-12
    # without the flag the field stays zero
    off = mjcf.compile_xml_string(PENDULUM.replace('<flag energy="enable"/>', ""))
    d2 = oracle_built.OracleData(off)
    d2.qpos[0] = th
    d2.forward()
    assert d2.energy[0] == 0 and d2.energy[1] == 0


def test_oracle_energy_drift_is_first_order(oracle_built):
    """No damping, no spring: semi-implicit Euler keeps the total energy within O(dt) of its start."""
    model = mjcf.compile_xml_string(PENDULUM.replace('stiffness="3" springref="0.2"', ""))
    d = oracle_built.OracleData(model)
    d.qpos[0] = 1.0
    d.forward()
    e0 = d.energy.sum()
    d.step(2000)
    assert abs(d.energy.sum() - e0) < 0.02 * abs(e0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["franka_like", "franka_table", "shadow_hand_like"])
def test_gpu_energy_and_metrics_match_oracle(oracle_built, name):
    from bench import initial_state
    from mujoco_ros_pkgs_amd import engine
    model = mjcf.Model(dict(mjcf.load_asset(name)))
    model["enableflags"] = 2
    nenv, K = 12, 7
    qpos, qvel = initial_state(name, model, nenv, seed=3)
    b = engine.Batch(engine.CompiledModel(model), nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set_ctrl_noise(1.0, 0.1, 12345, 0)
    b.step(K)
    en = b.get("energy")
    d = oracle_built.OracleData(model)
    ref = np.zeros((nenv, 2))
    qacc = np.zeros((nenv, model["nv"]))
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        for s in range(K):
            d.ctrl_noise(1.0, 0.1, 12345, e, s)
            d.step(1)
        ref[e] = d.energy   # energy of the LAST step's forward pass (state before its integration), as in mjData after mj_step
        qacc[e] = d.qacc
    assert np.allclose(en, ref, rtol=1e-7, atol=1e-8)
    m, raw = b.metrics()
    assert m["env_steps"] == nenv * K and m["nenv"] == nenv and m["auto_resets"] == 0
    assert abs(m["energy_potential"] - ref[:, 0].sum()) <= 1e-6 * (1 + abs(ref[:, 0].sum()))
    assert abs(m["energy_kinetic"] - ref[:, 1].sum()) <= 1e-6 * (1 + abs(ref[:, 1].sum()))
    assert abs(m["max_abs_qacc"] - np.abs(qacc).max()) <= 1e-5 * (1 + np.abs(qacc).max())
    assert abs(m["max_time"] - K * model["timestep"][0]) < 1e-12
    assert abs(m["max_abs_qvel"] - np.abs(b.get("qvel")).max()) == 0
    b.close()
