"""Actuators on fixed tendons (mjTRN_TENDON; SURVEY.md §8a rows A8 / A12, mj_transmission): length = gear * ten_length, velocity =
moment . qvel, qfrc_actuator += moment' force with moment = gear * the tendon's coefficients -- how a Shadow-Hand-like model couples two finger
joints to one motor ("tendon couplings").  Oracle against the definition in numpy, kernels against the oracle."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

XML = """
<mujoco model="tendon_drive">
  <compiler angle="radian"/>
  <option timestep="0.002" integrator="{integrator}" solver="{solver}" cone="{cone}" iterations="60" tolerance="1e-10"/>
  <size nconmax="{ncon}" njmax="{njmax}"/>
  <default><joint damping="0.05" armature="0.005"/></default>
  <worldbody>
    {floor}
    <body name="prox" pos="0 0 0.5">
      <joint name="j1" type="hinge" axis="0 1 0" limited="true" range="-0.3 1.6"/>
      <geom type="capsule" fromto="0 0 0 0.12 0 0" size="0.015" mass="0.1"/>
      <body name="mid" pos="0.12 0 0">
        <joint name="j2" type="hinge" axis="0 1 0" limited="true" range="0 1.6"/>
        <geom type="capsule" fromto="0 0 0 0.08 0 0" size="0.013" mass="0.06"/>
        <body name="dist" pos="0.08 0 0">
          <joint name="j3" type="hinge" axis="0 1 0" limited="true" range="0 1.6"/>
          <geom type="capsule" fromto="0 0 0 0.06 0 0" size="0.012" mass="0.04"/>
          <body name="nail" pos="0.06 0 0">
            <joint name="j4" type="slide" axis="1 0 0" limited="true" range="-0.01 0.01"/>
            <geom type="sphere" size="0.012" mass="0.01"/>
          </body>
        </body>
      </body>
    </body>
    {puck}
  </worldbody>
  <tendon>
    <fixed name="flex"><joint joint="j2" coef="1"/><joint joint="j3" coef="1"/></fixed>
    <fixed name="mix"><joint joint="j4" coef="20"/><joint joint="j1" coef="-0.5"/></fixed>
  </tendon>
  <actuator>
    <position name="a_flex" tendon="flex" kp="3" kv="0.1" ctrllimited="true" ctrlrange="0 3.2"/>
    <motor name="a_j1" joint="j1" gear="0.5"/>
    <motor name="a_mix" tendon="mix" gear="0.3" forcelimited="true" forcerange="-0.4 0.4"/>
    <general name="a_flex2" tendon="flex" dyntype="filter" dynprm="0.03" gainprm="0.5" gear="-1.5"/>
  </actuator>
  <sensor>
    <actuatorpos actuator="a_flex"/>
    <actuatorvel actuator="a_mix"/>
    <actuatorfrc actuator="a_flex"/>
    <jointactuatorfrc joint="j3"/>
    <tendonpos tendon="flex"/>
  </sensor>
</mujoco>
"""


def model_of(integrator="Euler", solver="Newton", cone="pyramidal", contacts=False):
    floor = '<geom name="floor" type="plane" size="3 3 0.1"/>' if contacts else ""
    puck = ('<body name="puck" pos="0.2 0 0.4"><freejoint/><geom type="sphere" size="0.04" mass="0.05"/></body>') if contacts else ""
    return mjcf.compile_xml_string(XML.format(integrator=integrator, solver=solver, cone=cone, ncon=8 if contacts else 0,
                                              njmax=40 if contacts else 12, floor=floor, puck=puck))


def test_loader_and_oracle_follow_the_definition(oracle_built):
    m = model_of()
    assert list(m["actuator_trntype"]) == [3, 0, 3, 3] and list(m["actuator_trnid"][:, 0]) == [0, 0, 1, 0]
    with pytest.raises(mjcf.MjcfError):
        mjcf.compile_xml_string(XML.format(integrator="Euler", solver="Newton", cone="pyramidal", ncon=0, njmax=8, floor="", puck="")
                                .replace('tendon="mix" gear', 'tendon="nope" gear'))
    with pytest.raises(mjcf.MjcfError):   # implicitfast: the servo's velocity term on a tendon would couple j2 and j3 in the implicit matrix
        model_of(integrator="implicitfast")
    d = oracle_built.OracleData(m)
    rng = np.random.default_rng(2)
    coef = np.zeros((2, 4))
    coef[0, [1, 2]] = 1.0
    coef[1, 3], coef[1, 0] = 20.0, -0.5
    gear = np.array([1.0, 0.5, 0.3, -1.5])
    for trial in range(8):
        d.reset()
        q, v = rng.uniform(0, 0.8, 4) * np.array([1, 1, 1, 0.01]), rng.uniform(-1, 1, 4)
        ctrl, act = rng.uniform(-0.5, 3.5, 4), rng.uniform(-1, 1, 1)
        d.qpos[:], d.qvel[:], d.ctrl[:], d.act[:] = q, v, ctrl, act
        d.forward()
        tl, tv = coef @ q, coef @ v
        length = np.array([gear[0] * tl[0], gear[1] * q[0], gear[2] * tl[1], gear[3] * tl[0]])
        vel = np.array([gear[0] * tv[0], gear[1] * v[0], gear[2] * tv[1], gear[3] * tv[0]])
        np.testing.assert_allclose(d.actuator_length, length, atol=1e-14)
        np.testing.assert_allclose(d.actuator_velocity, vel, atol=1e-14)
        force = np.array([3 * np.clip(ctrl[0], 0, 3.2) - 3 * length[0] - 0.1 * vel[0], ctrl[1], np.clip(ctrl[2], -0.4, 0.4), 0.5 * act[0]])
        np.testing.assert_allclose(d.actuator_force, force, atol=1e-13)
        moment = np.array([gear[0] * coef[0], [gear[1], 0, 0, 0], gear[2] * coef[1], gear[3] * coef[0]])
        np.testing.assert_allclose(d.qfrc_actuator, moment.T @ force, atol=1e-13)
        sd = np.array(d.sensordata)
        np.testing.assert_allclose(sd, [length[0], vel[2], force[0], (moment.T @ force)[2], tl[0]], atol=1e-13)


CASES = [("Euler", "Newton", "pyramidal", False), ("RK4", "Newton", "pyramidal", False), ("Euler", "PGS", "pyramidal", True),
         ("Euler", "Newton", "elliptic", True), ("Euler", "CG", "pyramidal", True)]


@pytest.mark.gpu
@pytest.mark.parametrize("integrator,solver,cone,contacts", CASES)
def test_gpu_tendon_actuators_match_oracle(oracle_built, integrator, solver, cone, contacts):
    from mujoco_ros_pkgs_amd import engine
    m = model_of(integrator, solver, cone, contacts)
    n = 40
    rng = np.random.default_rng(4)
    qpos = np.tile(np.asarray(m["qpos0"], float), (n, 1))
    qpos[:, :4] = rng.uniform(0, 0.9, (n, 4)) * np.array([1, 1, 1, 0.01])
    qvel = rng.uniform(-1, 1, (n, m["nv"]))
    ctrl = rng.uniform(-0.5, 3.5, (n, m["nu"]))
    cm = engine.CompiledModel(m)
    d = oracle_built.OracleData(m)
    for nstep in (1, 30):
        b = engine.Batch(cm, n)
        b.set("qpos", qpos); b.set("qvel", qvel); b.set("ctrl", ctrl)
        b.step(nstep)
        got = {k: b.get(k) for k in ("qpos", "qvel", "act", "sensordata")}
        assert not b.lane_env_info()[1]
        tol = 1e-11 if nstep == 1 else (1e-6 if solver == "CG" else 1e-8)
        for e in range(n):
            d.reset(); d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]; d.ctrl[:] = ctrl[e]
            d.step(nstep)
            for k in got:
                r = np.array(getattr(d, k))
                assert np.abs(got[k][e] - r).max() <= tol * (1 + np.abs(r).max()) * (100 if k == "qvel" else 1), (nstep, e, k, got[k][e], r)
        b.close()
    # the derived fields of one forward pass
    b = engine.Batch(cm, n)
    b.set("qpos", qpos); b.set("qvel", qvel); b.set("ctrl", ctrl)
    b.forward()
    for e in range(0, n, 7):
        d.reset(); d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]; d.ctrl[:] = ctrl[e]
        d.forward()
        for k in ("actuator_length", "actuator_velocity", "actuator_force", "qfrc_actuator"):
            np.testing.assert_allclose(b.get(k)[e], np.array(getattr(d, k)), rtol=0, atol=1e-12, err_msg=k)
    b.close()


def test_loader_refuses_options_it_would_otherwise_drop():
    one = '<mujoco><option %s/><worldbody><body><joint type="hinge"/><geom size="0.1"/></body></worldbody></mujoco>'
    for opt in ('noslip_iterations="3"', 'density="1.2"', 'viscosity="0.001"'):
        with pytest.raises(mjcf.MjcfError):
            mjcf.compile_xml_string(one % opt)
    assert mjcf.compile_xml_string(one % 'noslip_iterations="0" density="0" wind="1 0 0"')["nv"] == 1
