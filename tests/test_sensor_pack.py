"""The sensors-plugin equivalent: device-side `sensordata -> published float32 values` with
the registered noise models.  Reference semantics: mujoco_sensor_handler_plugin.cpp:175-437 (lastStageCallback) and
:123-173 (registerNoiseModelsCB); the reference's own tests pin them statistically (mujoco_sensors_test.cpp:281-711:
value == sensordata / cutoff without noise, sample mean / std of (value - truth) match the registered model).
CPU tests pin the oracle restatement that way; the GPU test compares the HIP kernel with it exactly."""
import os

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _model():
    return mjcf.compile_xml_file(os.path.join(GOLDEN, "sensors_world.xml"))


def _noise_setup(m):
    """A noise model on one sensor of every message kind present in the reference's sensors_world.xml."""
    ns = m["nsensor"]
    flag, mean, sigma = np.zeros(ns, np.int32), np.zeros((ns, 3)), np.zeros((ns, 3))
    picked = {}
    for i in range(ns):
        dim, typ = int(m["sensor_dim"][i]), int(m["sensor_type"][i])
        kind = "quat" if typ in (15, 24) else {1: "scalar", 3: "vector"}.get(dim)
        if kind and kind not in picked:
            picked[kind] = i
    if "scalar" in picked:
        i = picked["scalar"]
        flag[i], mean[i, 0], sigma[i, 0] = 1, 0.5, 0.25
    if "vector" in picked:
        i = picked["vector"]
        flag[i] = 0b101                       # x and z noisy; the 2nd SET bit reads index 1
        mean[i, :2], sigma[i, :2] = [1.0, -2.0], [0.025, 0.05]
    if "quat" in picked:
        i = picked["quat"]
        flag[i], mean[i], sigma[i] = 0b111, [0.0, 0.0, 0.0], [0.01, 0.02, 0.03]
    return flag, mean, sigma, picked


def test_pack_without_noise_is_value_over_cutoff(oracle_built):
    m = _model()
    rng = np.random.default_rng(0)
    sd = rng.normal(size=m["nsensordata"])
    ns = m["nsensor"]
    val, tru = oracle_built.sensor_pack(m, sd, np.zeros(ns, np.int32), np.zeros(3 * ns), np.zeros(3 * ns), 1, 0, 0)
    for i in range(ns):
        a, d = m["sensor_adr"][i], m["sensor_dim"][i]
        cut = m["sensor_cutoff"][i] if m["sensor_cutoff"][i] > 0 else 1.0
        np.testing.assert_array_equal(val[a:a + d], (sd[a:a + d] / cut).astype(np.float32))
    np.testing.assert_array_equal(val, tru)


def test_noise_statistics_match_the_registered_model(oracle_built):
    m = _model()
    flag, mean, sigma, picked = _noise_setup(m)
    assert set(picked) >= {"scalar", "vector"}
    sd = np.random.default_rng(1).normal(size=m["nsensordata"])
    N = 4000
    vals = np.stack([oracle_built.sensor_pack(m, sd, flag, mean.ravel(), sigma.ravel(), 7, e, 3)[0] for e in range(N)])
    i = picked["scalar"]
    a, cut = m["sensor_adr"][i], (m["sensor_cutoff"][i] if m["sensor_cutoff"][i] > 0 else 1.0)
    d = (vals[:, a] - sd[a]) * cut                      # the reference adds noise / cutoff to the RAW reading
    assert abs(d.mean() - 0.5) < 4 * 0.25 / np.sqrt(N) and abs(d.std() - 0.25) < 0.02
    i = picked["vector"]
    a, cut = m["sensor_adr"][i], (m["sensor_cutoff"][i] if m["sensor_cutoff"][i] > 0 else 1.0)
    dx, dy, dz = [(vals[:, a + k] - sd[a + k]) * cut for k in range(3)]
    assert abs(dx.mean() - 1.0) < 4 * 0.025 / np.sqrt(N) + 1e-5 and abs(dx.std() - 0.025) < 0.004
    assert np.all(np.abs(dy) < 1e-6)                    # bit 1 not set: untouched (float32 rounding only)
    assert abs(dz.mean() + 2.0) < 4 * 0.05 / np.sqrt(N) + 1e-5 and abs(dz.std() - 0.05) < 0.006
    if "quat" in picked:
        a = m["sensor_adr"][picked["quat"]]
        q = vals[:, a:a + 4]
        np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1, atol=1e-6)
        q0 = sd[a:a + 4] / np.linalg.norm(sd[a:a + 4])
        ang = 2 * np.arccos(np.clip(np.abs(q @ q0), 0, 1))
        assert 0.01 < ang.mean() < 0.08                 # ~ |N(0, (0.01, 0.02, 0.03))|
    # different envs / steps draw different numbers; the same key reproduces
    v1 = oracle_built.sensor_pack(m, sd, flag, mean.ravel(), sigma.ravel(), 7, 5, 3)[0]
    v2 = oracle_built.sensor_pack(m, sd, flag, mean.ravel(), sigma.ravel(), 7, 5, 3)[0]
    v3 = oracle_built.sensor_pack(m, sd, flag, mean.ravel(), sigma.ravel(), 7, 5, 4)[0]
    assert np.array_equal(v1, v2) and not np.array_equal(v1, v3)


@pytest.mark.gpu
def test_gpu_pack_matches_oracle(oracle_built):
    from mujoco_ros_pkgs_amd import engine
    m = _model()
    flag, mean, sigma, picked = _noise_setup(m)
    cm = engine.CompiledModel(m)
    nenv = 300
    b = engine.Batch(cm, nenv)
    b.set_ctrl_noise(0.0, 0.1, 0, 1000)                 # env_offset 1000: global env ids key the stream
    b.step(5)
    b.sensor_pack(seed=42)
    sd = b.get("sensordata")
    v0, t0 = b.sensor_messages("value"), b.sensor_messages("truth")
    np.testing.assert_array_equal(v0, t0)               # no model registered yet
    for i in range(m["nsensor"]):
        if flag[i]:
            b.sensor_set_noise(i, int(flag[i]), mean[i][:bin(flag[i]).count("1")], sigma[i][:bin(flag[i]).count("1")])
    b.sensor_pack(seed=42)
    v, t = b.sensor_messages("value"), b.sensor_messages("truth")
    assert not np.array_equal(v, t)
    for e in (0, 1, 17, nenv - 1):
        ov, ot = oracle_built.sensor_pack(m, sd[e], flag, mean.ravel(), sigma.ravel(), 42, 1000 + e, 5)
        np.testing.assert_array_equal(t[e], ot)
        np.testing.assert_allclose(v[e], ov, rtol=0, atol=2e-7)   # float32 of fp64 values that agree to ~1e-16
    b.close()
