"""CPU: the saved env-step behind tests/test_gpu_mixed_condim.py is what its README line says -- on the oracle: 12 contacts whose elliptic cones have
dimensions 4 x 10, 3, 4, two limit rows, 49 rows, a finite Newton solution.  (The GPU tests compare the HIP path with this; here the fixture itself is pinned.)"""
import os

import numpy as np

STATE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grasp_env184_step4059.npz")


def test_fixture_holds_cones_of_mixed_dimension(oracle_built):
    from mujoco_ros_pkgs_amd import mjcf
    po = oracle_built
    model = mjcf.load_asset("shadow_hand_grasp")
    st = np.load(STATE)
    assert st["qpos"].shape == (model["nq"],) and st["qvel"].shape == (model["nv"],) and st["ctrl_step"].shape == (model["nu"],)
    d = po.OracleData(model)
    d.reset()
    d.qpos[:] = st["qpos"]
    d.qvel[:] = st["qvel"]
    d.qacc_warmstart[:] = st["qacc_warmstart"]
    d.ctrl[:] = st["ctrl_step"]
    d.forward()
    nc, ne = int(d.ncon[0]), int(d.nefc[0])
    dims = np.array(d.contact_dim[:nc]).astype(int).tolist()
    assert (nc, ne) == (12, 49) and dims == [4] * 10 + [3, 4], (nc, ne, dims)
    types = np.array(d.efc_type[:ne]).astype(int)
    assert (types[:2] == types[0]).all() and (types[2:] == types[2]).all() and types[0] != types[2]  # two limit rows, then the cones' rows
    assert np.isfinite(np.array(d.qacc)).all() and 100 < np.abs(np.array(d.qacc)).max() < 1e4
    assert 1 <= int(d.solver_iter[0]) <= 10
    d.step()
    assert d.warning(6) == 0  # mj_checkAcc does not fire on the CPU restatement
