"""Residency guard: how many envs one CU's 160 KB of LDS holds is decided by the fused step's frame, and the round-2
throughput of the contact configs rests on it -- eight lean frames per CU for config 3 (two waves per SIMD with the 256-register PGS
kernel), four for config 5 -- one per SIMD, the 512-register Newton kernel's ceiling (round 3: 64 rows of EVERY per-row array in the
frame, the rest of its 200-row capacity in the env's HBM block).  Host-side only: mjb_compile needs no GPU."""
import os

from mujoco_ros_pkgs_amd import engine, mjcf

LDS = 160 * 1024


def _bytes(model):
    cm = engine.CompiledModel(model)
    return cm.lib.mjb_frame_bytes(cm.ptr, 0), cm.lib.mjb_frame_bytes(cm.ptr, 1)


def test_config3_lean_frame_fits_eight_per_cu():
    m = mjcf.load_asset("franka_table")
    assert (m["nconmax"], m["nefcmax"]) == (16, 73)   # SURVEY.md §8 table size
    full, fused = _bytes(m)
    assert fused * 8 <= LDS < fused * 9, (full, fused)
    assert fused < full // 2          # efc_J overlays dead fields, row bookkeeping cut to what the solver reads


def test_config5_lean_frame_fits_four_per_cu():
    m = mjcf.load_asset("shadow_hand_like")
    assert (m["nconmax"], m["nefcmax"]) == (48, 200)
    full, fused = _bytes(m)
    granules = (fused + 1279) // 1280                # gfx950 hands out LDS in 1280-byte granules, 128 per CU
    assert full <= LDS and 4 * granules <= 128, (full, fused)
    assert full - fused >= 8 * (200 - 64) * (m["nv"] + 7)   # the rows beyond the frame's share: efc_J and the seven per-row doubles


def test_unconstrained_compact_frame_unchanged():
    full, fused = _bytes(mjcf.load_asset("franka_like"))
    assert fused * 16 <= LDS and fused < full          # config 2: all 4096 envs resident at 16 per CU


def test_debug_knob_lowers_the_frames_share_of_J():
    m = mjcf.load_asset("shadow_hand_like")
    _, fused = _bytes(m)
    os.environ["MJB_DEBUG_JROWS"] = "8"
    try:
        _, fused8 = _bytes(m)
    finally:
        os.environ.pop("MJB_DEBUG_JROWS", None)
    assert fused8 <= fused      # (the overlay region cannot shrink below the fields efc_J shares it with)
