"""Pins of the CPU oracle and the scene/ray generators against the golden
vectors produced by the reference's own Python helpers
(oracle/gen_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

from lidar_rt_amd import scenes
from oracle import oracle


@pytest.fixture(scope="module")
def conv(golden_dir):
    return np.load(os.path.join(golden_dir, "conventions.npz"))


def test_quaternion_convention_matches_build_rotation(conv):
    # lib/utils/general_utils.py:176-197  (w,x,y,z), row-major R, normalises internally
    R = oracle.quat_to_R(conv["quats"], "f64")
    np.testing.assert_allclose(R, conv["build_rotation"], rtol=0, atol=2e-6)
    R32 = oracle.quat_to_R(conv["quats"], "f32")
    np.testing.assert_allclose(R32, conv["build_rotation"], rtol=0, atol=2e-6)


def test_quads_match_build2DRectangle(conv):
    # lib/utils/primitive_utils.py:182-224: corner order, factor, faces
    orc = oracle.Oracle(conv["means"], conv["scales"], conv["quats_unit"], conv["opacities"], "f32")
    np.testing.assert_allclose(orc.vertices, conv["rect_vertices"], rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(orc.faces, conv["rect_faces"])


def test_sh_basis_matches_eval_sh(conv):
    # lib/utils/sh_utils.py:58-113 ; kernel layout is (P,16,3) = transpose of eval_sh's (...,3,16)
    dirs, sh = conv["sh_dirs"], conv["sh_coeffs_c16"]
    for deg in range(4):
        b = oracle.sh_basis(deg, dirs, "f64")
        col = np.einsum("nk,nck->nc", b, sh)
        np.testing.assert_allclose(col, conv[f"eval_sh_deg{deg}"], rtol=1e-12, atol=1e-12)
        nb = (deg + 1) ** 2
        assert np.all(b[:, nb:] == 0)


def test_rgb2sh(conv):
    np.testing.assert_allclose((conv["rgb"] - 0.5) / scenes.SH_C0, conv["rgb2sh"], rtol=1e-12)


def test_kitti_rays_match_get_range_rays(golden_dir):
    g = np.load(os.path.join(golden_dir, "rays_kitti_16x256.npz"))
    o, d = scenes.kitti_rays(16, 256)
    np.testing.assert_array_equal(o, g["ray_o"])
    np.testing.assert_allclose(d, g["ray_d"], rtol=0, atol=3e-7)
    # orientation pins: azimuth decreases with column, top row = max inclination
    az = np.arctan2(d[0, :, 1], d[0, :, 0])
    assert az[1] > az[2] > az[3]          # (col 0 sits on the +-pi seam)
    assert d[0, 0, 2] > d[-1, 0, 2]
    s = np.load(os.path.join(golden_dir, "rays_kitti_64x2048_sample.npz"))
    o2, d2 = scenes.kitti_rays(64, 2048)
    np.testing.assert_allclose(d2[::7, ::61], s["ray_d"], rtol=0, atol=3e-7)


def test_scene_generator_is_deterministic(golden_dir):
    g = np.load(os.path.join(golden_dir, "s10k_golden.npz"))
    sc, o, d = scenes.s10k()
    h = hashlib.sha256(b"".join(sc[k].tobytes() for k in sorted(sc))).hexdigest()
    assert h == str(g["scene_sha256"])
    assert np.linalg.norm(sc["means"], axis=1).min() >= 0.5      # keep >= 0.5 m from the sensor
    assert sc["opacities"].min() > 1 / 255


def test_oracle_reproduces_s10k_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "s10k_golden.npz"))
    sc, o, d = scenes.s10k()
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], "f32")
    fw = orc.forward(o, d, sc["shs"], 3, scenes.BG_DEFAULT, stats=True)
    np.testing.assert_allclose(fw["out"], g["out"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(fw["n_comp"], g["n_comp"])
    bw = orc.backward(o, d, sc["shs"], 3, scenes.BG_DEFAULT, fw["out"], scenes.upstream_grad(16, 256))
    for k in ("means", "shs", "opacities", "scales", "rotations"):
        ref = g["d_" + k]
        np.testing.assert_allclose(bw[k], ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max())
