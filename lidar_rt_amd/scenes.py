"""Deterministic synthetic measurement inputs.

Pure numpy, no torch: the same generator feeds the oracle, the parity tests,
``bench.py`` and the golden-fixture script, on the CPU container and on the
GPU box (the 1 M scene is regenerated from the seed, never committed).

* ``kitti_rays``   – numpy restatement of ``LiDARSensor.get_range_rays`` in
  KITTI mode (reference ``lib/scene/lidar_sensor.py:395-434``; bounds from
  ``lib/dataloader/kitti_loader/__init__.py:187``).  Pinned against the
  imported reference in ``tests/golden/rays_*.npz``.
* ``make_scene``   – S10k / S1M Gaussian populations (ground / walls /
  clutter) with the reference's own SH-DC initialisation
  (``lib/dataloader/gs_loader.py:121-123``).
* ``upstream_grad``– dL/dout with channels 0-3 live, 4-8 zero.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

SEED = 20250725
SH_C0 = 0.28209479177387814
GROUND_Z = -1.73


def kitti_rays(H: int, W: int,
               inc_bounds_deg: Tuple[float, float] = (-24.9, 2.0),
               origin=(0.0, 0.0, 0.0)) -> Tuple[np.ndarray, np.ndarray]:
    """(ray_o, ray_d), each (H, W, 3) float32, row-major (beam, azimuth).

    Column w has azimuth (W-w)/W*2pi - pi (decreasing with w), row h has
    inclination (H-h)/H*(hi-lo)+lo (top row = max inclination); arithmetic is
    done in float32 like the reference's torch code.
    """
    lo = np.float32(math.radians(inc_bounds_deg[0]))
    hi = np.float32(math.radians(inc_bounds_deg[1]))
    x = (np.arange(W, 0, -1, dtype=np.float32)) / np.float32(W)
    az = x * np.float32(2.0) * np.float32(np.pi) - np.float32(np.pi)
    y = (np.arange(H, 0, -1, dtype=np.float32)) / np.float32(H)
    inc = y * (hi - lo) + lo
    inc = inc[:, None].astype(np.float32)
    az = az[None, :].astype(np.float32)
    d = np.stack([np.cos(inc) * np.cos(az),
                  np.cos(inc) * np.sin(az),
                  np.sin(inc) * np.ones_like(az)], axis=-1).astype(np.float32)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    o = np.broadcast_to(np.asarray(origin, np.float32), (H, W, 3)).copy()
    return o, np.ascontiguousarray(d, dtype=np.float32)


def _frame_quaternions(normal: np.ndarray, angle: np.ndarray) -> np.ndarray:
    """Quaternion (w,x,y,z) of a frame whose 3rd column is ``normal`` and whose
    in-plane axes are rotated by ``angle`` about it."""
    n = normal / np.linalg.norm(normal, axis=1, keepdims=True)
    helper = np.where(np.abs(n[:, 2:3]) < 0.9,
                      np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    t0 = np.cross(helper, n)
    t0 /= np.linalg.norm(t0, axis=1, keepdims=True)
    b0 = np.cross(n, t0)
    ca, sa = np.cos(angle)[:, None], np.sin(angle)[:, None]
    t = ca * t0 + sa * b0
    b = np.cross(n, t)
    R = np.stack([t, b, n], axis=2)  # columns = (t, b, n), right-handed
    # rotation matrix -> quaternion (branch on the largest diagonal term)
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    tr = m00 + m11 + m22
    q = np.empty((R.shape[0], 4))
    c0 = tr > 0
    c1 = (~c0) & (m00 >= m11) & (m00 >= m22)
    c2 = (~c0) & (~c1) & (m11 >= m22)
    c3 = (~c0) & (~c1) & (~c2)
    with np.errstate(invalid="ignore"):
        s = np.sqrt(np.maximum(tr + 1.0, 1e-30)) * 2
        q0 = np.stack([0.25 * s, (R[:, 2, 1] - R[:, 1, 2]) / s,
                       (R[:, 0, 2] - R[:, 2, 0]) / s, (R[:, 1, 0] - R[:, 0, 1]) / s], 1)
        s = np.sqrt(np.maximum(1.0 + m00 - m11 - m22, 1e-30)) * 2
        q1 = np.stack([(R[:, 2, 1] - R[:, 1, 2]) / s, 0.25 * s,
                       (R[:, 0, 1] + R[:, 1, 0]) / s, (R[:, 0, 2] + R[:, 2, 0]) / s], 1)
        s = np.sqrt(np.maximum(1.0 + m11 - m00 - m22, 1e-30)) * 2
        q2 = np.stack([(R[:, 0, 2] - R[:, 2, 0]) / s, (R[:, 0, 1] + R[:, 1, 0]) / s,
                       0.25 * s, (R[:, 1, 2] + R[:, 2, 1]) / s], 1)
        s = np.sqrt(np.maximum(1.0 + m22 - m00 - m11, 1e-30)) * 2
        q3 = np.stack([(R[:, 1, 0] - R[:, 0, 1]) / s, (R[:, 0, 2] + R[:, 2, 0]) / s,
                       (R[:, 1, 2] + R[:, 2, 1]) / s, 0.25 * s], 1)
    q[c0], q[c1], q[c2], q[c3] = q0[c0], q1[c1], q2[c2], q3[c3]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def make_scene(P: int, seed: int = SEED, radius_scale: float = 1.0,
               sh_coeffs: int = 16) -> Dict[str, np.ndarray]:
    """Synthetic static scene around a sensor at the origin.

    Returns float32 arrays: means (P,3), scales (P,2) [post-exp], rotations
    (P,4) (w,x,y,z) unit, opacities (P,1) [post-sigmoid], shs (P,16,3).
    Every Gaussian centre is >= 2*radius_scale metres from the origin, so the
    reference's t<0.2 stale-entry quirk is never exercised.
    """
    rng = np.random.default_rng(seed)
    rs = float(radius_scale)
    n_ground = P // 2
    n_wall = (P * 4) // 10
    n_clut = P - n_ground - n_wall

    # --- ground: annulus 2..60 m (uniform in area), slightly wavy normals
    r = np.sqrt(rng.uniform((2 * rs) ** 2, (60 * rs) ** 2, n_ground))
    a = rng.uniform(0, 2 * np.pi, n_ground)
    g_xyz = np.stack([r * np.cos(a), r * np.sin(a),
                      GROUND_Z + rng.normal(0, 0.02, n_ground)], 1)
    g_n = np.stack([rng.normal(0, .05, n_ground), rng.normal(0, .05, n_ground),
                    np.ones(n_ground)], 1)

    # --- walls: 32 vertical rectangles 20 m x 6 m, bottom on the ground
    n_rect = 32
    wc_r = rng.uniform(10 * rs, 50 * rs, n_rect)
    wc_a = rng.uniform(0, 2 * np.pi, n_rect)
    yaw = rng.uniform(0, 2 * np.pi, n_rect)
    which = rng.integers(0, n_rect, n_wall)
    along = rng.uniform(-10 * rs, 10 * rs, n_wall)
    up = rng.uniform(0, 6 * rs, n_wall)
    wdir = np.stack([np.cos(yaw), np.sin(yaw), np.zeros(n_rect)], 1)
    wnrm = np.stack([-np.sin(yaw), np.cos(yaw), np.zeros(n_rect)], 1)
    wcen = np.stack([wc_r * np.cos(wc_a), wc_r * np.sin(wc_a),
                     np.full(n_rect, GROUND_Z)], 1)
    w_xyz = wcen[which] + along[:, None] * wdir[which]
    w_xyz[:, 2] += up
    w_n = wnrm[which] + rng.normal(0, .05, (n_wall, 3))
    # keep walls out of the 2 m exclusion zone
    wr = np.linalg.norm(w_xyz[:, :2], axis=1)
    near = wr < 2 * rs
    w_xyz[near, :2] *= (2 * rs / np.maximum(wr[near], 1e-6))[:, None]

    # --- clutter: uniform box, excluding r < 2 m
    c_xyz = np.stack([rng.uniform(-50 * rs, 50 * rs, n_clut),
                      rng.uniform(-50 * rs, 50 * rs, n_clut),
                      rng.uniform(-1.7, 4.0, n_clut)], 1)
    cr = np.linalg.norm(c_xyz[:, :2], axis=1)
    near = cr < 2 * rs
    c_xyz[near, :2] *= (2 * rs / np.maximum(cr[near], 1e-6))[:, None]
    c_n = rng.normal(0, 1, (n_clut, 3))

    means = np.concatenate([g_xyz, w_xyz, c_xyz], 0)
    normals = np.concatenate([g_n, w_n, c_n], 0)
    perm = rng.permutation(P)  # no spatial order in the input arrays
    means, normals = means[perm], normals[perm]

    scales = np.exp(rng.uniform(np.log(0.03), np.log(0.25), (P, 2)))
    rot = _frame_quaternions(normals, rng.uniform(0, 2 * np.pi, P))
    opac = np.clip(1.0 / (1.0 + np.exp(-rng.normal(0, 2, (P, 1)))), 0.01, 0.99)
    # SH table generated directly in float32 (192 MB at 1 M; avoids a 384 MB f64 temp)
    shs = rng.standard_normal((P, sh_coeffs, 3), dtype=np.float32)
    shs *= np.float32(0.02)
    rgb = np.stack([rng.uniform(0, 1, P), np.ones(P), np.zeros(P)], 1)
    shs[:, 0, :] = ((rgb - 0.5) / SH_C0).astype(np.float32)  # RGB2SH
    return {
        "means": means.astype(np.float32),
        "scales": scales.astype(np.float32),
        "rotations": rot.astype(np.float32),
        "opacities": opac.astype(np.float32),
        "shs": shs,
    }


def upstream_grad(H: int, W: int, seed: int = SEED + 1) -> np.ndarray:
    """dL/dout (H,W,9) float32: N(0,1)/(H*W) on channels 0-3, zero on 4-8."""
    rng = np.random.default_rng(seed)
    g = np.zeros((H, W, 9), np.float32)
    g[..., 0:4] = (rng.normal(0, 1, (H, W, 4)) / (H * W)).astype(np.float32)
    return g


# Named workloads (BASELINE.json configs[0], configs[1])
def s10k():
    sc = make_scene(10_000, radius_scale=1.0 / 3.0)
    o, d = kitti_rays(16, 256)
    return sc, o, d


def s1m():
    sc = make_scene(1_000_000)
    o, d = kitti_rays(64, 2048)
    return sc, o, d


BG_DEFAULT = np.array([0.0, 0.0, 1.0], np.float32)  # train.py:104-106


def dense_translucent(P: int = 25000, seed: int = 21, H: int = 4, W: int = 48):
    """Stress scene: a 5 m world of large Gaussians (3x the usual scales) of opacity 0.03 -- about 130 candidate and 100
    composited hits per ray, up to ~600 / ~500 -- with everything that could come within 0.3 m of the sensor removed (hits
    nearer than 0.2 m are outside the parity contract).  Returns (scene, ray_o, ray_d)."""
    sc = make_scene(P, seed=seed, radius_scale=0.1)
    sc["opacities"][:] = 0.03
    sc["scales"] *= 3.0
    o, d = kitti_rays(H, W)
    reach = 0.3 + 2.1 * 1.4143 * sc["scales"].max(1)              # half diagonal of the quad at opacity 0.03 (2.02 sigma)
    keep = np.linalg.norm(sc["means"] - o.reshape(-1, 3)[0], axis=1) > reach
    return {k: np.ascontiguousarray(v[keep]) for k, v in sc.items()}, o, d
