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


# ------------------------------------------------------------------------------------------------------------------
# Dataset-shaped synthetic frames for BASELINE configs[2..4]: the range-image geometry of the two
# datasets the reference trains on -- the data themselves are not available here.
#   Waymo top LiDAR : 64 beams x 2650 columns, per-beam inclination table, 0.5-pixel offsets, sensor-to-ego yaw
#                     (lib/dataloader/waymo_loader/__init__.py:36-131, lib/scene/lidar_sensor.py:395-434)
#   KITTI-360 HDL-64: 66 x 1030 range image, inclination interpolated in (-24.9, 2.0) degrees
#                     (lib/dataloader/kitti_loader/__init__.py:186-187)
WAYMO_HW = (64, 2650)
KITTI360_HW = (66, 1030)


def waymo_inclinations(H: int = 64) -> np.ndarray:
    """A per-beam inclination table shaped like the Waymo top LiDAR's: -17.6 .. +2.4 degrees, beams packed more densely
    towards the horizon (ascending, radians, float32; the sensor model flips it so that row 0 is the highest beam)."""
    u = np.linspace(0.0, 1.0, H)
    deg = -17.6 + 20.0 * (0.35 * u + 0.65 * np.sqrt(u))
    return np.radians(deg).astype(np.float32)


def pose_matrix(t, yaw=0.0, pitch=0.0, roll=0.0) -> np.ndarray:
    """4x4 float32 rigid transform from a translation and z-y-x Euler angles (radians)."""
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                  [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                  [-sp, cp * sr, cp * cr]])
    m = np.eye(4)
    m[:3, :3] = R; m[:3, 3] = np.asarray(t, np.float64)
    return m.astype(np.float32)


def range_rays(H: int, W: int, inclination, sensor2world: np.ndarray, data_type: str = "KITTI",
               sensor2ego: np.ndarray = None) -> Tuple[np.ndarray, np.ndarray]:
    """numpy float32 restatement of ``LiDARSensor.get_range_rays`` (lib/scene/lidar_sensor.py:395-434) for both dataset
    modes; the torch version the training loop uses is ``training.RangeFrames.range_rays`` (pinned against the
    reference's golden grids) and ``tests/test_config_shapes_gpu.py`` checks the two against each other."""
    f = np.float32
    waymo = data_type == "Waymo"
    off = f(0.5) if waymo else f(0.0)
    yaw = f(math.atan2(float(sensor2ego[1, 0]), float(sensor2ego[0, 0]))) if waymo else f(0.0)
    x = (np.arange(W, 0, -1, dtype=f) - off) / f(W)
    az = (x * f(2.0) * f(np.pi) - f(np.pi) - yaw)[None, :].astype(f)
    inc = np.asarray(inclination, f).reshape(-1)
    if inc.size == 2:
        y = (np.arange(H, 0, -1, dtype=f) - off) / f(H)
        incl = (y * (inc[1] - inc[0]) + inc[0])[:, None].astype(f)
    else:
        incl = inc[::-1][:, None].astype(f)
    d = np.stack([np.cos(incl) * np.cos(az), np.cos(incl) * np.sin(az), np.sin(incl) * np.ones_like(az)], -1).astype(f)
    d = (d @ sensor2world[:3, :3].T.astype(f)).astype(f)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True).astype(f)
    o = np.broadcast_to(sensor2world[:3, 3].astype(f), (H, W, 3)).copy()
    return o, np.ascontiguousarray(d, dtype=f)


def waymo_frame(P: int = 2_000_000, seed: int = SEED + 7, radius_scale: float = 1.25, frame: int = 0):
    """BASELINE configs[2] shape: a static scene of ~2 M background Gaussians seen by a Waymo-style top LiDAR (64 x 2650,
    per-beam inclinations, posed sensor).  The sensor sits 0.45 m above the scene generator's origin and moves 0.8 m per frame;
    every Gaussian centre stays > 1.5 m away from it.  Returns (scene, ray_o, ray_d)."""
    sc = make_scene(P, seed=seed, radius_scale=radius_scale)
    o, d = waymo_sensor_rays(frame)
    return sc, o, d


def waymo_sensor_rays(frame: int = 0):
    """The 64 x 2650 ray grid of the synthetic Waymo-style sensor at `frame` (0.8 m forward and a little yaw per frame)."""
    H, W = WAYMO_HW
    s2w = pose_matrix((0.8 * frame, 0.1 * frame, 0.45), yaw=0.03 * frame + 0.4, pitch=0.01, roll=-0.008)
    s2e = pose_matrix((1.43, 0.0, 2.18), yaw=0.02)
    return range_rays(H, W, waymo_inclinations(H), s2w, "Waymo", s2e)


def actor_asset(n: int, rng, size=(4.4, 1.9, 1.6)) -> Dict[str, np.ndarray]:
    """Gaussians on the surface of a car-sized box in the ACTOR frame (what the reference keeps per rigid object,
    lib/scene/gaussian_model.py:129-134): means (n,3), scales, rotations (local), opacities, shs."""
    sx, sy, sz = size
    face = rng.integers(0, 5, n)                                   # 4 sides + roof
    u, v = rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)
    p = np.zeros((n, 3)); nrm = np.zeros((n, 3))
    for f_, (ax, sgn) in enumerate(((0, 1), (0, -1), (1, 1), (1, -1), (2, 1))):
        m = face == f_
        a1, a2 = [a for a in range(3) if a != ax]
        p[m, ax] = sgn * 0.5 * size[ax]; p[m, a1] = u[m] * size[a1]; p[m, a2] = v[m] * size[a2]
        nrm[m, ax] = sgn
    p[:, 2] += 0.5 * sz
    nrm += rng.normal(0, 0.05, (n, 3))
    scales = np.exp(rng.uniform(np.log(0.03), np.log(0.15), (n, 2)))
    rot = _frame_quaternions(nrm, rng.uniform(0, 2 * np.pi, n))
    opac = np.clip(1.0 / (1.0 + np.exp(-rng.normal(0.5, 1.5, (n, 1)))), 0.02, 0.98)
    shs = (rng.standard_normal((n, 16, 3)) * 0.02).astype(np.float32)
    rgb = np.stack([rng.uniform(0, 1, n), np.ones(n), np.zeros(n)], 1)
    shs[:, 0, :] = ((rgb - 0.5) / SH_C0).astype(np.float32)
    return {"means": p.astype(np.float32), "scales": scales.astype(np.float32), "rotations": rot.astype(np.float32),
            "opacities": opac.astype(np.float32), "shs": shs}


def kitti360_dynamic(P_bg: int = 500_000, n_actors: int = 8, per_actor: int = 8000, seed: int = SEED + 11):
    """BASELINE configs[3] shape: background Gaussians + ``n_actors`` rigid actor sets with a pose per frame, seen through
    a 66 x 1030 KITTI-360 range image.  Returns (background scene, [actor assets], poses(frame) -> [(t (3,), q (4,))],
    rays(frame) -> (ray_o, ray_d)).  Actors drive on circles of 6..30 m radius around the sensor, which itself advances
    0.5 m per frame; nothing comes within 2 m of it."""
    rng = np.random.default_rng(seed)
    bg = make_scene(P_bg, seed=seed, radius_scale=1.0)
    actors = [actor_asset(per_actor, rng) for _ in range(n_actors)]
    rad = rng.uniform(6.0, 30.0, n_actors); ph0 = rng.uniform(0, 2 * np.pi, n_actors); om = rng.uniform(-0.04, 0.04, n_actors)

    def poses(frame: int):
        out = []
        for a in range(n_actors):
            ph = ph0[a] + om[a] * frame
            t = np.array([rad[a] * np.cos(ph) + 0.5 * frame, rad[a] * np.sin(ph), GROUND_Z], np.float32)
            yaw = ph + np.pi / 2
            q = np.array([np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)], np.float32) * np.float32(1.0 + 0.3 * a)   # un-normalised
            out.append((t, q))
        return out

    def rays(frame: int):
        H, W = KITTI360_HW
        s2w = pose_matrix((0.5 * frame, 0.0, 0.0), yaw=0.01 * frame)
        return range_rays(H, W, (math.radians(-24.9), math.radians(2.0)), s2w, "KITTI")
    return bg, actors, poses, rays


def waymo_dynamic_4m(P_bg: int = 3_900_000, n_actors: int = 10, per_actor: int = 10_000, seed: int = SEED + 13, frame: int = 0):
    """BASELINE configs[4] shape: ~4 M Gaussians (background + actors already posed into the world) under the Waymo grid.
    Returns (scene, ray_o, ray_d); the scene spans 1.5x the S1M radius."""
    rng = np.random.default_rng(seed)
    sc = make_scene(P_bg, seed=seed, radius_scale=1.5)
    parts = [sc]
    for a in range(n_actors):
        A = actor_asset(per_actor, rng)
        r, ph = rng.uniform(8.0, 60.0), rng.uniform(0, 2 * np.pi)
        yaw = ph + np.pi / 2
        c, s = np.cos(yaw), np.sin(yaw)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        A["means"] = (A["means"].astype(np.float64) @ R.T + np.array([r * np.cos(ph), r * np.sin(ph), GROUND_Z])).astype(np.float32)
        qa = np.array([np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)])
        ql = A["rotations"].astype(np.float64)
        A["rotations"] = np.stack([qa[0] * ql[:, 0] - qa[3] * ql[:, 3], qa[0] * ql[:, 1] - qa[3] * ql[:, 2],
                                   qa[0] * ql[:, 2] + qa[3] * ql[:, 1], qa[0] * ql[:, 3] + qa[3] * ql[:, 0]], 1).astype(np.float32)
        parts.append(A)
    scene = {k: np.ascontiguousarray(np.concatenate([p[k] for p in parts], 0)) for k in sc}
    o, d = waymo_sensor_rays(frame)
    return scene, o, d
