"""Write a SYNTHETIC LiDAR sequence in the on-disk layout of lidar_rt_amd/sequence.py, shaped like one of BASELINE.json's dataset configs
(the datasets themselves are not available here): the ground-truth range images are rendered from a seeded Gaussian scene with the HIP tracer
(renderer.raytracing in evaluation mode), the training run then starts from the back-projected returns like the reference does.

    python tools/make_sequence.py --shape kitti360_dynamic --out /tmp/seq --frames 8 [--scale 0.1]

  waymo_static      configs[2]: ~2 M background Gaussians, 64 x 2650 Waymo-style grid (per-beam inclinations, posed sensor)
  kitti360_dynamic  configs[3]: 66 x 1030, 500 k background + 8 rigid actors x 8 k with a pose per frame
  waymo_dynamic     configs[4]: ~3.9 M background + 10 actors x 10 k under the Waymo grid
--scale shrinks the Gaussian counts (tests); the image sizes stay the datasets'.
"""
import argparse
import math
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from lidar_rt_amd import renderer, scenes, sequence


class _GT:
    """A fixed Gaussian set with the attribute surface renderer.raytracing reads (getter chain: activated values as they are)."""

    def __init__(self, sc, dev, box=None):
        t = lambda a: torch.as_tensor(a, device=dev)
        self.xyz, self.sc, self.rot, self.op, self.sh = t(sc["means"]), t(sc["scales"]), t(sc["rotations"]), t(sc["opacities"]), t(sc["shs"])
        self.active_sh_degree, self.bounding_box = 3, box
        self.get_scaling, self.get_opacity, self.get_features = self.sc, self.op, self.sh

    def _pose(self, ts):
        return self.bounding_box.frame[ts] if (self.bounding_box is not None and ts in self.bounding_box.frame) else None

    def get_rotation(self, ts=0.0):
        ps = self._pose(ts)
        return (ps[1] if ps is not None else torch.zeros((1, 4), device=self.xyz.device)), torch.nn.functional.normalize(self.rot)

    def get_world_xyz(self, ts=0.0):
        from lidar_rt_amd.training import _rotation_matrix
        ps = self._pose(ts)
        return self.xyz if ps is None else self.xyz @ _rotation_matrix(ps[1].reshape(1, 4)).squeeze(0).T + ps[0]


def make(shape: str, out: str, n_frames: int, scale: float = 1.0, device="cuda:0", test_every: int = 0):
    dev = torch.device(device)
    n = lambda x: max(int(x * scale), 64)
    boxes, actors = None, []
    if shape == "kitti360_dynamic":
        H, W = scenes.KITTI360_HW
        bg, act, poses_of, _ = scenes.kitti360_dynamic(P_bg=n(500_000), n_actors=8, per_actor=n(8000))
        data_type, s2e, inc = "KITTI", None, (math.radians(-24.9), math.radians(2.0))
        pose_s = lambda f: scenes.pose_matrix((0.5 * f, 0.0, 0.0), yaw=0.01 * f)
        size = np.tile(np.array([[4.6, 2.1, 1.8]], np.float32), (len(act), 1))
        tr = np.stack([[poses_of(f)[a][0] for f in range(n_frames)] for a in range(len(act))])
        qu = np.stack([[poses_of(f)[a][1] / np.linalg.norm(poses_of(f)[a][1]) for f in range(n_frames)] for a in range(len(act))])
        # the actor frame of scenes.actor_asset has its floor at z = 0: centre the tracking box on the body
        boxes = {"frames": list(range(n_frames)), "translation": tr, "quaternion": qu, "size": size + np.array([[0, 0, 1.8]], np.float32)}
        actors = act
    else:
        H, W = scenes.WAYMO_HW
        data_type, s2e, inc = "Waymo", scenes.pose_matrix((1.43, 0.0, 2.18), yaw=0.02), scenes.waymo_inclinations(H)
        pose_s = lambda f: scenes.pose_matrix((0.8 * f, 0.1 * f, 0.45), yaw=0.03 * f + 0.4, pitch=0.01, roll=-0.008)
        if shape == "waymo_static":
            bg = scenes.make_scene(n(2_000_000), seed=scenes.SEED + 7, radius_scale=1.25)
        elif shape == "waymo_dynamic":
            rng = np.random.default_rng(scenes.SEED + 13)
            bg = scenes.make_scene(n(3_900_000), seed=scenes.SEED + 13, radius_scale=1.5)
            actors = [scenes.actor_asset(n(10_000), rng) for _ in range(10)]
            r, ph, om = rng.uniform(8.0, 60.0, 10), rng.uniform(0, 2 * np.pi, 10), rng.uniform(-0.03, 0.03, 10)
            tr = np.stack([[[r[a] * np.cos(ph[a] + om[a] * f), r[a] * np.sin(ph[a] + om[a] * f), scenes.GROUND_Z] for f in range(n_frames)] for a in range(10)]).astype(np.float32)
            yaw = np.stack([[ph[a] + om[a] * f + np.pi / 2 for f in range(n_frames)] for a in range(10)])
            qu = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)], -1).astype(np.float32)
            boxes = {"frames": list(range(n_frames)), "translation": tr, "quaternion": qu, "size": np.tile(np.array([[4.6, 2.1, 3.6]], np.float32), (10, 1))}
        else:
            raise SystemExit(f"unknown shape {shape}")
    tbs = []
    if boxes is not None:
        for a in range(len(actors)):
            tb = sequence.TrackingBox(boxes["size"][a], dev)
            for k, f in enumerate(boxes["frames"]):
                tb.frame[f] = (torch.as_tensor(boxes["translation"][a, k], device=dev), torch.as_tensor(boxes["quaternion"][a, k], device=dev).reshape(1, 4), None, None)
            tbs.append(tb)
    assets = [_GT(bg, dev)] + [_GT(a_, dev, tb) for a_, tb in zip(actors, tbs)]
    args = types.SimpleNamespace(dynamic=boxes is not None, opt=types.SimpleNamespace(use_rayhit=True), pipe=types.SimpleNamespace())      # ray-drop = softmax([hit, drop]): a return where the accumulated weight exceeds the final transmittance
    from lidar_rt_amd.training import RangeFrames
    old = renderer.tracer_2dgs, renderer.use_fused_preprocess, renderer.deferred_accum
    renderer.tracer_2dgs, renderer.use_fused_preprocess, renderer.deferred_accum = None, False, False
    frames = []
    try:
        with torch.no_grad():
            for f in range(n_frames):
                s2w = torch.as_tensor(pose_s(f), device=dev)
                o, d = RangeFrames.range_rays(H, W, [float(x) for x in inc] if len(inc) > 2 else (float(inc[0]), float(inc[1])), s2w, data_type,
                                              None if s2e is None else torch.as_tensor(s2e, device=dev))
                renderer.tracer_2dgs = renderer.tracer_2dgs or renderer.Tracer()
                renderer.tracer_2dgs.eval()
                pkg = renderer.raytracing(f, assets, (o, d, o[0, 0]), torch.tensor([0.0, 0.0, 1.0]), args)
                drop = pkg["raydrop"].squeeze(-1)
                mask = (drop < 0.5) & (pkg["depth"].squeeze(-1) > 0.2)
                frames.append({"id": f, "depth": pkg["depth"].squeeze(-1) * mask, "intensity": pkg["intensity"].squeeze(-1).clamp(0, 1) * mask, "mask": mask,
                               "inclination": np.asarray(inc, np.float32), "sensor2world": s2w})
    finally:
        renderer.tracer_2dgs, renderer.use_fused_preprocess, renderer.deferred_accum = old
    test = [f for f in range(n_frames) if test_every and f % test_every == test_every - 1]
    return sequence.write_sequence(out, frames, data_type=data_type, extent=float(np.abs(bg["means"]).max()), sensor2ego=s2e, boxes=boxes, test_frames=test)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--shape", required=True, choices=["waymo_static", "kitti360_dynamic", "waymo_dynamic"])
    ap.add_argument("--out", required=True)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--test-every", type=int, default=0)
    a = ap.parse_args()
    print(make(a.shape, a.out, a.frames, a.scale, test_every=a.test_every))
