import sys, numpy as np
sys.path.insert(0, "/root/repo")
from lidar_rt_amd import scenes
from oracle import oracle
from tests.hip_util import run_hip, rel_l2
from tests.test_hip_parity import oracle_run
rng = np.random.default_rng(5)
P = 6000
sc = scenes.make_scene(P, seed=12, radius_scale=0.3)
sc["means"] = (rng.uniform(-2.5, 2.5, (P, 3)) * np.array([1, 1, 0.4])).astype(np.float32)
sc["scales"] = (sc["scales"] * 0.5).astype(np.float32)
o, d = scenes.kitti_rays(8, 96)
dL = scenes.upstream_grad(8, 96, seed=4)
oracle.set_sorted_anyhit(True)
fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
oracle.set_sorted_anyhit(False)
for mode in ({"fwd_mode": 2, "defer_colour": 1}, {"fwd_mode": 2, "defer_colour": 1, "bwd_mode": 2}, {"fwd_mode": 2, "defer_colour": 1, "bwd_mode": 0}, {"fwd_mode": 0, "bwd_mode": 0}, {"fwd_mode": 0, "bwd_mode": 2}):
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts=mode)
    print(mode, {k: round(rel_l2(h["grads"][k].reshape(bw[k].shape), bw[k]), 4) for k in ("means", "scales", "rotations", "opacities", "shs")})

from oracle.bruteforce import QuadScene
qs = QuadScene(sc["means"], sc["scales"], sc["rotations"], sc["opacities"])
near = np.array([bool((qs.candidates(oo, dd)[1] < 0.2).any()) for oo, dd in zip(o.reshape(-1, 3), d.reshape(-1, 3))]).reshape(o.shape[:2])
print("near rays", near.sum(), "of", near.size)
for name, mask in (("only near rays", near), ("only far rays", ~near)):
    dl2 = dL * mask[..., None]
    oracle.set_sorted_anyhit(True)
    fw2, bw2 = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dl2)
    oracle.set_sorted_anyhit(False)
    for mode in ({"fwd_mode": 2, "defer_colour": 1, "bwd_mode": 0}, {"fwd_mode": 2, "defer_colour": 1, "bwd_mode": 3}):
        h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dl2, opts=mode)
        print(name, mode["bwd_mode"], {k: round(rel_l2(h["grads"][k].reshape(bw2[k].shape), bw2[k]), 4) for k in ("means", "scales", "rotations", "opacities", "shs")})
