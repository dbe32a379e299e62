"""Is the image of two independent tracer states bit-identical?  (round 6: bench.py's value_deferred_accum reported image_identical = False)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer

dev = torch.device("cuda:0")
sc, ro, rd = scenes.s1m() if os.environ.get("WORKLOAD", "s1m") == "s1m" else scenes.s10k()
H, W = ro.shape[:2]
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
o, d = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev); dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)


def run(deferred, warm_other_pose=False, n=3):
    tr = ShardedTracer(deferred_accum=deferred)
    if warm_other_pose:
        o2 = (o + torch.tensor([0.3, -0.2, 0.05], device=dev)).contiguous()
        for _ in range(2):
            tr.forward(o2, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg); tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, dL)
    for _ in range(n):
        out, _ = tr.forward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
        g = tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, dL)
    torch.cuda.synchronize()
    return out.clone(), {k: v.clone() for k, v in g.items()}


a, ga = run(False); b, gb = run(False); c, gc = run(True); e, ge = run(False, warm_other_pose=True)
for name, x in (("fresh vs fresh", b), ("fresh vs deferred", c), ("fresh vs state that saw another pose", e)):
    diff = (x != a)
    print(name, "identical" if not diff.any() else f"{int(diff.any(-1).sum())} rays differ, max abs {float((x - a).abs().max()):.3e}, channels {diff.reshape(-1, 9).any(0).tolist()}")
print("accum deferred vs atomics rel L2", float((gc["accum"] - ga["accum"]).norm() / ga["accum"].norm()))
for k in ("means", "shs"):
    print("grad", k, "rel L2 deferred vs not", float((gc[k] - ga[k]).norm() / ga[k].norm()), "fresh vs fresh", float((gb[k] - ga[k]).norm() / ga[k].norm()))
