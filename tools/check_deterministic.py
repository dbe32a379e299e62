"""Option deterministic: are image, hit weights and gradients of independent tracer states bit-identical, whatever the states saw before?
usage: [WORKLOAD=s1m|s10k|s200k] python tools/check_deterministic.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer

dev = torch.device("cuda:0")
wl = os.environ.get("WORKLOAD", "s1m")
sc, ro, rd = getattr(scenes, wl)()
H, W = ro.shape[:2]
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
o, d = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev); dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)


def run(det, warm_other_pose=False, n=2, timeit=False):
    tr = ShardedTracer(deterministic=det)
    args = (t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    if warm_other_pose:
        o2 = (o + torch.tensor([0.3, -0.2, 0.05], device=dev)).contiguous()
        for _ in range(3):
            tr.forward(o2, d, *args); tr.backward(*args, dL)
    for _ in range(n):
        out, _ = tr.forward(o, d, *args); g = tr.backward(*args, dL)
    torch.cuda.synchronize()
    ms = None
    if timeit:
        t0 = time.perf_counter()
        for _ in range(50):
            tr.forward(o, d, *args); tr.backward(*args, dL)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 50
    return out.clone(), {k: v.clone() for k, v in g.items()}, ms


ref_o, ref_g, ms_nd = run(False, timeit=True)
a, ga, ms_d = run(True, timeit=True)
print(wl, "step ms: default %.4f deterministic %.4f" % (ms_nd, ms_d))
for name, (b, gb, _) in (("fresh vs fresh", run(True)), ("fresh vs a state that saw another pose", run(True, warm_other_pose=True)), ("one step vs five", run(True, n=5))):
    bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
    print("deterministic,", name + ":", "image", "identical" if torch.equal(a, b) else "DIFFERS (%d rays)" % int((a != b).any(-1).sum()),
          "| gradients + accum", "identical" if not bad else "DIFFER: " + ", ".join("%s %d elements" % (k, int((ga[k] != gb[k]).sum())) for k in bad))
for k in ga:
    den = float(ref_g[k].double().norm())
    print("  deterministic vs default, %-10s rel L2 %.3e" % (k, float((ga[k].double() - ref_g[k].double()).norm()) / max(den, 1e-30)))
print("  image max abs vs default %.3e" % float((a - ref_o).abs().max()))
