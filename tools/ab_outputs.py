"""Digest of the forward image, the hit weights and the gradients of a few steps -- run once per library variant (LRT_HIP_LIB) and compare the lines:
a variant that claims bit-identical results must print the same digests.  usage: [WORKLOAD=s1m|s10k|s200k|waymo4m] python tools/ab_outputs.py [steps]"""
import sys, os, hashlib
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
tr = ShardedTracer()
dig = lambda x: hashlib.sha256(x.detach().cpu().numpy().tobytes()).hexdigest()[:12]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out, acc = tr.forward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    g = tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, dL)
    torch.cuda.synchronize()
    print(wl, "step", it, "image", dig(out), "accum-sum %.9e" % float(acc.double().sum()), "image-sum %.12e" % float(out.double().sum()),
          "grads", {k: "%.7e" % float(v.double().abs().sum()) for k, v in g.items() if k != "accum"})
