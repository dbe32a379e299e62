#!/usr/bin/env python3
"""Developer probe (test infrastructure; uses the oracle): where do the HIP path and the fp32 oracle leave the fp64 oracle?"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from oracle import oracle
from tests.hip_util import run_hip
sc = scenes.make_scene(200_000, radius_scale=0.5); o, d = scenes.kitti_rays(32, 512)
if os.environ.get("SHIFT"):
    o = o + np.array([0.8, 0.1, 0.45], np.float32)
if os.environ.get("ROT"):
    R = scenes.pose_matrix((0, 0, 0), yaw=0.4, pitch=0.01, roll=-0.008)[:3, :3]
    d = (d @ R.T).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
if os.environ.get("WAYMO"):
    sc, o, d = scenes.waymo_frame(); o = np.ascontiguousarray(o[:, ::8]); d = np.ascontiguousarray(d[:, ::8])
res = {}
for prec in ("f32", "f64"):
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec)
    res[prec] = orc.forward(o, d, sc["shs"], 3, scenes.BG_DEFAULT, stats=True)
opts = dict(kv.split("=") for kv in os.environ.get("LRT_OPTS", "").split(",") if kv)
h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, opts={k: int(v) for k, v in opts.items()})
for ch, name in ((3, "depth"), (4, "weight"), (0, "intensity")):
    ref = res["f64"]["out"][..., ch].astype(np.float64)
    scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max())
    eh = np.abs(h["out"][..., ch] - ref) / scale; ef = np.abs(res["f32"]["out"][..., ch] - ref) / scale
    print(name, "hip: frac>1e-4 %.2e  >1e-3 %.2e  >1e-2 %.2e | f32: %.2e %.2e %.2e" % ((eh > 1e-4).mean(), (eh > 1e-3).mean(), (eh > 1e-2).mean(),
          (ef > 1e-4).mean(), (ef > 1e-3).mean(), (ef > 1e-2).mean()))
    print("   quantiles hip", np.quantile(eh, [0.5, 0.9, 0.99, 0.999]), " f32", np.quantile(ef, [0.5, 0.9, 0.99, 0.999]))
    both = (eh > 1e-4) & (ef > 1e-4)
    print("   pixels beyond 1e-4: hip only %d, f32 only %d, both %d" % (((eh > 1e-4) & ~both).sum(), ((ef > 1e-4) & ~both).sum(), both.sum()))
# composited-hit counts per ray: HIP hit record vs the oracles
import ctypes as C
from lidar_rt_amd.parallel import HipBackend
dev = torch.device("cuda:0")
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
be = HipBackend()
be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
be.forward(torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev), t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3,
           torch.as_tensor(scenes.BG_DEFAULT, device=dev))
torch.cuda.synchronize()
idx, hd = be.state.handle(dev)
HW = o.shape[0] * o.shape[1]
hn = np.empty(HW, np.int32)
be.state._lib.lrt_debug_read.restype = C.c_longlong
be.state._lib.lrt_debug_read(C.c_void_p(hd), 5, hn.ctypes.data_as(C.c_void_p), C.c_longlong(hn.nbytes), None)
n32 = res["f32"]["n_comp"].reshape(-1); n64 = res["f64"]["n_comp"].reshape(-1)
print("rays with a different number of composited hits: hip vs f32 %d, hip vs f64 %d, f32 vs f64 %d (of %d)" % (
    (hn != n32).sum(), (hn != n64).sum(), (n32 != n64).sum(), HW))
print("   sign of (hip - f64):", int((hn > n64).sum()), "more,", int((hn < n64).sum()), "fewer;  (f32 - f64):", int((n32 > n64).sum()), "more,", int((n32 < n64).sum()), "fewer")
c32 = res["f32"]["n_cand"].reshape(-1); c64 = res["f64"]["n_cand"].reshape(-1)
print("   candidates differ f32 vs f64 on %d rays" % (c32 != c64).sum())
ref = res["f64"]["out"].reshape(-1, 9).astype(np.float64); H_ = h["out"].reshape(-1, 9).astype(np.float64); F_ = res["f32"]["out"].reshape(-1, 9).astype(np.float64)
sc0 = np.maximum(np.abs(ref[:, 0]), 1e-3 * np.abs(ref[:, 0]).max())
eh = np.abs(H_[:, 0] - ref[:, 0]) / sc0; ef = np.abs(F_[:, 0] - ref[:, 0]) / sc0
sel = np.nonzero((eh > 1e-4) & (ef < 1e-5))[0]
print("hip-only intensity outliers:", len(sel), " of which same n_comp as f64:", int((hn[sel] == n64[sel]).sum()))
print("   their depth rel err (hip vs f64): median %.1e, weight: median %.1e, T: median %.1e" % (
    np.median(np.abs(H_[sel, 3] - ref[sel, 3]) / np.maximum(ref[sel, 3], 1e-3)), np.median(np.abs(H_[sel, 4] - ref[sel, 4])), np.median(np.abs(H_[sel, 8] - ref[sel, 8]))))
print("   intensity err distribution:", np.quantile(eh[sel], [0.1, 0.5, 0.9]), " n_comp median", np.median(n64[sel]), " rows:", np.bincount(sel // o.shape[1], minlength=o.shape[0]).tolist())
for r in sel[:6]:
    print("   ray", r, "row", r // o.shape[1], "n", hn[r], n32[r], n64[r], "I hip/f32/f64 %.6f %.6f %.6f  D %.5f %.5f %.5f  T %.2e %.2e" % (H_[r, 0], F_[r, 0], ref[r, 0], H_[r, 3], F_[r, 3], ref[r, 3], H_[r, 8], ref[r, 8]))
