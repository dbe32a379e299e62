#!/usr/bin/env python3
"""Developer tool: which composited-hit sequence is right on the dense translucent scene?  For the rays where the HIP
image differs from the oracle's, a brute-force numpy restatement of the reference's raygen loop (all quads of the ray,
sorted; chunks of 16 with the restart at t16 + 1e-5; forward.cu:146-308 as restated in oracle/lrt_oracle_impl.inc) is
compared with the hit record of the HIP forward and with the oracle's counts."""
import os, sys, ctypes as C
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.diff_lidar_tracer import Tracer
from tests.test_hip_parity import oracle_run
from tests.hip_util import settings, DEFAULT_OPTS

sc, o, d = scenes.dense_translucent()
H, W = o.shape[:2]
dL = scenes.upstream_grad(H, W, seed=2)
fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL, prec="f64")
tr = Tracer()
for k, v in {**DEFAULT_OPTS, "hit_cap": 1024}.items():
    tr.optix_context.set_option(k, v)
t = {k: torch.as_tensor(v, device="cuda:0") for k, v in sc.items()}
ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
out, _ = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
            scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
out = out.cpu().numpy()
st = tr.optix_context
idx, h = st.handle(torch.device("cuda:0"))
def read(which, n, dtype):
    buf = np.empty(n, dtype)
    st._lib.lrt_debug_read(h, which, buf.ctypes.data_as(C.c_void_p), buf.nbytes, None)
    return buf
cap = 1024
hit_n = read(5, H * W, np.int32); hit_t = read(6, H * W * cap, np.float32).reshape(H * W, cap); hit_g = read(7, H * W * cap, np.int32).reshape(H * W, cap)

# ---- brute force: all quads of one ray (float64), own intersection code, no tree (oracle/bruteforce.py)
from oracle.bruteforce import QuadScene, raygen_loop
qs = QuadScene(sc["means"], sc["scales"], sc["rotations"], sc["opacities"])
ray_candidates = qs.candidates


def reference_loop(g, tt, al):
    comp, _, _, drops = raygen_loop(g, tt, al)
    return [(c[0], c[1]) for c in comp], drops


err = np.abs(out - fw["out"]).reshape(-1, 9).max(1)
bad = np.argsort(-err)[:4]
for r in bad:
    g, tt, al = ray_candidates(o.reshape(-1, 3)[r].astype(np.float64), d.reshape(-1, 3)[r].astype(np.float64))
    comp, drops = reference_loop(g, tt, al)
    hg = hit_g[r, :hit_n[r]].tolist(); ht = hit_t[r, :hit_n[r]].tolist()
    eg = [c[0] for c in comp]
    first = next((i for i in range(min(len(hg), len(eg))) if hg[i] != eg[i]), None)
    print(f"ray {r}: err {err[r]:.2e}  candidates {len(g)} (oracle {fw['n_cand'].reshape(-1)[r]})  composited: oracle {fw['n_comp'].reshape(-1)[r]}  numpy-ref {len(eg)}  HIP {len(hg)}  epsilon drops in numpy-ref {drops}")
    if first is not None:
        print(f"   first difference at composited #{first}: HIP (g {hg[first]}, t {ht[first]:.7f})  numpy-ref (g {eg[first]}, t {comp[first][1]:.7f})")
        k = int(np.nonzero(g == eg[first])[0][0]); k2 = np.nonzero(g == hg[first])[0]
        print(f"   numpy-ref hit is candidate #{k} (chunk pos {k % 16}) alpha {al[k]:.6f}; HIP hit is candidate #{int(k2[0]) if len(k2) else None}; neighbours t: {tt[max(k-2,0):k+3].round(7).tolist()}")
