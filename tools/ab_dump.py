"""Dump (or compare with a dump) the forward image, hit weights and gradients of one step: python tools/ab_dump.py dump|cmp FILE   (WORKLOAD as in ab_outputs.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer
dev = torch.device("cuda:0")
wl = os.environ.get("WORKLOAD", "s10k")
sc, ro, rd = getattr(scenes, wl)()
H, W = ro.shape[:2]
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
o, d = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev); dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)
tr = ShardedTracer()
out, acc = tr.forward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
g = tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, dL)
torch.cuda.synchronize()
import ctypes as C
def hits(state, HW):
    idx, h = state.handle(dev)
    cap = state.get_option("hit_cap", dev)
    hn = np.empty(HW, np.int32); hg = np.empty((HW, cap), np.int32); ht = np.empty((HW, cap), np.float32); wa = np.empty((HW, cap, 2), np.float32)
    state._lib.lrt_debug_read.restype = C.c_longlong
    for which, arr in ((5, hn), (6, ht), (7, hg), (8, wa)):
        state._lib.lrt_debug_read(h, which, arr.ctypes.data_as(C.c_void_p), C.c_longlong(arr.nbytes), None)
    return hn, ht, hg, wa
hn, ht, hg, wa = hits(tr.backend.state, H * W)
cur = {"hn": hn, "ht": ht, "hg": hg, "wa": wa, "out": out.cpu().numpy(), "accum": acc.cpu().numpy(), **{"g_" + k: v.cpu().numpy() for k, v in g.items() if k != "accum"}}
if sys.argv[1] == "dump":
    np.savez(sys.argv[2], **cur)
else:
    ref = np.load(sys.argv[2])
    for k in cur:
        a, b = cur[k], ref[k]
        print(k, "identical" if np.array_equal(a, b) else "DIFFER: %d elements, max abs %.3e" % (int((a != b).sum()), float(np.abs(a - b).max())))
    df = (cur["out"] != ref["out"]).any(-1)
    ys, xs = np.nonzero(df)
    print("rays that differ:", len(ys), "rows", np.bincount(ys, minlength=H)[:16], "cols mod 8", np.bincount(xs % 8, minlength=8))
    for y, x in list(zip(ys, xs))[:6]:
        print((y, x), "this", cur["out"][y, x, 3:], "ref", ref["out"][y, x, 3:])
    for y, x in list(zip(ys, xs))[:3]:
        r = y * W + x
        print("ray", (y, x), "n", cur["hn"][r], ref["hn"][r])
        for j in range(max(cur["hn"][r], ref["hn"][r])):
            print("   ", j, "this", cur["ht"][r, j], cur["hg"][r, j], cur["wa"][r, j], " ref", ref["ht"][r, j], ref["hg"][r, j], ref["wa"][r, j])
