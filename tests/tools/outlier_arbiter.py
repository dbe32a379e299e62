#!/usr/bin/env python3
"""Developer probe: for rays where only the HIP image leaves the fp64 oracle (the fp32 oracle does not), compare the HIP hit
record with the brute-force float64 restatement of the raygen loop (oracle/bruteforce.py) hit by hit."""
import os, sys, ctypes as C
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend
from oracle import oracle
from oracle.bruteforce import QuadScene, raygen_loop
dev = torch.device("cuda:0")
sc, o, d = scenes.waymo_frame(); o = np.ascontiguousarray(o[:, ::8]); d = np.ascontiguousarray(d[:, ::8])
res = {}
for prec in ("f32", "f64"):
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec)
    res[prec] = orc.forward(o, d, sc["shs"], 3, scenes.BG_DEFAULT, stats=True)["out"].reshape(-1, 9).astype(np.float64)
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
be = HipBackend()
be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
out, _ = be.forward(torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev), t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3,
                    torch.as_tensor(scenes.BG_DEFAULT, device=dev))
torch.cuda.synchronize()
H_ = out.cpu().numpy().reshape(-1, 9).astype(np.float64)
idx, hd = be.state.handle(dev)
HW = o.shape[0] * o.shape[1]; cap = 256
hn = np.empty(HW, np.int32); ht = np.empty((HW, cap), np.float32); hg = np.empty((HW, cap), np.int32)
be.state._lib.lrt_debug_read.restype = C.c_longlong
for which, arr in ((5, hn), (6, ht), (7, hg)):
    be.state._lib.lrt_debug_read(C.c_void_p(hd), which, arr.ctypes.data_as(C.c_void_p), C.c_longlong(arr.nbytes), None)
ref = res["f64"]
sc0 = np.maximum(np.abs(ref[:, 0]), 1e-3 * np.abs(ref[:, 0]).max())
eh = np.abs(H_[:, 0] - ref[:, 0]) / sc0; ef = np.abs(res["f32"][:, 0] - ref[:, 0]) / sc0
sel = np.nonzero((eh > 1e-4) & (ef < 1e-5))[0]
qs = QuadScene(sc["means"], sc["scales"], sc["rotations"], sc["opacities"])
kinds = {}
for r in sel[:40]:
    g, tt, al = qs.candidates(o.reshape(-1, 3)[r], d.reshape(-1, 3)[r])
    comp, T, consumed, drops = raygen_loop(g, tt, al)
    eg = [c[0] for c in comp]; hgl = hg[r, :hn[r]].tolist()
    if eg == hgl:
        kind = "same sequence"
    elif sorted(eg) == sorted(hgl):
        i = next(i for i in range(len(eg)) if eg[i] != hgl[i])
        kind = "permutation"; extra = f"first at #{i}: exact t {comp[i][1]:.8f} vs next {comp[i + 1][1]:.8f} (gap {comp[i + 1][1] - comp[i][1]:.2e}); hip t {ht[r, i]:.8f} {ht[r, i + 1]:.8f}"
    else:
        miss = [x for x in eg if x not in hgl]; add = [x for x in hgl if x not in eg]
        kind = "different set"; extra = f"missing in hip {miss[:3]} extra in hip {add[:3]} drops(ref) {drops[:3]}"
    kinds[kind] = kinds.get(kind, 0) + 1
    if kinds[kind] <= 4:
        print(f"ray {r}: {kind}; err hip {eh[r]:.2e}; n ref {len(eg)} hip {len(hgl)}; " + (extra if kind != "same sequence" else ""))
print(kinds)
