#!/usr/bin/env python3
"""Developer tool: per-rank compute time of an N-way azimuth split, measured on ONE GPU by tracing a single slab
(rank 0's columns) of the S1M frame: build / forward / backward HIP-event times for N = 1, 2, 4, 8."""
import os, sys
os.environ.setdefault("LRT_PREZERO", "force")      # a rank of a split keeps its gradient buffer zero by list (what the exchange path does)
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend, column_slab, GradLayout

dev = torch.device("cuda:0")
sc, ro, rd = scenes.waymo_dynamic_4m() if os.environ.get("WORKLOAD", "s1m") == "waymo4m" else scenes.s1m()      # WORKLOAD=waymo4m: BASELINE configs[4]'s shape
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
H, W = ro.shape[:2]
dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
from lidar_rt_amd.parallel import ShardedTracer
tr = ShardedTracer()                              # one rank: what a rank of an N-way split does for ITS slab (build, forward, backward incl. the prezero protocol)
be = tr.backend
for kv in os.environ.get("LRT_OPTS", "").split(","):
    if kv: be.state.set_option(kv.split("=")[0], int(kv.split("=")[1]))
lay = GradLayout(int(sc["means"].shape[0]), 16, dev)
grads = {k: lay.views[k] for k in ("means", "scales", "rotations", "opacities", "shs")}
NS = tuple(int(x) for x in os.environ.get("SLAB_N", "1,2,4,8").split(","))
for N in NS:
    for r in sorted({0, N // 2}):
        a, b = column_slab(W, r, N)
        o = torch.as_tensor(ro[:, a:b].copy(), device=dev); d = torch.as_tensor(rd[:, a:b].copy(), device=dev)
        g = dL[:, a:b].contiguous()
        tr.cull_build = N >= int(os.environ.get("CULL_FROM", "4")) and os.environ.get("CULL", "0") == "1"
        be.state.enable_timing(True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n_it, n_warm = 8, 3
        for it in range(n_it):
            if it == n_warm: be.state.get_timing(dev); ev[0].record()     # drop the warm-up samples
            tr.forward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, cull_key=(N, r))
            tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, g)
        ev[1].record()
        torch.cuda.synchronize()
        tm = be.state.get_timing(dev)
        # the whole step once more WITHOUT the library's region timers (8 event records per step on the launch stream cost 30-40 us at N=8)
        be.state.enable_timing(False)
        ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for it in range(n_it + 4):
            if it == n_warm: ev2[0].record()
            tr.forward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, cull_key=(N, r))
            tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, g)
        ev2[1].record(); torch.cuda.synchronize()
        whole = ev2[0].elapsed_time(ev2[1]) / (n_it + 4 - n_warm)
        f = lambda k: tm[k][0] / max(tm[k][1], 1)
        kept = be.state.built_count(dev)
        print(f"N={N} rank {r}: cols {b - a:4d}  kept {kept:7d}  build {f('build'):.3f}  fwd {f('fwd'):.3f}  bwd {f('bwd'):.3f}  sum {f('build') + f('fwd') + f('bwd'):.3f} ms"
              f"   whole per-rank step (events around build + forward + backward + list bookkeeping): {whole:.3f} ms (with the library's region timers on: {ev[0].elapsed_time(ev[1]) / (n_it - n_warm):.3f})")

# ---- local cost of the owner-based gradient exchange for one rank (owner map + listing / packing of the foreign rows), and how
# many rows a rank would send: the network leg itself cannot be measured on one GPU
import ctypes as C
from lidar_rt_amd import _capi
lib = _capi.load(); p = _capi.ptr
for N in ((2, 4, 8) if os.environ.get("SLAB_OWNER", "0") == "1" and hasattr(lib, "lrt_owner_by_direction") else ()):      # (rounds 2-5: the owner exchange's device helpers, removed from the library in round 6)
    st = ShardedTracer.__new__(ShardedTracer); st.world = N; st.rank = 0
    st._rays_full = (torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev))
    a, b = column_slab(W, 0, N)
    o = torch.as_tensor(ro[:, a:b].copy(), device=dev); d = torch.as_tensor(rd[:, a:b].copy(), device=dev)
    be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
    out, acc = be.forward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    be.backward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, out, dL[:, a:b].contiguous(), grads_out=grads)
    lay.views["accum"].copy_(acc)
    origin, axes = st.slab_axes()
    owner = torch.empty(1000000, dtype=torch.int32, device=dev)
    cap = 8192
    cnt = torch.zeros(N, dtype=torch.int32, device=dev); idx = torch.empty((N, cap), dtype=torch.int32, device=dev)
    rows = torch.empty((N, cap, lay.width), dtype=torch.float32, device=dev)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for rep in range(3):
        e0.record()
        lib.lrt_owner_by_direction(0, 1000000, p(t["means"]), p(origin), N, p(axes), p(owner), s)
        e1.record()
        v = lay.views
        lib.lrt_grad_pack_foreign(0, 1000000, 16, N, 0, cap, p(owner), p(v["means"]), p(v["scales"]), p(v["rotations"]), p(v["opacities"]), p(v["shs"]),
                                  p(v["accum"]), p(idx), p(cnt), p(rows), s)
        e2.record()
    torch.cuda.synchronize()
    touched = int((acc > 0).sum()); own = int(((acc > 0) & (owner == 0)).sum())
    print(f"N={N} rank 0 owner exchange: owner map {e0.elapsed_time(e1) * 1e3:.0f} us, list + pack {e1.elapsed_time(e2) * 1e3:.0f} us; touched {touched}, "
          f"of which owned by the rank itself {own} ({100.0 * own / max(touched, 1):.1f} %), rows sent {cnt.tolist()} = {int(cnt.sum()) * lay.width * 4 / 1e6:.1f} MB "
          f"(a replicating exchange would gather {touched * lay.width * 4 / 1e6:.1f} MB from every rank)")
