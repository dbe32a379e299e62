#!/usr/bin/env python3
"""Developer tool: inspect the LBVH built on the GPU and compare cull / no-cull traces with the oracle."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from lidar_rt_amd import scenes, _capi                              # noqa: E402
from lidar_rt_amd.diff_lidar_tracer import Tracer                   # noqa: E402
from oracle import oracle                                           # noqa: E402
from tools.gpu_probe import run_hip, dev                            # noqa: E402


def dbg_read(tr, which, dtype, shape):
    st = tr.optix_context
    idx, h = st.handle(dev)
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = np.empty(shape, dtype)
    got = st._lib.lrt_debug_read(h, which, buf.ctypes.data_as(C.c_void_p), n, None)
    assert got == n, (which, got, n)
    return buf


def main():
    sc, o, d = scenes.s10k()
    P = sc["means"].shape[0]
    bg = scenes.BG_DEFAULT
    tr = Tracer()
    h = run_hip(tr, sc, o, d, 3, bg)
    order = dbg_read(tr, 0, np.uint32, (P,))
    print("order is permutation:", np.array_equal(np.sort(order), np.arange(P)), "n unique", len(np.unique(order)))
    rec = dbg_read(tr, 1, np.float32, (P, 16))
    gid = rec[:, 11].view(np.int32)
    print("rec gidx == order:", np.array_equal(gid, order.astype(np.int32)))
    print("rec mu == means[order]:", np.array_equal(rec[:, 4:7], sc["means"][order]))
    aabb = dbg_read(tr, 3, np.float32, (P, 6))
    nl = (P + 7) // 8
    cnt = []; c = nl
    while True:
        c = max((c + 7) // 8, 1); cnt.append(c)
        if c == 1: break
    L = len(cnt)                       # cnt[l-1] nodes at level l
    off = [0] * (L + 1); o_ = 0
    for l in range(L, 0, -1):
        off[l] = o_; o_ += cnt[l - 1]
    nodes = dbg_read(tr, 2, np.float32, (o_, 64))
    print("levels", L, "cnt", cnt, "off", off[1:], "total", o_)
    hdr = nodes[:, 48:50].view(np.int32)
    # level 1: children are leaves
    ok = True
    for j in range(cnt[0]):
        nd = nodes[off[1] + j]
        if hdr[off[1] + j, 0] != 8 * j or hdr[off[1] + j, 1] != 1: ok = False
        for cch in range(8):
            leaf = 8 * j + cch
            ks = np.arange(leaf * 8, min(leaf * 8 + 8, P))
            lo = nd[[cch, 8 + cch, 16 + cch]]; hi = nd[[24 + cch, 32 + cch, 40 + cch]]
            if len(ks) == 0:
                if not (lo[0] >= 1e30 and hi[0] >= 1e30): ok = False
                continue
            if not (np.all(aabb[ks, :3].min(0) == lo) and np.all(aabb[ks, 3:].max(0) == hi)): ok = False
    print("level-1 boxes/hdr ok:", ok)
    for l in range(2, L + 1):
        ok = True
        for j in range(cnt[l - 1]):
            nd = nodes[off[l] + j]
            if hdr[off[l] + j, 0] != off[l - 1] + 8 * j or hdr[off[l] + j, 1] != 0: ok = False
            for cch in range(8):
                ch = 8 * j + cch
                lo = nd[[cch, 8 + cch, 16 + cch]]; hi = nd[[24 + cch, 32 + cch, 40 + cch]]
                if ch >= cnt[l - 2]:
                    if not (lo[0] >= 1e30 and hi[0] >= 1e30): ok = False
                    continue
                cn = nodes[off[l - 1] + ch]
                v = cn[0:8] < 1e30; clo = np.array([cn[0:8][v].min(), cn[8:16][v].min(), cn[16:24][v].min()]); chi = np.array([cn[24:32][v].max(), cn[32:40][v].max(), cn[40:48][v].max()])
                if not (np.all(clo == lo) and np.all(chi == hi)): ok = False
        print(f"level-{l} boxes/hdr ok:", ok)

    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], "f32")
    fw = orc.forward(o, d, sc["shs"], 3, bg, stats=True)
    def report(tag, hh):
        err = np.abs(hh["out"] - fw["out"]).max(-1)
        print(f"{tag}: bad rays {(err > 1e-3).sum()} / {err.size}; max err {err.max():.3e}; "
              f"W mean hip {hh['out'][...,4].mean():.5f} orc {fw['out'][...,4].mean():.5f}")
        return err
    e1 = report("cull", h)
    tr.optix_context.set_option("no_cull", 1)
    h2 = run_hip(tr, sc, o, d, 3, bg)
    e2 = report("no_cull", h2)
    tr.optix_context.set_option("no_cull", 0)
    for tw in (64, 8, 1):
        tr.optix_context.set_option("tile_w", tw)
        report(f"tile_w={tw}", run_hip(tr, sc, o, d, 3, bg))
    tr.optix_context.set_option("tile_w", 16)
    bad = np.argwhere(e1 > 1e-3)
    print("bad rows hist", np.bincount(bad[:, 0], minlength=16).tolist())
    print("bad cols (first row)", bad[bad[:, 0] == bad[0, 0], 1][:40].tolist())
    # one ray alone
    hh, ww = bad[0]
    h3 = run_hip(tr, sc, o[hh:hh+1, ww:ww+1], d[hh:hh+1, ww:ww+1], 3, bg)
    print("single ray hip", h3["out"][0, 0].tolist()); print("single ray orc", fw["out"][hh, ww].tolist())


if __name__ == "__main__" and "--lists" not in sys.argv:
    main()


def debug_ray_lists():
    sc, o, d = scenes.s10k()
    bg = scenes.BG_DEFAULT
    tr = Tracer()
    tr.optix_context.set_option("debug_rays", 4096)
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], "f32")
    fw = orc.forward(o, d, sc["shs"], 3, bg, stats=True)
    res = {}
    for mode in (0, 1, 2, 3):
        tr.optix_context.set_option("no_cull", mode)
        h = run_hip(tr, sc, o, d, 3, bg)
        dbg = dbg_read(tr, 4, np.float32, (16, 256, 32, 2))
        res[mode] = (h, dbg)
    e = np.abs(res[0][0]["out"] - fw["out"]).max(-1)
    bad = np.argwhere(e > 1e-3)
    for hh, ww in bad[:4]:
        for mode in (0, 1, 2, 3):
            dbg = res[mode][1][hh, ww]
            n = int((dbg[:, 0] > 0).sum())
            print(f"ray ({hh},{ww}) mode no_cull={mode}: n={n}")
            print("   t:", np.round(dbg[:n, 0], 4).tolist())
            print("   g:", dbg[:n, 1].view(np.int32).tolist())
        print("   oracle ncand", int(fw["n_cand"][hh, ww]))


if __name__ == "__main__" and "--lists" in sys.argv:
    debug_ray_lists()
