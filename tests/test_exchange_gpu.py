"""GPU: the device-side gathering exchange (include/lrt.h lrt_xchg_pack / lrt_xchg_apply; lidar_rt_amd/parallel.py `_exchange_lists`)
against plain torch sums in rank order, on one device: N simulated ranks, each with its own partial gradient buffer."""
import ctypes as C

import numpy as np
import pytest
import torch

from lidar_rt_amd import _capi
from lidar_rt_amd.parallel import GradLayout

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dense(lay):
    v = lay.views
    return [_capi.ptr(v[k]) for k in ("means", "scales", "rotations", "opacities", "shs", "accum")]


def _make_ranks(P, M, N, frac, seed):
    """N partial buffers: rank r touched a random subset (accum > 0 exactly there), rows random; neighbouring ranks overlap."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lays, masks = [], []
    for r in range(N):
        lay = GradLayout(P, M, DEV)
        lay.flat.zero_()
        mask = torch.rand(P, generator=g) < frac
        if r > 0:                                                         # some Gaussians are touched by several ranks
            mask |= masks[r - 1].cpu() & (torch.rand(P, generator=g) < 0.3)
        idx = torch.nonzero(mask).squeeze(1).to(DEV)
        for k, t in lay.views.items():
            w = t.reshape(P, -1).shape[1]
            vals = torch.randn((idx.numel(), w), generator=g).to(DEV)
            if k == "accum":
                vals = vals.abs() + 0.1
            t.reshape(P, -1)[idx] = vals
        lays.append(lay); masks.append(mask.to(DEV))
    return lays, masks


@pytest.mark.parametrize("P,M,N", [(5000, 16, 3), (1024, 4, 2), (70_001, 16, 4), (17, 1, 2)])
def test_pack_and_apply_give_the_rank_ordered_sums(P, M, N):
    lib = _capi.load()
    lays, masks = _make_ranks(P, M, N, 0.1, seed=P + N)
    cap = int(max(int(m.sum()) for m in masks)) + 3
    words = int(lib.lrt_xchg_msg_words(P, M, cap, 1))
    counters = torch.zeros(2, dtype=torch.int32, device=DEV)
    msgs = torch.empty((N, words), dtype=torch.int32, device=DEV)
    par = 0
    for r in range(N):
        _capi.check(lib.lrt_xchg_pack(0, P, M, cap, *_dense(lays[r]), _capi.ptr(msgs[r]), _capi.ptr(counters), par, 1, _stream()), "lrt_xchg_pack")
        par ^= 1
    # expected: for every field, 0 + rows of rank 0 + rank 1 + ... in that order (a rank's rows are zero where it did not touch)
    want = {k: sum((lays[r].views[k].clone() for r in range(1, N)), lays[0].views[k].clone()) for k in lays[0].views}
    B = (P + 1023) // 1024
    for r in range(N):                                                     # every replica ends with the same bits
        lay = GradLayout(P, M, DEV); lay.flat.copy_(lays[r].flat)
        status = torch.zeros(1 + N, dtype=torch.int32, device=DEV)
        _capi.check(lib.lrt_xchg_apply(0, P, M, N, r, cap, _capi.ptr(msgs), C.c_longlong(words), *_dense(lay), _capi.ptr(status), 0, _stream()), "lrt_xchg_apply")
        st = status.cpu().tolist()
        assert st[0] == 0 and st[1:] == [int(m.sum()) for m in masks]
        for k in want:
            assert torch.equal(lay.views[k], want[k]), (r, k)
        # the header: block b's entries are its touched Gaussians in ascending order
        hdr = msgs[r].cpu().numpy()
        off, n, idx = hdr[:B], hdr[B:2 * B], hdr[2 * B:2 * B + cap]
        mk = masks[r].cpu().numpy()
        for b in (0, B // 2, B - 1):
            exp = np.nonzero(mk[b * 1024:(b + 1) * 1024])[0] + b * 1024
            assert n[b] == exp.size and np.array_equal(idx[off[b]:off[b] + n[b]], exp)
        # zero_only: clearing the rows of all lists leaves the buffer all-zero (what keeps a persistent buffer clean between steps)
        _capi.check(lib.lrt_xchg_apply(0, P, M, N, 0, cap, _capi.ptr(msgs), C.c_longlong(words), *_dense(lay), None, 1, _stream()), "lrt_xchg_apply")
        assert not bool(lay.flat.any())


def test_a_list_beyond_the_capacity_raises_the_device_flag_and_list_only_messages_work():
    lib = _capi.load()
    P, M, N = 9000, 16, 2
    lays, masks = _make_ranks(P, M, N, 0.2, seed=3)
    cap = int(masks[0].sum()) // 2                                         # too small on purpose
    words = int(lib.lrt_xchg_msg_words(P, M, cap, 1))
    counters = torch.zeros(2, dtype=torch.int32, device=DEV)
    msgs = torch.empty((N, words), dtype=torch.int32, device=DEV)
    for r in range(N):
        _capi.check(lib.lrt_xchg_pack(0, P, M, cap, *_dense(lays[r]), _capi.ptr(msgs[r]), _capi.ptr(counters), r & 1, 1, _stream()), "lrt_xchg_pack")
    lay = GradLayout(P, M, DEV); lay.flat.copy_(lays[0].flat)
    status = torch.zeros(1 + N, dtype=torch.int32, device=DEV)
    _capi.check(lib.lrt_xchg_apply(0, P, M, N, 0, cap, _capi.ptr(msgs), C.c_longlong(words), *_dense(lay), _capi.ptr(status), 0, _stream()), "lrt_xchg_apply")
    st = status.cpu().tolist()
    assert st[0] == 1 and st[1:] == [int(m.sum()) for m in masks]          # flagged, and the true lengths are reported for the next capacity
    # list-only message (what a single rank leaves behind to clear its rows by list in the next step)
    wl = int(lib.lrt_xchg_msg_words(P, M, P, 0))
    lst = torch.empty(wl, dtype=torch.int32, device=DEV)
    _capi.check(lib.lrt_xchg_pack(0, P, M, P, *_dense(lays[1]), _capi.ptr(lst), _capi.ptr(counters), 0, 0, _stream()), "lrt_xchg_pack")
    lay.flat.copy_(lays[1].flat)
    _capi.check(lib.lrt_xchg_apply(0, P, M, 1, 0, P, _capi.ptr(lst), C.c_longlong(wl), *_dense(lay), None, 1, _stream()), "lrt_xchg_apply")
    assert not bool(lay.flat.any())


def test_prezeroed_backward_writes_only_touched_rows_and_sharded_tracer_keeps_the_buffer_clean():
    """ShardedTracer on one rank (the bench's direct path): with the prezero protocol the library stores only the rows of Gaussians with a
    hit, the rows of the previous step are cleared by list -- results equal those of the zero-filling library over several steps."""
    from lidar_rt_amd import scenes
    from lidar_rt_amd.parallel import ShardedTracer
    sc, ro, rd = scenes.s10k()
    t = {k: torch.as_tensor(v, device=DEV) for k, v in sc.items()}
    ray_o, ray_d = torch.as_tensor(ro, device=DEV), torch.as_tensor(rd, device=DEV)
    bg = torch.as_tensor(scenes.BG_DEFAULT, device=DEV)
    H, W = ro.shape[:2]
    res = {}
    import os
    for pz in (False, True):
        os.environ["LRT_PREZERO"] = "force" if pz else "0"              # "force": the protocol without an exchange (what tools/slab_timing.py measures)
        tr = ShardedTracer()
        outs = []
        for step in range(3):
            dL = torch.as_tensor(scenes.upstream_grad(H, W, seed=step), device=DEV)
            mv = {k: (v + 0.01 * step if k == "means" else v) for k, v in t.items()}
            tr.forward(ray_o, ray_d, mv["means"], mv["scales"], mv["rotations"], mv["opacities"], mv["shs"], 3, bg)
            g = tr.backward(mv["means"], mv["scales"], mv["rotations"], mv["opacities"], mv["shs"], 3, bg, dL)
            outs.append({k: v.clone() for k, v in g.items()})
        res[pz] = outs
    os.environ.pop("LRT_PREZERO", None)
    for a, b in zip(res[False], res[True]):
        for k in a:
            assert float((a[k] - b[k]).abs().max()) <= 2e-6 * float(a[k].abs().max()), k   # runs that span two waves are summed with float atomics
            assert bool(((a[k] == 0) == (b[k] == 0)).all()), k                # untouched rows are exactly zero either way
