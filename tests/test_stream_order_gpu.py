"""The step is stream-ordered (include/lrt.h: "no call synchronises the host"; contrast DLT/trace_surfels.cpp:250-260, which ends
every call in cudaStreamSynchronize): build + forward + backward are enqueued behind a long-running kernel and return while the
GPU is still busy with it; the backward is then sized speculatively and decides on the device."""
import time

import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes
from lidar_rt_amd.diff_lidar_tracer import Tracer

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tests.hip_util import settings, rel_l2, DEV, DEFAULT_OPTS

GRADS = ("means", "scales", "rotations", "opacities", "shs")


_SLEEP_RATE = None


def _busy(seconds: float):
    """Keep the current stream busy for about `seconds` with ONE long-running kernel (many short ones would fill the launch
    queue and make the host wait for slots, which is not what these tests are about).  Nothing here waits for it."""
    global _SLEEP_RATE
    if _SLEEP_RATE is None:                                # calibrate torch's spin kernel: cycles per second
        torch.cuda._sleep(1_000_000); torch.cuda.synchronize()
        t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize()
        _SLEEP_RATE = 20_000_000 / max(time.perf_counter() - t0, 1e-6)
    torch.cuda._sleep(int(seconds * _SLEEP_RATE))


_SETTINGS = None


def _step(tr, t, ro, rd, dL):
    global _SETTINGS
    if _SETTINGS is None:        # made once: a host->device copy of pageable memory (the background colour) waits for the stream
        _SETTINGS = settings(scenes.BG_DEFAULT, 3)
    for v in t.values():
        v.grad = None
    tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
    out, acc = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
                  rotations=t["rotations"], tracer_settings=_SETTINGS)
    out.backward(dL)
    return out


def _setup(P=60_000, H=32, W=256, seed=4):
    sc = scenes.make_scene(P, seed=seed, radius_scale=0.4)
    o, d = scenes.kitti_rays(H, W)
    t = {k: torch.as_tensor(v, device=DEV).requires_grad_(True) for k, v in sc.items()}
    return sc, t, torch.as_tensor(o, device=DEV), torch.as_tensor(d, device=DEV), torch.as_tensor(scenes.upstream_grad(H, W), device=DEV)


def test_a_step_is_enqueued_behind_a_busy_gpu_without_waiting():
    sc, t, ro, rd, dL = _setup()
    tr = Tracer()
    for _ in range(3):                                   # sizes the workspaces, the first backward of an image size waits once
        _step(tr, t, ro, rd, dL)
    torch.cuda.synchronize()
    ref = {k: t[k].grad.clone() for k in GRADS}
    keep = _busy(0.4)
    marker = torch.cuda.Event(); marker.record()
    t0 = time.perf_counter()
    out = _step(tr, t, ro, rd, dL)
    host_s = time.perf_counter() - t0
    still_busy = not marker.query()
    assert still_busy, f"the GPU finished the dummy work before the step was enqueued ({host_s * 1e3:.1f} ms of host time)"
    assert host_s < 0.1, f"enqueueing one step took {host_s * 1e3:.1f} ms of host time while the GPU was busy: something waited"
    assert tr.optix_context.get_option("last_bwd_speculative", DEV) == 1
    torch.cuda.synchronize()
    tr.check(DEV)
    for k in GRADS:                                      # same inputs as the reference step: same gradients
        assert rel_l2(t[k].grad.cpu().numpy(), ref[k].cpu().numpy()) < 1e-6, k
    del keep, out


def test_speculated_size_too_small_falls_back_on_the_device():
    """Frame A composites few hits, frame B (same geometry, a quarter of the opacity) many more than 1.125 x A: a backward of B
    that is enqueued before A's... before B's count is known takes the re-tracing fallback, decided on the device."""
    sc, t, ro, rd, dL = _setup(P=40_000, H=16, W=256, seed=6)
    tr = Tracer()
    tr.optix_context.set_option("spec_margin", 0)
    for _ in range(2):
        _step(tr, t, ro, rd, dL)
    torch.cuda.synchronize()
    tB = {k: v.detach().clone().requires_grad_(True) for k, v in t.items()}
    with torch.no_grad():
        tB["opacities"].mul_(0.25)
    # reference for B on a fresh tracer (waits for the forward: exact sizes)
    tr2 = Tracer(); tr2.optix_context.set_option("spec_bwd", 0)
    _step(tr2, tB, ro, rd, dL)
    torch.cuda.synchronize()
    ref = {k: tB[k].grad.clone() for k in GRADS}
    keep = _busy(0.3)
    _step(tr, tB, ro, rd, dL)
    assert tr.optix_context.get_option("last_bwd_speculative", DEV) == 1
    torch.cuda.synchronize()
    for k in GRADS:                                      # re-trace + atomics vs sorted reduction: same hits, other summation order
        assert rel_l2(tB[k].grad.cpu().numpy(), ref[k].cpu().numpy()) < 1e-4, k
    # the next frames know B's count: sorted reduction again, same result
    _step(tr, tB, ro, rd, dL); torch.cuda.synchronize()
    for k in GRADS:
        assert rel_l2(tB[k].grad.cpu().numpy(), ref[k].cpu().numpy()) < 1e-5, k
    del keep


def test_overflow_in_a_run_ahead_loop_is_still_reported():
    """Errors are sticky: a host that never waits learns about an overflow at the next call that finds a finished forward."""
    from lidar_rt_amd._capi import LrtError
    sc, t, ro, rd, dL = _setup(P=20_000, H=16, W=128, seed=8)
    tr = Tracer()
    for _ in range(2):
        _step(tr, t, ro, rd, dL)
    torch.cuda.synchronize()
    tr.optix_context.set_option("c4_queue_limit", 136); tr.optix_context.set_option("c4_waves", 4)
    keep = _busy(0.2)
    _step(tr, t, ro, rd, dL)                              # overflows on the device; nobody looks yet
    tr.optix_context.set_option("c4_queue_limit", 1024); tr.optix_context.set_option("c4_waves", 0)
    torch.cuda.synchronize()
    with pytest.raises(LrtError, match="internal overflow"):
        _step(tr, t, ro, rd, dL)                          # the next forward reports it
    _step(tr, t, ro, rd, dL); torch.cuda.synchronize()    # reported once; the state recovers
    tr.check(DEV)
    del keep


def test_hip_graph_replay_gives_the_eager_results_bit_for_bit():
    """Option "graph": the launch sequence of every API call is recorded, fingerprinted and replayed from an instantiated HIP graph
    (one graph launch per call instead of one launch per kernel; the legacy default stream cannot be captured, so the step runs on a side
    stream).  With persistent parameter / gradient buffers (ShardedTracer, what bench.py --graph drives) the calls of later steps ARE
    replays; over steps with changing parameter values the results must equal the eager path's -- bit for bit in the image (hits are
    ordered by (t, gidx)), to float-atomic noise elsewhere -- and another P (new capacities, other launch arguments) must simply be
    another set of graphs."""
    from lidar_rt_amd.parallel import ShardedTracer
    sc, o, d = scenes.s10k()
    dL = scenes.upstream_grad(*o.shape[:2])
    ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
    g = torch.as_tensor(dL, device="cuda:0")
    bg = torch.as_tensor(scenes.BG_DEFAULT, device="cuda:0")
    side = torch.cuda.Stream()
    res = {}
    for graph in (0, 1):
        tr = ShardedTracer()
        st = tr.backend.state
        for k, v in {**DEFAULT_OPTS, "graph": graph}.items():
            st.set_option(k, v)
        outs = []
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for P_cut in (0, 100):                                         # two sizes: each has its own buffers and its own graphs
                t = {k: torch.as_tensor(np.ascontiguousarray(v[:v.shape[0] - P_cut]), device="cuda:0") for k, v in sc.items()}
                base = t["means"].clone()
                for step in range(10):
                    t["means"].copy_(base + 0.002 * (step % 5))            # in place: the same addresses every step
                    out, acc = tr.forward(ro, rd, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
                    gr = tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, g)
                    outs.append((out.clone(), acc.clone(), {k: v.clone() for k, v in gr.items() if k != "accum"}))
        side.synchronize()
        res[graph] = outs
        if graph:
            hits, caps = st.get_option("graph_hits", "cuda:0"), st.get_option("graph_captures", "cuda:0")
            # 60 calls.  A call is a replay only when EVERY launch argument repeats: the build's sequences have period 6 (three rotating
            # Morton boxes x two sort buffers), the forward's fresh output tensors alternate between addresses, the backward is sized exactly
            # or speculatively depending on whether the forward's status has arrived -- so a short test sees both first-sight captures and
            # replays (bench.py --graph: 14,939 replays against 10 instantiated graphs), which is what it should exercise
            assert hits >= 8 and 6 <= caps <= 56 and hits + caps >= 56, (hits, caps)
        st.set_option("graph", 0)
    for (oa, aa, ga), (ob, ab, gb) in zip(res[0], res[1]):
        assert torch.equal(oa, ob)
        assert float((aa - ab).abs().max()) <= 2e-6 * float(aa.abs().max())      # (float atomics of the forward: two trees, two arrival orders)
        for k in ga:
            assert float((ga[k] - gb[k]).abs().max()) <= 2e-6 * float(ga[k].abs().max()), k
