"""Azimuth-sector sharding of one LiDAR frame across the GPUs of a node
(not present in the reference, which is single-GPU).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI;
"gloo" on CPU for the tests).  Every rank holds the full Gaussian set; rank r
traces the contiguous column slab ``[r*W/N, (r+1)*W/N)`` of the (H, W) range
image.  A rank's LBVH holds only the Gaussians its slab can reach: the cone
around its rays (useful from 4 ranks on) and the wedge between the planes of
its edge columns (``lrt_build_for_slab``: what culls at 2 and 3 ranks).

* forward : local slab -> ONE ``all_gather`` of ``[status | (H, W/N, 9) slab]``
  (4.7 MB at 64x2048) so every rank sees the whole image for image-space losses
  -- and every rank's overflow status, see "errors" below.
* backward: each rank's partial gradients live in ONE flat fp32 buffer
  ``[d_means | d_scales | d_rotations | d_opacities | d_shs | accum]``
  (59 + 1 floats per Gaussian at M=16).  Three exchanges:
    - ``sparse`` (what bench.py and the sharded training step use): REPLICATED result.  Every rank packs the rows of the Gaussians
      it touched (``accum > 0``: list + pack on the device, fixed speculated capacity, no ``nonzero`` / ``tolist``) and ONE
      ``all_gather`` of ``[count | indices | rows]`` moves them to everybody; a rank then clears the rows it wrote itself and adds
      every rank's list in rank order, so all replicas hold bit-identical sums -- what a replicated optimizer (and the replicated
      densification decisions of the training loop) needs: parameters never diverge and no parameter synchronisation exists.
    - ``dense``  : one ``all_reduce`` of the flat buffer (replicated; a single large message so that RCCL can use all 7 xGMI links).
    - ``owner``  : reduce-scatter semantics.  Every Gaussian has ONE owning rank -- the rank whose slab axis is closest to the
      direction sensor -> Gaussian -- and a rank's rows travel to their owner only: one padded ``all_to_all``.  The gradient of
      Gaussian g is COMPLETE on ``owner[g]`` and meaningless elsewhere (``last_owner`` holds the map).  Azimuth sectors touch
      mostly their own Gaussians, so a rank sends a few per cent of what it touched -- but a training loop on top of it has to
      synchronise parameters AND Adam moments of the touched rows afterwards (3 x 59 floats per row against the 59 the gathering
      exchange moves once), which is why the training step uses ``sparse``.
  Capacities are speculated from the previous step; the true counts travel with the rows and are VERIFIED INSIDE THE STEP (one
  small device->host read after the collective): an overflow re-runs the exchange with an exact capacity before anything is
  added, on all ranks alike, so no step ever hands incomplete gradients to an optimizer.
* errors: the kernels raise device-side flags; a rank that raised alone would leave the others blocked in the next
  collective.  Every rank therefore sends its status word with its slab, keeps what it received in pinned memory, and ALL ranks
  raise the same ``LrtError`` at their next call into ``ShardedTracer`` (or ``check()``) -- one step late, consistently, without a
  host wait in the forward itself.

The local tracer is injected (``backend``): the product default drives the HIP
library; the CPU tests inject an oracle-backed stand-in to exercise the
collective logic with gloo (no GPU in CI).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import collections
import os
import torch
import torch.distributed as dist


def column_slab(W: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous column range of rank ``rank``; slabs differ by at most one column."""
    base, rem = divmod(W, world)
    a = rank * base + min(rank, rem)
    return a, a + base + (1 if rank < rem else 0)


class GradLayout:
    """Views into one flat fp32 buffer so the gradient exchange is a single collective."""
    FIELDS = (("means", 3), ("scales", 2), ("rotations", 4), ("opacities", 1))

    def __init__(self, P: int, M: int, device, with_accum: bool = True):
        self.P, self.M = P, M
        n = P * (3 + 2 + 4 + 1 + 3 * M) + (P if with_accum else 0)
        self.flat = torch.empty(n, dtype=torch.float32, device=device)    # every view is fully written by the backward
        self.views: Dict[str, torch.Tensor] = {}
        o = 0
        for name, k in self.FIELDS:
            self.views[name] = self.flat[o:o + P * k].view(P, k); o += P * k
        self.views["shs"] = self.flat[o:o + P * M * 3].view(P, M, 3); o += P * M * 3
        if with_accum:
            self.views["accum"] = self.flat[o:o + P]; o += P

    @property
    def width(self) -> int:
        return 11 + 3 * self.M


class HipBackend:
    """Local tracer on this rank's GPU (the product path)."""

    def __init__(self):
        from .diff_lidar_tracer import _C
        self._C = _C
        self.state = _C.OptiXStateWrapper("")

    def build(self, means, scales, rotations, opacities, mod=1.0, cull_rays=None):
        self._C.build_from_gaussians(self.state, means, scales, rotations, opacities, mod, cull_rays=cull_rays)

    def forward(self, ray_o, ray_d, means, scales, rotations, opacities, shs, deg, bg, mod=1.0):
        e = torch.empty(0, device=means.device)
        out, _, accum = self._C.trace_surfels(self.state, True, ray_o, ray_d, e, bg, means, shs, deg, e, opacities,
                                              scales, mod, rotations, e, e, e, e, False, False)
        self.last_serial = self.state.last_serial      # identifies the hit record this forward left in the library state
        return out, accum

    def backward(self, ray_o, ray_d, means, scales, rotations, opacities, shs, deg, bg, out, dL, mod=1.0, grads_out=None, forward_serial=None,
                 accum_out=None):
        e = torch.empty(0, device=means.device)
        g = self._C.trace_surfels_backward(self.state, ray_o, ray_d, e, bg, means, shs, deg, e, opacities, scales,
                                           mod, rotations, e, e, e, e, False, False, out, None, dL, grads_out=grads_out,
                                           forward_serial=forward_serial, accum_out=accum_out)
        return {"means": g[0], "shs": g[1], "opacities": g[3], "scales": g[4], "rotations": g[5]}

    def defer_errors(self, on: bool = True):
        """The library's forward / backward stop reporting overflows themselves: ShardedTracer gathers every rank's status and
        raises on all ranks alike (a rank raising alone would leave the others blocked in the next collective)."""
        self.state.set_option("defer_errors", 1 if on else 0)

    def status_to(self, dst: torch.Tensor):
        """This rank's error bits (last forward | sticky) as one float at dst[0] (device), stream-ordered."""
        import ctypes as C
        from . import _capi
        idx, h = self.state.handle(dst.device)
        with torch.cuda.device(idx):
            _capi.check(self.state._lib.lrt_status_to_device(h, _capi.ptr(dst), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "lrt_status_to_device")

    def clear_errors(self, device):
        """Consume the library's own (sticky) report of an overflow that ShardedTracer reports for all ranks."""
        from . import _capi
        try:
            self.state.check(device, wait=True)
        except _capi.LrtError:
            pass


_TAKES_CACHE: Dict[tuple, bool] = {}


class ShardedTracer:
    """Traces one frame sharded by azimuth sector over ``dist``'s world."""

    def __init__(self, backend=None, group=None, exchange: str = "sparse", world: Optional[int] = None, rank: Optional[int] = None,
                 deferred_accum: bool = False, deterministic: bool = False):
        """deferred_accum: the local forward leaves the hit weights `accum` all-zero and the local BACKWARD writes them (library option
        deferred_accum, lrt_backward_accum) -- straight into the flat exchange buffer when there is an exchange.  A training step reads the
        weights after the backward only (train.py:156,219); callers that need them from the forward keep the default.
        exchange: "owner" = rows of touched Gaussians go to their owning rank (complete gradient on the owner only); "dense" = one
        all_reduce of the flat gradient buffer (replicated); "sparse" = all_gather of the touched rows (replicated); "auto" = sparse
        unless the ranks together touched more than `sparse_max_fraction` of the Gaussians, then dense.
        world / rank: override what `torch.distributed` reports -- world=1 makes a plain single-rank tracer inside a multi-rank job
        (bench.py's `value_n1`: rank 0 times the unsharded step of the same workload beside the sharded one)."""
        if exchange not in ("auto", "dense", "sparse", "owner"):
            raise ValueError("exchange must be 'auto', 'dense', 'sparse' or 'owner'")
        self.exchange = exchange
        self.sparse_max_fraction = 0.6
        self.last_exchange = None                      # what the last backward used: "dense" | "sparse" | "owner" | None
        self.last_owner: Optional[torch.Tensor] = None  # (P,) int32 owner map of the last "owner" exchange
        self.backend = backend if backend is not None else HipBackend()
        self.deterministic = bool(deterministic) and isinstance(self.backend, HipBackend)      # library option deterministic (see Tracer): this rank's sums in a fixed order
        if self.deterministic:
            self.backend.state.set_option("deterministic", 1)
        self.deferred_accum = (bool(deferred_accum) or self.deterministic) and isinstance(self.backend, HipBackend)
        if self.deferred_accum:
            self.backend.state.set_option("deferred_accum", 1)
        self.group = group
        self.world = int(world) if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = int(rank) if rank is not None else (dist.get_rank(group) if (dist.is_initialized() and world is None) else 0)
        # build the LBVH for this rank's rays only (lrt_build_for_rays): Gaussians outside the cone around the slab's rays are left
        # out.  The library sizes the sort and the tree from the previous frame's kept count, so no read-back stalls the launch
        # queue.  Measured on S1M (tools/slab_timing.py, CULL=1, profiles/r03_summary.md): with the division-free cone test of round 3
        # the culled build is ahead from 4 ranks on (N=4: 0.265 -> 0.238 ms with a quarter of the Gaussians kept, N=8: 0.262 -> 0.182 ms
        # with an eighth); what is left is a chain of ~16 small launches.  A 180-degree slab (N=2) has no useful cone; 3 ranks: 120 degrees, none either.
        # round 4: (H, W, 3) slabs are culled by the wedge between their edge columns as well (lrt_build_for_slab), which works at 2 and 3 ranks
        # too -- and does not pay there (S1M, N=2: half the Gaussians kept, build 0.162 -> 0.227 ms: the culled path's cone kernel, compaction
        # and unfused sort cost more than the smaller tree saves; forward unchanged).  So the default stays: from 4 ranks on
        self.cull_build = self.world >= 4
        if os.environ.get("LRT_CULL_BUILD", "") in ("0", "1"):         # developer / test switch
            self.cull_build = os.environ["LRT_CULL_BUILD"] == "1"
        if self.world > 1 and hasattr(self.backend, "defer_errors"):
            self.backend.defer_errors(True)
        self.force_collectives = False # test hook: run the collective code paths with a world of one rank too (RCCL pre-flight)
        self._phase_on = False
        self._phase_ev = []            # (name, start event, end event) of the collectives' regions, read back by phase_timing()
        self._pending = []             # [(what, pinned host tensor, event or None)] status words received from all ranks, not yet looked at
        self._cap = None               # rows per message of the owner / gathering exchange (speculated from the previous step's counts)
        self._cap_key = None
        self.exchange_reruns = 0       # exchanges that had to run again with an exact capacity (statistics / tests)
        self.first_cap = None          # test hook: capacity of the first exchange of a size instead of the built-in guess
        self._bufs = {}                # reused message buffers of the exchanges, keyed by (what, shape)
        # device path of the gathering exchange (lrt_xchg_*): no host read inside the step.  The gradient buffer is kept ALL-ZERO between
        # steps (the rows of the previous step are cleared by list at the start of the next backward: `_prev_lists`), the library is told so
        # (option grads_prezeroed) and writes only the rows of Gaussians with a hit -- no 232 MB of zero rows per step and rank at 1 M
        self.prezero = os.environ.get("LRT_PREZERO", "1") != "0"
        self._prev_lists = None        # (messages tensor, N, cap, msg_words) whose rows make up everything non-zero in the flat buffer
        self._flat_dirty = True        # the flat buffer may hold anything: clear it whole before the next backward
        self._xchg_par = 0
        self._cap_hist = []            # list lengths of the last exchanges (capacity = 1.5 x their maximum + 4096)
        # load balance: azimuth sectors of equal width are not sectors of equal work (waymo4m shape, 4 ranks: 1.7 against 2.4 ms per rank).  Every
        # rank times its own build + forward + backward (two events), the times travel in the header of the slab all_gather, and every
        # `balance_every` steps all ranks move the slab edges towards equal time -- the same arithmetic on the same gathered numbers
        self.balance = isinstance(self.backend, HipBackend) and os.environ.get("LRT_BALANCE", "1") == "1"
        self.balance_every = int(os.environ.get("LRT_BALANCE_EVERY", "8"))
        self._edges, self._edges_key = None, None      # column edges [0, .., W] of the current split (None: equal widths)
        self._step_no = 0
        self._t_events, self._t_last_ms, self._times = None, 0.0, None
        self._cull_counts, self._cull_prev_key, self.cull_readbacks = collections.OrderedDict(), None, 0     # see _cull_sizing
        self._cull_exact_next = False  # the next culled build reads its count back (after error 8, see _drain)
        self._xchg_exact_next = False  # the next device-side exchange reads its list lengths back (after any reported problem)
        self._build_id = 0             # counts the builds of this tracer: a backward re-traces only in the structure its forward used (see backward)

    # ---- per-phase timing of the collective regions (bench.py --gpus N: build / forward / backward come from the library's HIP events)
    def enable_phase_timing(self, on: bool = True, every: int = 1):
        """HIP events around the collective regions; every = K: only every K-th call of a region is timed (an event record between two
        kernels costs ~5 us of pipeline on the stream)."""
        self._phase_on = bool(on); self._phase_ev = []; self._phase_every = max(1, int(every)); self._phase_calls = {}

    class _Region:
        def __init__(self, owner, name, device):
            self.o, self.name, self.cuda = owner, name, (device.type == "cuda" and owner._phase_on)
            if self.cuda and getattr(owner, "_phase_every", 1) > 1:
                c = owner._phase_calls.get(name, 0); owner._phase_calls[name] = c + 1
                self.cuda = (c % owner._phase_every) == 0
        def __enter__(self):
            if self.cuda:
                self.a = torch.cuda.Event(enable_timing=True); self.b = torch.cuda.Event(enable_timing=True); self.a.record()
        def __exit__(self, *exc):
            if self.cuda:
                self.b.record(); self.o._phase_ev.append((self.name, self.a, self.b))

    def phase_timing(self) -> Dict[str, float]:
        """Mean ms per call of every timed region since enable_phase_timing (synchronises)."""
        if not self._phase_ev:
            return {}
        torch.cuda.synchronize()
        acc: Dict[str, list] = {}
        for name, a, b in self._phase_ev:
            acc.setdefault(name, []).append(a.elapsed_time(b))
        self._phase_ev = []
        return {k: float(sum(v) / len(v)) for k, v in acc.items()}

    # ---- consistent error reporting -------------------------------------------------------------------------------------------
    def _remember(self, what: str, dev_values: torch.Tensor):
        """Keep status words that ALL ranks received identically; they are looked at by the next call (no wait now)."""
        if dev_values.is_cuda:
            host = torch.empty(dev_values.shape, dtype=dev_values.dtype, pin_memory=True)
            host.copy_(dev_values, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            self._pending.append((what, host, ev))
        else:
            self._pending.append((what, dev_values.clone(), None))

    def _drain(self, wait: bool):
        """Look at the status words whose copies have arrived (wait=True: all of them).  Returns the first problem as (what, per-rank words) or
        None; list lengths and compute times are absorbed on the way.  Identical on every rank: all ranks saw the same gathered words."""
        still = []
        bad = None
        for what, host, ev in self._pending:
            if ev is not None and not wait and not ev.query():
                still.append((what, host, ev)); continue
            if ev is not None:
                ev.synchronize()
            if what == "forward+times":                              # (N, 2): status word and compute time (ms) of an earlier step, identical on all ranks
                self._times = [float(v) for v in host[:, 1].tolist()]
                what, host = "forward", host[:, 0]
            vals = host.reshape(-1).tolist()
            if what == "exchange":                                   # [overflow flag, list length of every rank]: identical on all ranks
                self._cap_hist = (self._cap_hist + [max(int(v) for v in vals[1:])])[-8:]
                vals = vals[:1]
                if vals[0] != 0:
                    self.exchange_reruns += 1                        # (kept name) exchanges whose capacity was exceeded
                    self._flat_dirty = True
            if any(v != 0 for v in vals) and bad is None:
                bad = (what, vals)
        self._pending = still
        if bad is not None:
            self._xchg_exact_next = True                           # whatever went wrong: the next exchange reads its list lengths back (a step that is
            #                                                        redone must not inherit a capacity learnt from the incomplete attempt)
            if hasattr(self.backend, "clear_errors") and getattr(self, "_dev", None) is not None and self._dev.type == "cuda":
                self.backend.clear_errors(self._dev)
            if bad[0].startswith("forward") and any(int(v) & 8 for v in bad[1]):
                # a speculatively sized culled build lost primitives: the count remembered for that ray set was wrong; the next culled build
                # of this tracer reads its count back (exact size)
                self._cull_exact_next = True
                if self._cull_prev_key is not None:
                    self._cull_counts.pop(self._cull_prev_key, None)
        return bad

    _STATUS_HELP = ("forward status bits: 1 = candidate list, 2 = BVH queue, 4 = colour overflow list, "
                    "8 = the speculatively sized ray-culled build lost primitives (the next build is sized exactly); "
                    "'exchange' word 1 = a rank touched more Gaussians than the speculated message capacity (1.5 x the recent maximum): "
                    "that step's gradients are incomplete on every rank alike; the capacity has been raised")

    def check(self, wait: bool = True):
        """Raise on EVERY rank alike if any rank reported an overflow in a step whose status has arrived (wait=True: in all
        earlier steps).  Called at the start of forward()."""
        bad = self._drain(wait)
        if bad is not None:
            from ._capi import LrtError
            what, vals = bad
            raise LrtError(f"sharded tracer: a rank reported an overflow in an earlier step ({what}; per-rank words {vals}); that step's "
                           "results are incomplete on it.  " + self._STATUS_HELP)

    def verify_step(self):
        """The guard a training loop calls between ``loss.backward()`` and the optimizer step (train.py:215-220): WAIT for the status words of
        the step just enqueued -- every rank's forward / build bits and the exchange's overflow flag -- and return the first problem as
        ``(what, per-rank words)``, or None when the step's image and gradients are complete on every rank.  Nothing is raised and the report
        is consumed: the next build / exchange of this tracer is sized exactly (count read-back, capacity from the true list lengths), so a
        caller re-runs the step and calls verify_step() again; a second failure must raise BEFORE the parameters move (training_step does
        both).  Same answer on every rank (all ranks saw the same gathered words), so the ranks stay in step.  A single rank has nothing
        speculative to verify: None."""
        if self.world == 1 and not self.force_collectives and not self._pending:
            return None
        return self._drain(True)

    def check_replicas(self, tensors, what: str = "parameters"):
        """The replicated training design has NO parameter synchronisation: every rank must draw the same random numbers
        (``torch.manual_seed`` with one seed on all ranks before ``densify_and_split`` is ever reached) and must add the same
        gradient rows.  This is the cheap guard: a checksum of `tensors` (sum of the values and of their squares, float64) is
        all-reduced with MIN and MAX and every rank raises alike when the two differ.  One small collective + one host read; call
        it every few hundred steps (training_step does, ``opt.replica_check_interval``)."""
        if self.world == 1 and not self.force_collectives:
            return
        acc = []
        for t in tensors:
            x = t.detach().double().reshape(-1)
            acc += [x.sum(), (x * x).sum(), torch.tensor(float(x.numel()), dtype=torch.float64, device=x.device)]
        v = torch.stack(acc)
        lo, hi = v.clone(), v.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group); dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not bool(torch.equal(lo, hi)):
            raise RuntimeError(f"sharded tracer: the replicas' {what} have diverged (checksums differ between ranks): every rank must "
                               "be seeded identically (torch.manual_seed) and run the same optimizer / densification code")

    @staticmethod
    def _backend_takes(fn, name: str) -> bool:
        """Capability of an injected backend, from its signature (a TypeError raised INSIDE the backend must not be mistaken for
        a missing keyword)."""
        import inspect
        key = (getattr(fn, "__func__", fn), name)                  # looked up three times per step: inspect.signature costs 35 us a call
        hit = _TAKES_CACHE.get(key)
        if hit is not None:
            return hit
        try:
            ps = inspect.signature(fn).parameters
            res = name in ps or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in ps.values())
        except (TypeError, ValueError):
            res = False
        _TAKES_CACHE[key] = res
        return res

    # ---- slab edges ----------------------------------------------------------------------------------------------------------------
    def slab_of(self, W: int, rank: int) -> Tuple[int, int]:
        if self._edges is not None and self._edges_key == (W, self.world):
            return self._edges[rank], self._edges[rank + 1]
        return column_slab(W, rank, self.world)

    def _rebalance(self, W: int):
        """Move the slab edges towards equal compute time per rank.  Rank r processed its `w_r` columns in `t_r` ms, i.e. at w_r / t_r columns
        per ms; widths proportional to those rates would equalise the times if the cost per column were uniform inside a slab, so the step
        is damped (half way) and widths stay multiples of the 8-column tile.  Deterministic: identical inputs on every rank."""
        t = self._times
        if not t or len(t) != self.world or min(t) <= 0.0:
            return
        if max(t) * len(t) < 1.05 * sum(t):                    # within 5 % of the mean: leave the edges alone (every move costs the ranks whose
            return                                             # slab changes a culled build whose size has to be found again)
        if self._edges is None or self._edges_key != (W, self.world):
            self._edges = [column_slab(W, r, self.world)[0] for r in range(self.world)] + [W]; self._edges_key = (W, self.world)
        w = [self._edges[r + 1] - self._edges[r] for r in range(self.world)]
        rate = [w[r] / t[r] for r in range(self.world)]
        tot = sum(rate)
        want = [0.5 * w[r] + 0.5 * W * rate[r] / tot for r in range(self.world)]
        edges, acc = [0], 0.0
        for r in range(self.world - 1):
            acc += want[r]
            e = int(round(acc / 8.0)) * 8
            e = max(e, edges[-1] + 8); e = min(e, W - 8 * (self.world - 1 - r))
            edges.append(e)
        edges.append(W)
        if all(edges[i + 1] > edges[i] for i in range(self.world)):
            self._edges = edges

    # ---- forward ----------------------------------------------------------------------------------------------------------------
    def _cull_sizing(self, key):
        """How the next ray-culled build is sized (library option cull_next).  The library can only guess from the PREVIOUS build's count,
        which is right while consecutive builds see the same rays (a benchmark, a static sensor) and wrong when training draws frames at
        random from a drive: the count of a sector can change several-fold from one pose to the next, a too small size loses primitives (a loud
        error, code 8, but the optimizer step has happened by then).  So the counts are remembered PER RAY SET: a set seen before is sized
        speculatively from its own last count (x 1.25 + 4096, no host wait), a new one reads its count back (one 8-byte copy: the host waits for
        the build's first kernels once).  `key` identifies the ray set: the caller's cull_key (renderer: the frame index) plus the slab and P.
        A ray set WITHOUT a cull_key has no identity (round 4 used the tensors' addresses: the caching allocator hands the same address to the
        next frame's rays, so different ray sets collided on one key -- ADVICE r04): its count is read back on every build.  A loop over the
        same rays (bench.py, tools/slab_timing.py) passes a cull_key."""
        st = getattr(self.backend, "state", None)
        if st is None or not hasattr(st, "get_option"):
            return
        if self._cull_prev_key is not None:
            n = st.get_option("cull_last", self._dev)          # copied behind that build's first kernels: long on the host
            if n >= 0:
                self._cull_counts[self._cull_prev_key] = n
                self._cull_counts.move_to_end(self._cull_prev_key)
                while len(self._cull_counts) > 16384:
                    self._cull_counts.popitem(last=False)
        g = self._cull_counts.get(key)
        if self._cull_exact_next:                              # a speculative size just lost primitives: this build reads its count back
            g, self._cull_exact_next = None, False
        elif g is None:
            # the same rays and P with other slab edges (the balancer moved them by a few tiles): the kept count scales about like the width
            if isinstance(key, tuple) and len(key) == 4 and key[0] is not None:
                ident, a, b, P_ = key
                for seen, (k2, n2) in enumerate(reversed(self._cull_counts.items())):
                    if seen >= 256:
                        break
                    if isinstance(k2, tuple) and len(k2) == 4 and k2[0] == ident and k2[3] == P_ and \
                            min(b, k2[2]) - max(a, k2[1]) >= 0.75 * max(b - a, k2[2] - k2[1]):
                        g = int(n2 * (b - a) / max(k2[2] - k2[1], 1) * 1.15); break
        if isinstance(key, tuple) and len(key) == 4 and key[0] is None:
            g = None                                           # a ray set without a name: nothing is known about it, its count is read back
        st.set_option("cull_next", 0 if g is None else g + g // 4 + 4096)
        self._cull_prev_key = key if not (isinstance(key, tuple) and len(key) == 4 and key[0] is None) else None
        self.cull_readbacks += 1 if g is None else 0

    def forward(self, ray_o, ray_d, means, scales, rotations, opacities, shs, deg, bg, mod=1.0, rebuild=True, cull_key=None):
        H, W = ray_o.shape[:2]
        self._dev = means.device
        if self.world > 1 or self.force_collectives:
            self.check(wait=True)                              # statuses and counts of the previous step: long there; all ranks agree
        timed = self.balance and self.world > 1 and means.is_cuda and W >= 16 * self.world
        if timed:
            if self._t_events is not None and self._t_events[3].query():
                e = self._t_events                             # the rank's OWN work: build + forward, and the local backward -- not the slab
                self._t_last_ms = float(e[0].elapsed_time(e[1]) + e[2].elapsed_time(e[3]))     # all_gather between them, which waits for the slowest rank
            self._step_no += 1
            if self._step_no % self.balance_every == 0:
                self._rebalance(W)
            # the rank's time is MEASURED on the step behind a rebalance only (and until a first measurement exists): four event records on
            # the launch stream cost ~20 us, 5 % of a rank's 0.36 ms step at N=8; the last measured time travels with every slab message
            measure = (self._step_no % self.balance_every == 0) or self._t_events is None
            if measure:
                ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
        a, b = self.slab_of(W, self.rank)
        self._slab = (a, b)
        self._ro = ray_o[:, a:b].contiguous(); self._rd = ray_d[:, a:b].contiguous()
        self._rays_full = (ray_o, ray_d)
        cull = (self._ro, self._rd) if self.cull_build else None
        if cull is not None:
            self._cull_sizing((cull_key, a, b, int(means.shape[0])))
        if rebuild or cull is not None:                        # a ray-culled structure must be rebuilt for every ray set
            self._build_id += 1
            if self._backend_takes(self.backend.build, "cull_rays"):
                self.backend.build(means, scales, rotations, opacities, mod, cull_rays=cull)
            else:                                                                 # backend without culling (test stand-ins)
                self.backend.build(means, scales, rotations, opacities, mod)
        if hasattr(self.backend, "state"):
            # the library keeps what it learns per tile (first-slab widths, tile lengths -> queue boundaries) per NAMED ray set (option ray_set, 256 sets):
            # a training loop draws its frames at random, and a table learnt on another sensor pose costs the forward 15 %
            rs = -1 if cull_key is None else (cull_key if isinstance(cull_key, int) and 0 <= cull_key < 2 ** 30 else (hash(cull_key) & 0x3fffffff))
            if rs != getattr(self, "_ray_set", None):
                self.backend.state.set_option("ray_set", rs); self._ray_set = rs
        out_loc, accum_loc = self.backend.forward(self._ro, self._rd, means, scales, rotations, opacities, shs,
                                                  deg, bg, mod)
        self._out_loc, self._accum_loc = out_loc, accum_loc
        if timed and measure:
            evf = torch.cuda.Event(enable_timing=True); evf.record()
        # what backward() needs of THIS forward: an autograd Function keeps it in its ctx (renderer._ShardedTrace), so that a second
        # forward before loss.backward() -- an evaluation render, another frame -- cannot make the backward differentiate the wrong slab
        self.last_ctx = {"slab": (a, b), "ro": self._ro, "rd": self._rd, "out_loc": out_loc, "accum_loc": accum_loc,
                         "serial": getattr(self.backend, "last_serial", None), "build_id": self._build_id, "mod": mod}
        if self.world == 1 and not self.force_collectives:
            return out_loc, accum_loc
        # all_gather needs equal shapes: slabs padded to the widest one; element 0 of the message = this rank's status word
        self._timed_ev0 = (ev0, evf) if (timed and measure) else None
        with self._Region(self, "slab_all_gather", out_loc.device):
            wmax = max(self.slab_of(W, r)[1] - self.slab_of(W, r)[0] for r in range(self.world))
            msg = torch.zeros(2 + H * wmax * 9, dtype=out_loc.dtype, device=out_loc.device)
            if timed:
                msg[1] = self._t_last_ms                     # word 1: this rank's compute time (ms) of its last timed step
            msg[2:].view(H, wmax, 9)[:, :b - a] = out_loc
            if hasattr(self.backend, "status_to") and out_loc.is_cuda:
                self.backend.status_to(msg)                  # word 0: this rank's forward / build status bits
            parts = self._all_gather_rows(msg)               # one flat receive buffer with RCCL
            cols = []
            for r in range(self.world):
                ra, rb = self.slab_of(W, r)
                cols.append(parts[r][2:].view(H, wmax, 9)[:, :rb - ra])
            full = torch.cat(cols, dim=1)
            hdr = torch.stack([p[:2] for p in parts])        # (N, 2), identical on every rank
            # status words and (when the slabs are balanced) compute times in ONE copy to pinned memory: every copy + event pair on the
            # launch stream costs ~9 us of a rank's 0.36 ms step
            self._remember("forward+times" if timed else "forward", hdr if timed else hdr[:, 0])
        return full, accum_loc                          # accum is completed by backward()'s exchange

    # ---- backward ---------------------------------------------------------------------------------------------------------------
    def backward(self, means, scales, rotations, opacities, shs, deg, bg, dL_full, mod=1.0,
                 reduce: bool = True, fwd_ctx=None) -> Dict[str, torch.Tensor]:
        """fwd_ctx: the `last_ctx` of the forward to differentiate (default: the most recent one).  When another forward has
        replaced the library's hit record since, the local backward re-traces (forward_serial mismatch) -- and when that forward also
        REBUILT the structure (another frame's actor poses, another ray set's culled build, a rebalanced slab: ADVICE r04), the structure
        of THIS forward is built again first: culled for its rays, sized exactly.  Same gradients."""
        fc = fwd_ctx if fwd_ctx is not None else self.last_ctx
        if fc.get("build_id", self._build_id) != self._build_id:
            cull = (fc["ro"], fc["rd"]) if self.cull_build else None
            st_ = getattr(self.backend, "state", None)
            if cull is not None and st_ is not None and hasattr(st_, "set_option"):
                st_.set_option("cull_next", 0)                 # read the count back: nothing is known about the size any more
                self._cull_prev_key = None
            self._build_id += 1
            if self._backend_takes(self.backend.build, "cull_rays"):
                self.backend.build(means, scales, rotations, opacities, fc.get("mod", mod), cull_rays=cull)
            else:
                self.backend.build(means, scales, rotations, opacities, fc.get("mod", mod))
            fc = dict(fc); fc["build_id"] = self._build_id; fc["serial"] = -1      # the hit record is not this forward's: re-trace
        a, b = fc["slab"]
        ro_, rd_, out_loc_, accum_loc_ = fc["ro"], fc["rd"], fc["out_loc"], fc["accum_loc"]
        dL = dL_full[:, a:b].contiguous()
        if getattr(self, "_timed_ev0", None) is not None:
            evb0 = torch.cuda.Event(enable_timing=True); evb0.record()
        P = means.shape[0]; M = shs.shape[1]
        lay = getattr(self, "_layout", None)
        if lay is None or lay.P != P or lay.M != M or lay.flat.device != means.device:
            lay = self._layout = GradLayout(P, M, means.device)           # reused across steps: no per-step 240 MB allocation
            self._prev_lists, self._flat_dirty = None, True
        direct = {k: lay.views[k] for k in ("means", "scales", "rotations", "opacities", "shs")}
        exchanging = reduce and (self.world > 1 or self.force_collectives)
        # measured on S1M (profiles/r04_summary.md): a single rank gains nothing (zero rows inside k_bk_sort 37 us against list + clear-by-list
        # 25 us plus two launches), a rank of an 8-way split saves the 45 us its dense zero rows cost -- so the protocol runs with the exchange only
        pz = (self.prezero and lay.flat.is_cuda and hasattr(self.backend, "state") and exchanging and self.exchange == "sparse")
        if os.environ.get("LRT_PREZERO", "") == "force":                  # tools/slab_timing.py: one process plays a rank of an N-way split
            pz = lay.flat.is_cuda and hasattr(self.backend, "state") and (not exchanging or self.exchange == "sparse")
        if hasattr(self.backend, "state") and getattr(self, "_pz_set", None) != pz:
            self.backend.state.set_option("grads_prezeroed", 1 if pz else 0); self._pz_set = pz
            self._flat_dirty, self._prev_lists = True, None                   # the protocol changed hands: nothing is known about the buffer
        if not pz:
            # a backward outside the prezero protocol writes every row of the flat buffer: the lists of an earlier prezero step no longer
            # describe what is non-zero in it (stale rows survived in the returned views)
            self._flat_dirty, self._prev_lists = True, None
        if pz:
            if self._flat_dirty or self._prev_lists is None:
                lay.flat.zero_()                                           # first step / after an error / after a dense exchange
            else:
                msgs, n_l, cap_l, words_l = self._prev_lists
                self._xchg_apply(lay, msgs, n_l, 0, cap_l, words_l, None, zero_only=True)
            self._flat_dirty = True                                        # until this step has left its lists behind
        if self._backend_takes(self.backend.backward, "grads_out"):
            kw = {"forward_serial": fc.get("serial")} if self._backend_takes(self.backend.backward, "forward_serial") else {}
            if self.deferred_accum:
                # the backward completes the weights: into the forward's own (all-zero) tensor, or -- with an exchange to follow -- straight into
                # the flat buffer's accum view (no 4 P-byte copy; under the prezero protocol that view is all-zero on entry like the rest)
                kw["accum_out"] = lay.views["accum"] if exchanging else accum_loc_
            self.backend.backward(ro_, rd_, means, scales, rotations, opacities, shs, deg, bg,
                                  out_loc_, dL, mod, grads_out=direct, **kw)     # kernels write straight into the flat buffer
        else:                                                                     # backend without grads_out (test stand-ins)
            g = self.backend.backward(ro_, rd_, means, scales, rotations, opacities, shs, deg, bg, out_loc_, dL, mod)
            for k in direct:
                direct[k].copy_(g[k].view_as(direct[k]))
        if getattr(self, "_timed_ev0", None) is not None:          # this rank's own compute of the step ends here (the exchange is not the rank's work)
            ev1 = torch.cuda.Event(enable_timing=True); ev1.record()
            self._t_events = (self._timed_ev0[0], self._timed_ev0[1], evb0, ev1); self._timed_ev0 = None
        self.last_exchange = None
        if not exchanging:
            if pz:                                             # leave this step's own list behind for the next step's clearing
                cap = self._pz_cap(lay)
                words = self._xchg_words(lay, cap, False)
                msg = self._buf("pz_list", (words,), torch.int32, lay.flat.device)
                self._xchg_pack(lay, msg, cap, accum_loc_, with_rows=False)
                self._prev_lists, self._flat_dirty = (msg, 1, cap, words), False
            return {**lay.views, "accum": accum_loc_}          # nothing to exchange: the forward's own accum tensor, no 4 P-byte copy
        if not self.deferred_accum:
            lay.views["accum"].copy_(accum_loc_)
        if reduce and (self.world > 1 or self.force_collectives):
            with self._Region(self, "gradient_exchange", lay.flat.device):
                mode = self.exchange
                if mode == "owner":
                    self._exchange_owner(lay, means)
                elif mode == "sparse" and lay.flat.is_cuda and hasattr(self.backend, "state"):
                    self._exchange_lists(lay, pz)
                elif mode == "dense" or not self._exchange_sparse(lay):
                    dist.all_reduce(lay.flat, op=dist.ReduceOp.SUM, group=self.group)
                    self.last_exchange = "dense"
        return lay.views

    def _all_gather_rows(self, t: torch.Tensor):
        """all_gather of equally shaped tensors; one flat receive buffer when the backend offers it (RCCL), so that the
        collective is a single large message without a per-rank copy."""
        world = self.world
        if t.is_cuda and dist.get_backend(self.group) == "nccl":
            out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out.view(-1, *t.shape[1:]) if t.dim() > 0 else out, t.contiguous(), group=self.group)
            return [out[r] for r in range(world)]
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous(), group=self.group)
        return parts

    # ---- owner-based exchange ---------------------------------------------------------------------------------------------------
    def slab_axes(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(origin (3,), axes (N,3)): the frame's sensor origin (mean ray origin) and the unit mean ray direction of every rank's
        slab -- every rank computes all of them from the full ray grid, on the device."""
        ray_o, ray_d = self._rays_full
        W = ray_o.shape[1]
        origin = ray_o.reshape(-1, 3).mean(0)
        axes = []
        for r in range(self.world):
            ra, rb = self.slab_of(W, r)
            d = ray_d[:, ra:rb].reshape(-1, 3)
            m = (d / d.norm(dim=1, keepdim=True).clamp_min(1e-30)).mean(0)
            axes.append(m / m.norm().clamp_min(1e-30))
        return origin.contiguous(), torch.stack(axes, 0).contiguous()

    def owner_map(self, means: torch.Tensor) -> torch.Tensor:
        """owner[g] (int32): the rank whose slab axis is closest to the direction sensor -> Gaussian g (ties: the lowest rank)."""
        origin, axes = self.slab_axes()
        P = means.shape[0]
        return ((means.detach() - origin) @ axes.T).argmax(1).to(torch.int32)     # torch's argmax also returns the first maximum

    # ---- helpers of the exchanges ------------------------------------------------------------------------------------------------
    def _buf(self, what: str, shape, dtype, dev) -> torch.Tensor:
        """A message buffer that lives across steps (no per-step allocation of the padded lists)."""
        key = (what, tuple(shape), dtype, str(dev))
        t = self._bufs.get(key)
        if t is None:
            for k in [k for k in self._bufs if k[0] == what]:
                del self._bufs[k]
            t = self._bufs[key] = torch.empty(shape, dtype=dtype, device=dev)
        return t

    @staticmethod
    def _to_host(t: torch.Tensor) -> list:
        """Values of a small device tensor: the ONE host wait of an exchange (the counts that verify its capacity)."""
        if t.is_cuda:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()
            return h.reshape(-1).tolist()
        return t.reshape(-1).tolist()

    def _start_cap(self, key, P: int, guess: int) -> int:
        if self._cap_key != key:
            self._cap, self._cap_key = None, key
        if self._cap is None and self.first_cap is not None:
            guess = int(self.first_cap)
        return min(max(self._cap if self._cap is not None else guess, 1), max(P, 1))

    def _adapt_cap(self, biggest: int):
        want = biggest + biggest // 4 + 1024
        if self._cap is None or want > self._cap or want < self._cap // 2:
            self._cap = want

    @staticmethod
    def _row_fields(M: int):
        return [(name, k) for name, k in GradLayout.FIELDS] + [("shs", 3 * M), ("accum", 1)]

    def _exchange_owner(self, lay: GradLayout, means: torch.Tensor):
        """Rows of touched Gaussians go to their owner: ONE all_to_all of ``N x [count | cap indices | cap rows]`` per rank.
        The capacity per (source, owner) pair is speculated from the previous exchange's largest count (x1.25 + 1024); the true
        counts travel with the rows; their maximum over all pairs is all-reduced and read by the host BEFORE anything is added: a
        count beyond the capacity re-runs pack + all_to_all with an exact capacity, on all ranks alike."""
        P, M, N, rank = lay.P, lay.M, self.world, self.rank
        width = lay.width
        dev = lay.flat.device
        owner = self.owner_map(means)
        self.last_owner = owner
        cap = self._start_cap(("owner", P, M, N), P, P // (4 * N) + 1024)
        v = lay.views
        # (round 6: the device helpers of this non-default exchange left the library; torch's index ops move the rows)
        while True:
            blk = 1 + cap + cap * width                                   # int32 words per (source, owner) block: count | indices | rows
            send = self._buf("own_send", (N, blk), torch.int32, dev); recv = self._buf("own_recv", (N, blk), torch.int32, dev)
            cnt = send[:, 0]                                              # written in place: the message needs no assembly
            idx = send[:, 1:1 + cap]
            rows = send[:, 1 + cap:].view(torch.float32).view(N, cap, width)
            touched = (v["accum"] > 0)
            cnt.zero_()
            for d in range(N):
                if d == rank:
                    continue
                g = torch.nonzero(touched & (owner == d)).squeeze(1)
                cnt[d] = g.numel()
                g = g[:cap]
                idx[d, :g.numel()] = g.to(torch.int32)
                col = 0
                for name, k in self._row_fields(M):
                    rows[d, :g.numel(), col:col + k] = v[name].reshape(P, k).index_select(0, g)
                    col += k
            dist.all_to_all_single(recv, send, group=self.group)
            rcnt = recv[:, 0].contiguous()
            # in-step verification: no pair may have overflowed anywhere -- every rank learns the global maximum
            worst = torch.maximum(cnt.max(), rcnt.max()).reshape(1).contiguous()
            dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=self.group)
            biggest = int(self._to_host(worst)[0])
            if biggest <= cap:
                break
            self.exchange_reruns += 1
            cap = min(biggest + biggest // 4 + 1024, max(P, 1))           # the dense tensors are untouched so far: just run again
        ridx = recv[:, 1:1 + cap]
        rrows = recv[:, 1 + cap:].view(torch.float32).view(N, cap, width)
        for s_ in range(N):                                               # fixed order of the sources -> a deterministic sum
            if s_ == rank:
                continue
            c = min(int(rcnt[s_]), cap)
            g = ridx[s_, :c].long()
            col = 0
            for name, k in self._row_fields(M):
                v[name].view(P, k).index_add_(0, g, rrows[s_, :c, col:col + k])
                col += k
        self._adapt_cap(biggest)
        self.last_exchange = "owner"

    # ---- replicated gathering exchange, device path (lrt_xchg_*): no host read, one launch per side ---------------------------------
    @staticmethod
    def _xchg_words(lay: GradLayout, cap: int, with_rows: bool) -> int:
        from . import _capi
        return int(_capi.load().lrt_xchg_msg_words(lay.P, lay.M, int(cap), 1 if with_rows else 0))

    def _pz_cap(self, lay: GradLayout) -> int:
        """Capacity of a LIST-ONLY message (4 bytes per entry): every Gaussian fits, so this list can never overflow."""
        return max(lay.P, 1)

    def _xchg_counters(self, dev) -> torch.Tensor:
        c = self._bufs.get(("xchg_counters", str(dev)))
        if c is None:
            c = self._bufs[("xchg_counters", str(dev))] = torch.zeros(2, dtype=torch.int32, device=dev)
            self._xchg_par = 0
        return c

    def _xchg_pack(self, lay: GradLayout, msg: torch.Tensor, cap: int, accum: torch.Tensor, with_rows: bool):
        import ctypes as C
        from . import _capi
        v, p = lay.views, _capi.ptr
        dev = lay.flat.device
        di = dev.index if dev.index is not None else torch.cuda.current_device()
        cnt = self._xchg_counters(dev)
        with torch.cuda.device(di):
            _capi.check(_capi.load().lrt_xchg_pack(di, lay.P, lay.M, int(cap), p(v["means"]), p(v["scales"]), p(v["rotations"]), p(v["opacities"]), p(v["shs"]),
                                                   p(accum), p(msg), p(cnt), self._xchg_par, 1 if with_rows else 0,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lrt_xchg_pack")
        self._xchg_par ^= 1

    def _xchg_apply(self, lay: GradLayout, msgs: torch.Tensor, n_lists: int, rank: int, cap: int, words: int, status, zero_only: bool):
        import ctypes as C
        from . import _capi
        v, p = lay.views, _capi.ptr
        dev = lay.flat.device
        di = dev.index if dev.index is not None else torch.cuda.current_device()
        with torch.cuda.device(di):
            _capi.check(_capi.load().lrt_xchg_apply(di, lay.P, lay.M, int(n_lists), int(rank), int(cap), p(msgs), C.c_longlong(int(words)), p(v["means"]), p(v["scales"]),
                                                    p(v["rotations"]), p(v["opacities"]), p(v["shs"]), p(v["accum"]), p(status) if status is not None else None,
                                                    1 if zero_only else 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lrt_xchg_apply")

    def _exchange_lists(self, lay: GradLayout, pz: bool):
        """Replicated sums of the ranks' partial gradients, on the device from end to end: ONE launch lists and packs this rank's touched
        rows into ``[off[B] | n[B] | idx[cap] | rows[cap][60]]`` (blocks of 1024 Gaussian indices), ONE all_gather moves the messages,
        ONE launch clears this rank's own rows and adds every rank's list in rank order, block by block (a block of indices is owned by
        one workgroup: bit-identical sums on all replicas).  Nothing is read back inside the step: the capacity is speculated (1.5 x the
        largest list of the last 8 exchanges + 4096; the first exchange of a size reads its counts back once), the apply kernel raises a
        device flag when a list did not fit (it skips the affected blocks), the flag and the list lengths travel to pinned memory
        asynchronously; a training loop reads them with verify_step() BEFORE its optimizer step and re-runs the step, any other caller gets
        the NEXT forward's check(), which raises on all ranks alike (every rank saw the same messages); either way the capacity is raised."""
        P, M, N, rank = lay.P, lay.M, self.world, self.rank
        dev = lay.flat.device
        key = ("lists", P, M, N)
        if self._cap_key != key:
            self._cap_key, self._cap_hist = key, []
        if self.first_cap is not None and not self._cap_hist and not self._xchg_exact_next:
            cap = int(self.first_cap)
        elif self._cap_hist and not self._xchg_exact_next:
            cap = max(self._cap_hist) + max(self._cap_hist) // 2 + 4096
        else:
            self._xchg_exact_next = False
            # the first exchange of a size: nothing is known about the list lengths (round 4 guessed P / 4 and overflowed deterministically
            # with 2-3 ranks or wide sectors: ADVICE r04).  ONE count read-back sizes it exactly -- the largest list over the ranks, so that
            # every rank allocates the same message -- and later steps speculate from the measured lengths
            n_own = (lay.views["accum"] > 0).sum().reshape(1).to(torch.int64)
            dist.all_reduce(n_own, op=dist.ReduceOp.MAX, group=self.group)
            biggest = int(self._to_host(n_own)[0])
            cap = biggest + biggest // 2 + 4096
            self.exchange_readbacks = getattr(self, "exchange_readbacks", 0) + 1
        cap = max(1, min(cap, max(P, 1)))
        words = self._xchg_words(lay, cap, True)
        msg = self._buf("xl_send", (words,), torch.int32, dev)
        self._xchg_pack(lay, msg, cap, lay.views["accum"], with_rows=True)
        if dist.get_backend(self.group) == "nccl":
            recv = self._buf("xl_recv", (N * words,), torch.int32, dev)
            dist.all_gather_into_tensor(recv, msg, group=self.group)
        else:
            parts = [torch.empty_like(msg) for _ in range(N)]
            dist.all_gather(parts, msg, group=self.group)
            recv = self._buf("xl_recv", (N * words,), torch.int32, dev)
            recv.view(N, words).copy_(torch.stack(parts))
        status = self._buf("xl_status", (1 + N,), torch.int32, dev)
        status.zero_()
        self._xchg_apply(lay, recv, N, rank, cap, words, status, zero_only=False)
        self._remember("exchange", status)
        if pz:
            self._prev_lists, self._flat_dirty = (recv, N, cap, words), False
        self.last_exchange = "sparse"

    # ---- replicated gathering exchange -------------------------------------------------------------------------------------------
    def _exchange_sparse(self, lay: GradLayout) -> bool:
        """Sum the ranks' partial gradients by exchanging only the rows of touched Gaussians, replicated on every rank.

        A Gaussian has a non-zero partial gradient on a rank only if one of that rank's rays composited it, and every
        composited hit adds a weight > 0 to `accum` (forward.cu:268), so `accum > 0` is the exact mask.  Each rank packs
        ``[count | indices | 60-float rows]`` of its touched Gaussians into one message of a speculated capacity (device-side
        listing, no ``nonzero``), ONE all_gather moves the messages, the host reads the N counts (the one wait of the step: an
        overflow re-runs with an exact capacity before anything is added), then every rank clears the rows it wrote itself --
        the rest of the buffers is zero already -- and adds all lists in rank order: indices are unique inside a list, so every
        rank ends with bit-identical sums.  Returns False (nothing exchanged) when ``exchange == "auto"`` and the ranks
        together touched more than ``sparse_max_fraction`` of the Gaussians (the dense all_reduce then moves fewer bytes)."""
        P, M, N, rank = lay.P, lay.M, self.world, self.rank
        width = lay.width
        dev = lay.flat.device
        v = lay.views
        # (round 6: the device helpers of this round-3 exchange left the library; the training exchange is _exchange_lists)
        cap = self._start_cap(("sparse", P, M, N), P, P // (2 * N) + 1024)
        while True:
            blk = 1 + cap + cap * width
            msg = self._buf("sp_send", (blk,), torch.int32, dev)
            cnt, idx, rows = msg[0:1], msg[1:1 + cap], msg[1 + cap:].view(torch.float32).view(cap, width)
            g = torch.nonzero(v["accum"] > 0).squeeze(1)
            cnt[0] = g.numel()
            g = g[:cap]
            idx[:g.numel()] = g.to(torch.int32)
            col = 0
            for name, k in self._row_fields(M):
                rows[:g.numel(), col:col + k] = v[name].reshape(P, k).index_select(0, g)
                col += k
            parts = self._all_gather_rows(msg)
            counts = [int(c) for c in self._to_host(torch.stack([q[0] for q in parts]))]
            biggest = max(counts)
            if biggest <= cap:
                break
            self.exchange_reruns += 1
            cap = min(biggest + biggest // 4 + 1024, max(P, 1))
        if self.exchange == "auto" and sum(counts) > self.sparse_max_fraction * P:
            self._adapt_cap(biggest)
            return False                                                  # nothing was modified: the caller all-reduces the flat buffer
        # clear what this rank wrote (its own touched rows), then add every rank's list in rank order
        g = idx[:counts[rank]].long()
        for name, k in self._row_fields(M):
            v[name].view(P, k).index_fill_(0, g, 0.0)
        for r in range(N):
            c = counts[r]
            if c == 0:
                continue
            q = parts[r]
            ridx, rrows = q[1:1 + cap], q[1 + cap:].view(torch.float32).view(cap, width)
            g = ridx[:c].long()
            col = 0
            for name, k in self._row_fields(M):
                v[name].view(P, k).index_add_(0, g, rrows[:c, col:col + k])
                col += k
        self._adapt_cap(biggest)
        self.last_exchange = "sparse"
        return True
