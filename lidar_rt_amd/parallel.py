"""Azimuth-sector sharding of one LiDAR frame across the GPUs of a node
(not present in the reference, which is single-GPU).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI;
"gloo" on CPU for the tests).  Every rank holds the full Gaussian set and
builds an identical read-only LBVH; rank r traces the contiguous column slab
``[r*W/N, (r+1)*W/N)`` of the (H, W) range image.

* forward : local slab -> ``all_gather`` of the (H, W/N, 9) slabs (4.7 MB at
  64x2048, negligible) so every rank sees the whole image for image-space
  losses; per-Gaussian hit weights are part of the fused reduction below.
* backward: each rank scatters its partial gradients into ONE flat fp32 buffer
  ``[d_means | d_scales | d_rotations | d_opacities | d_shs | accum]``
  (59 + 1 floats per Gaussian at M=16), reduced by ONE ``all_reduce`` (sum).
  xGMI is a point-to-point mesh, so a single large collective lets RCCL use
  all 7 links (reduce-scatter + all-gather); many small ones would be
  latency-bound.

The local tracer is injected (``backend``): the product default drives the HIP
library; the CPU tests inject an oracle-backed stand-in to exercise the
collective logic with gloo (no GPU in CI).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

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
        return out, accum

    def backward(self, ray_o, ray_d, means, scales, rotations, opacities, shs, deg, bg, out, dL, mod=1.0, grads_out=None):
        e = torch.empty(0, device=means.device)
        g = self._C.trace_surfels_backward(self.state, ray_o, ray_d, e, bg, means, shs, deg, e, opacities, scales,
                                           mod, rotations, e, e, e, e, False, False, out, None, dL, grads_out=grads_out)
        return {"means": g[0], "shs": g[1], "opacities": g[3], "scales": g[4], "rotations": g[5]}


class ShardedTracer:
    """Traces one frame sharded by azimuth sector over ``dist``'s world."""

    def __init__(self, backend=None, group=None, exchange: str = "auto"):
        """exchange: "dense" = one all_reduce of the flat gradient buffer; "sparse" = all_gather of the rows of the
        Gaussians each rank's rays touched (azimuth sectors touch mostly disjoint Gaussians); "auto" = sparse unless the
        ranks together touched more than `sparse_max_fraction` of the Gaussians."""
        if exchange not in ("auto", "dense", "sparse"):
            raise ValueError("exchange must be 'auto', 'dense' or 'sparse'")
        self.exchange = exchange
        self.sparse_max_fraction = 0.6
        self.last_exchange = None                      # what the last backward used: "dense" | "sparse" | None
        self.backend = backend if backend is not None else HipBackend()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # build the LBVH for this rank's rays only (lrt_build_for_rays; meaningful from 3 ranks on).  The library sizes the sort
        # and the tree from the previous frame's kept count, so no read-back stalls the launch queue; on S1M the culled build
        # beats the full one from ~1/8 of the frame per rank on (N=8: 0.25 -> 0.19-0.22 ms; N=4: 0.25 -> 0.27 ms), hence the
        # default: on for 8 ranks and more
        self.cull_build = self.world >= 8
        if os.environ.get("LRT_CULL_BUILD", "") in ("0", "1"):         # developer / test switch
            self.cull_build = os.environ["LRT_CULL_BUILD"] == "1"
        self._phase_on = False
        self._phase_ev = []            # (name, start event, end event) of the collectives' regions, read back by phase_timing()

    # ---- per-phase timing of the collective regions (bench.py --gpus N: build / forward / backward come from the library's HIP events)
    def enable_phase_timing(self, on: bool = True):
        self._phase_on = bool(on); self._phase_ev = []

    class _Region:
        def __init__(self, owner, name, device):
            self.o, self.name, self.cuda = owner, name, (device.type == "cuda" and owner._phase_on)
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

    @staticmethod
    def _backend_takes(fn, name: str) -> bool:
        """Capability of an injected backend, from its signature (a TypeError raised INSIDE the backend must not be mistaken for
        a missing keyword)."""
        import inspect
        try:
            ps = inspect.signature(fn).parameters
        except (TypeError, ValueError):
            return False
        return name in ps or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in ps.values())

    def forward(self, ray_o, ray_d, means, scales, rotations, opacities, shs, deg, bg, mod=1.0, rebuild=True):
        H, W = ray_o.shape[:2]
        a, b = column_slab(W, self.rank, self.world)
        self._slab = (a, b)
        self._ro = ray_o[:, a:b].contiguous(); self._rd = ray_d[:, a:b].contiguous()
        # a rank only needs the Gaussians its slab's rays can reach: optionally (cull_build) the LBVH is built for the slab's ray
        # cone from 3 ranks on (slabs narrower than ~120 degrees); such a structure must be rebuilt for every ray set
        cull = (self._ro, self._rd) if (self.world >= 3 and self.cull_build) else None
        if rebuild or cull is not None:
            if self._backend_takes(self.backend.build, "cull_rays"):
                self.backend.build(means, scales, rotations, opacities, mod, cull_rays=cull)
            else:                                                                 # backend without culling (test stand-ins)
                self.backend.build(means, scales, rotations, opacities, mod)
        out_loc, accum_loc = self.backend.forward(self._ro, self._rd, means, scales, rotations, opacities, shs,
                                                  deg, bg, mod)
        self._out_loc, self._accum_loc = out_loc, accum_loc
        if self.world == 1:
            return out_loc, accum_loc
        # all_gather needs equal shapes: pad slabs to the widest one
        wmax = max(column_slab(W, r, self.world)[1] - column_slab(W, r, self.world)[0] for r in range(self.world))
        with self._Region(self, "slab_all_gather", out_loc.device):
            pad = torch.zeros((H, wmax, 9), dtype=out_loc.dtype, device=out_loc.device)
            pad[:, :b - a] = out_loc
            parts = self._all_gather_rows(pad)               # one flat receive buffer with RCCL
            cols = []
            for r in range(self.world):
                ra, rb = column_slab(W, r, self.world)
                cols.append(parts[r][:, :rb - ra])
            full = torch.cat(cols, dim=1)
        return full, accum_loc                          # accum is completed by backward()'s fused reduction

    def backward(self, means, scales, rotations, opacities, shs, deg, bg, dL_full, mod=1.0,
                 reduce: bool = True) -> Dict[str, torch.Tensor]:
        a, b = self._slab
        dL = dL_full[:, a:b].contiguous()
        P = means.shape[0]; M = shs.shape[1]
        lay = getattr(self, "_layout", None)
        if lay is None or lay.P != P or lay.M != M or lay.flat.device != means.device:
            lay = self._layout = GradLayout(P, M, means.device)           # reused across steps: no per-step 240 MB allocation
        direct = {k: lay.views[k] for k in ("means", "scales", "rotations", "opacities", "shs")}
        if self._backend_takes(self.backend.backward, "grads_out"):
            self.backend.backward(self._ro, self._rd, means, scales, rotations, opacities, shs, deg, bg,
                                  self._out_loc, dL, mod, grads_out=direct)      # kernels write straight into the flat buffer
        else:                                                                     # backend without grads_out (test stand-ins)
            g = self.backend.backward(self._ro, self._rd, means, scales, rotations, opacities, shs, deg, bg,
                                      self._out_loc, dL, mod)
            for k in direct:
                direct[k].copy_(g[k].view_as(direct[k]))
        lay.views["accum"].copy_(self._accum_loc)
        self.last_exchange = None
        if reduce and self.world > 1:
            with self._Region(self, "gradient_exchange", lay.flat.device):
                if self.exchange == "dense" or not self._exchange_sparse(lay):
                    dist.all_reduce(lay.flat, op=dist.ReduceOp.SUM, group=self.group)
                    self.last_exchange = "dense"
        return lay.views

    @staticmethod
    def _grad_rows(fn: str, lay: GradLayout, n: int, ids: torch.Tensor, rows: torch.Tensor):
        """lrt_grad_gather / lrt_grad_scatter_add (include/lrt.h) on the views of the flat buffer."""
        import ctypes as C
        from . import _capi
        v, p = lay.views, _capi.ptr
        dev = rows.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        dense = (p(v["means"]), p(v["scales"]), p(v["rotations"]), p(v["opacities"]), p(v["shs"]), p(v["accum"]))
        lib = _capi.load()
        if fn == "lrt_grad_gather":
            rc = lib.lrt_grad_gather(idx, lay.P, lay.M, int(n), p(ids), *dense, p(rows), stream)
        else:
            rc = lib.lrt_grad_scatter_add(idx, lay.P, lay.M, int(n), p(ids), p(rows), *dense, stream)
        _capi.check(rc, fn)

    def _all_gather_rows(self, t: torch.Tensor):
        """all_gather of equally shaped tensors; one flat receive buffer when the backend offers it (RCCL), so that the
        collective is a single large message without a per-rank copy."""
        world = self.world
        if t.is_cuda:
            try:
                out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
                dist.all_gather_into_tensor(out.view(-1, *t.shape[1:]) if t.dim() > 0 else out, t, group=self.group)
                return [out[r] for r in range(world)]
            except (RuntimeError, NotImplementedError, AttributeError):
                pass                                                  # e.g. gloo with device tensors: fall through
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=self.group)
        return parts

    def _exchange_sparse(self, lay: GradLayout) -> bool:
        """Sum the ranks' partial gradients by exchanging only the rows of touched Gaussians.

        A Gaussian has a non-zero partial gradient on a rank only if one of that rank's rays composited it, and every
        composited hit adds a weight > 0 to `accum` (forward.cu:268), so `accum > 0` is the exact mask.  Each rank
        all_gathers (index, 60-float row) of its touched Gaussians (padded to the largest count), clears its buffer and
        adds the ranks' rows in rank order: indices are unique inside a rank, so every rank ends with bit-identical
        sums.  Returns False (nothing exchanged) when the dense all_reduce moves fewer bytes."""
        P, M, world = lay.P, lay.M, self.world
        fields = [(name, k) for name, k in GradLayout.FIELDS] + [("shs", 3 * M), ("accum", 1)]
        width = sum(k for _, k in fields)
        dev = lay.flat.device
        idx = torch.nonzero(lay.views["accum"] > 0).squeeze(1)
        n = torch.tensor([idx.numel()], dtype=torch.int64, device=dev)
        counts = torch.cat(self._all_gather_rows(n)).tolist()         # the same list on every rank; ONE device->host copy
        nmax = max(counts)
        if self.exchange == "auto" and sum(counts) > self.sparse_max_fraction * P:
            return False
        pay = torch.zeros((max(nmax, 1), width), dtype=torch.float32, device=dev)
        ids = torch.zeros(max(nmax, 1), dtype=torch.int32, device=dev)
        m = idx.numel()
        hip = dev.type == "cuda"                                      # HIP tensors: one pack launch and one add launch per rank
        if m:
            ids[:m] = idx.to(torch.int32)
            if hip:
                self._grad_rows("lrt_grad_gather", lay, m, ids, pay)
            else:
                col = 0
                for name, k in fields:
                    pay[:m, col:col + k] = lay.views[name].reshape(P, k).index_select(0, idx)
                    col += k
        pays, idss = self._all_gather_rows(pay), self._all_gather_rows(ids)
        lay.flat.zero_()
        for r in range(world):                                        # fixed order -> identical rounding on every rank
            c = counts[r]
            if c == 0:
                continue
            if hip:
                self._grad_rows("lrt_grad_scatter_add", lay, c, idss[r], pays[r])
                continue
            ridx = idss[r][:c].long()
            col = 0
            for name, k in fields:
                lay.views[name].view(P, k).index_add_(0, ridx, pays[r][:c, col:col + k])
                col += k
        self.last_exchange = "sparse"
        return True
