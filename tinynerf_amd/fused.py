"""Fused K-Planes render path: the whole of ``NerfRenderer.forward`` (reference core.py:225-267) as one
autograd node that drives the C-ABI kernels back to back on the current stream.

The module-by-module path in ``core.NerfRenderer`` mirrors the reference's data flow, including its
boolean gather of the samples with w > 0 (a host sync, a [M,96] gather, an index_copy and their autograd
counterparts).  Here every sample goes through the colour head in place -- samples with w == 0 contribute
exactly 0 to the composite in both directions, as in the reference -- so there is no sync, no gather and
no intermediate autograd graph: forward = gather+sigma -> weights scan -> colour -> composite,
backward = composite -> colour head -> weights -> sigma head -> plane scatter, 9 launches in total.
Parameter gradients are accumulated straight into ``param.grad`` when it exists (the harness keeps the
gradient buffers allocated), which saves two passes over the 126 MiB of plane gradients per step.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, List, Optional, Sequence

import torch
from torch.autograd import Function

from . import _lib as L
from .models import _hwc, _kplanes_desc, _mlp_desc


class Arena:
    """Capacity-based scratch buffers for the harness: dynamic batches change N by a few percent every step, and a
    caching allocator that sees a new 2.4 GB workspace size every few steps falls back to hipMalloc in the middle of
    the step (measured: +60 ms stalls).  Buffers are allocated once with 25 % headroom and handed out as views.
    Only valid when each forward's backward runs before the next forward (the training loop); opt-in."""

    def __init__(self):
        self.buf = {}

    def get(self, name: str, shape, dev: torch.device, dtype=torch.float32) -> torch.Tensor:
        numel = 1
        for d in shape:
            numel *= int(d)
        t = self.buf.get(name)
        if t is None or t.numel() < numel or t.device != dev or t.dtype != dtype:
            t = torch.empty(int(numel * 1.25) + 1024, device=dev, dtype=dtype)
            self.buf[name] = t
        return t[:numel].view(*shape)


def _alloc(arena: Optional[Arena], name: str, shape, dev: torch.device) -> torch.Tensor:
    return arena.get(name, shape, dev) if arena is not None else torch.empty(tuple(shape), device=dev)


def _workspace(desc: L.MlpDesc, n: int, dev: torch.device, arena: Optional[Arena] = None, name: str = "ws"):
    fn = L.lib().tn_mlp_bwd_workspace_bytes
    fn.restype = C.c_int64
    nbytes = int(fn(C.byref(desc), C.c_int64(n)))
    if not nbytes:
        return None, 0
    return _alloc(arena, name, (nbytes // 4,), dev), nbytes


class _RenderKPlanes(Function):
    @staticmethod
    def forward(ctx: Any, packed: torch.Tensor, info: torch.Tensor, bg: Optional[torch.Tensor], thr: float,
                freqs: torch.Tensor, n_freqs: int, n_planes: int, n_sigma: int, accumulate: bool, arena: Optional[Arena],
                *params: torch.Tensor) -> torch.Tensor:  # type: ignore
        planes = list(params[:n_planes])
        sig_p = [p.contiguous() for p in params[n_planes:n_planes + n_sigma]]
        rgb_p = [p.contiguous() for p in params[n_planes + n_sigma:]]
        dev = L.require_cuda(packed, info, *sig_p, *rgb_p)
        n, R = packed.size(0), info.size(0)
        kdesc, keep = _kplanes_desc(planes)
        F = kdesc.n_scales * kdesc.channels
        feat = _alloc(arena, "feat", (n, F), dev)
        L.call("tn_kplanes_fwd", dev, C.byref(kdesc), L.ptr(packed), C.c_int64(7), C.c_int64(n), L.ptr(feat))
        sdesc = _mlp_desc(sig_p, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None)
        sigma = _alloc(arena, "sigma", (n,), dev)
        L.call("tn_mlp_fwd", dev, C.byref(sdesc), L.ptr(feat), C.c_void_p(None), C.c_int64(n), L.ptr(sigma), C.c_void_p(None))
        steps = _alloc(arena, "steps", (n,), dev).copy_(packed[:, 6])
        dirs = _alloc(arena, "dirs", (n, 3), dev).copy_(packed[:, 3:6])
        weights = _alloc(arena, "weights", (n,), dev)
        L.call("tn_weights_fwd", dev, L.ptr(sigma), L.ptr(steps), L.ptr(info), C.c_float(thr), L.ptr(weights),
               C.c_int64(n), C.c_int64(R))
        rdesc = _mlp_desc(rgb_p, F, L.ENC_DIR_CAT, n_freqs, L.ACT_SIGMOID, freqs)
        rgbs = _alloc(arena, "rgbs", (n, 3), dev)
        L.call("tn_mlp_fwd", dev, C.byref(rdesc), L.ptr(feat), L.ptr(dirs), C.c_int64(n), L.ptr(rgbs), C.c_void_p(None))
        out = torch.empty((R, 3), device=dev)
        L.call("tn_composite_fwd", dev, L.ptr(rgbs), L.ptr(weights), L.ptr(info), L.ptr(bg), L.ptr(out), C.c_void_p(None),
               C.c_int64(n), C.c_int64(R))
        ctx.save_for_backward(packed, info, bg, freqs, feat, sigma, steps, dirs, weights, rgbs, *params)
        ctx.cfg = (n_freqs, n_planes, n_sigma, accumulate)
        ctx.arena = arena
        ctx.param_refs = params if accumulate else None
        return out

    @staticmethod
    def backward(ctx: Any, grad_out: torch.Tensor):  # type: ignore
        packed, info, bg, freqs, feat, sigma, steps, dirs, weights, rgbs, *params = ctx.saved_tensors
        n_freqs, n_planes, n_sigma, accumulate = ctx.cfg
        planes = list(params[:n_planes])
        sig_p = [p.contiguous() for p in params[n_planes:n_planes + n_sigma]]
        rgb_p = [p.contiguous() for p in params[n_planes + n_sigma:]]
        dev = packed.device
        n, R = packed.size(0), info.size(0)
        F = feat.size(1)
        g_out = grad_out.contiguous()

        def grad_buffer(p: torch.Tensor, ref: Optional[torch.Tensor]):
            if accumulate and ref is not None and ref.grad is not None and ref.grad.stride() == p.stride():
                return ref.grad, True
            return torch.zeros_like(p), False

        refs: Sequence[Optional[torch.Tensor]] = ctx.param_refs if ctx.param_refs is not None else [None] * len(params)
        bufs = [grad_buffer(p, r) for p, r in zip(params, refs)]
        g_planes = [b[0] for b in bufs[:n_planes]]
        g_sig = [b[0] for b in bufs[n_planes:n_planes + n_sigma]]
        g_rgb = [b[0] for b in bufs[n_planes + n_sigma:]]

        arena = ctx.arena
        g_rgbs = _alloc(arena, "g_rgbs", (n, 3), dev)
        g_w = _alloc(arena, "g_w", (n,), dev)
        L.call("tn_composite_bwd", dev, L.ptr(rgbs), L.ptr(weights), L.ptr(info), L.ptr(bg), L.ptr(g_out), L.ptr(g_rgbs),
               L.ptr(g_w), C.c_int64(n), C.c_int64(R))
        # colour head: grads of its parameters + d/d feat
        rdesc = _mlp_desc(rgb_p, F, L.ENC_DIR_CAT, n_freqs, L.ACT_SIGMOID, freqs)
        nl = len(rgb_p) // 2
        gw = (C.c_void_p * nl)(*[g.data_ptr() for g in g_rgb[0::2]])
        gb = (C.c_void_p * nl)(*[g.data_ptr() for g in g_rgb[1::2]])
        g_feat = _alloc(arena, "g_feat", (n, F), dev)
        ws, ws_bytes = _workspace(rdesc, n, dev, arena, "ws_rgb")
        L.call("tn_mlp_bwd", dev, C.byref(rdesc), L.ptr(feat), L.ptr(dirs), L.ptr(g_rgbs), C.c_int64(n), gw, gb, L.ptr(g_feat),
               L.ptr(ws), C.c_int64(ws_bytes))
        # weights -> sigma
        g_sigma = _alloc(arena, "g_sigma", (n,), dev).zero_()
        L.call("tn_weights_bwd", dev, L.ptr(sigma), L.ptr(steps), L.ptr(info), L.ptr(weights), L.ptr(g_w), L.ptr(g_sigma),
               C.c_int64(n), C.c_int64(R))
        sdesc = _mlp_desc(sig_p, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None, L.MLP_ACCUM_GRAD_X)   # g_feat += d sigma / d feat
        nl = len(sig_p) // 2
        gw = (C.c_void_p * nl)(*[g.data_ptr() for g in g_sig[0::2]])
        gb = (C.c_void_p * nl)(*[g.data_ptr() for g in g_sig[1::2]])
        ws, ws_bytes = _workspace(sdesc, n, dev, arena, "ws_sigma")
        L.call("tn_mlp_bwd", dev, C.byref(sdesc), L.ptr(feat), C.c_void_p(None), L.ptr(g_sigma), C.c_int64(n), gw, gb,
               L.ptr(g_feat), L.ptr(ws), C.c_int64(ws_bytes))
        # plane scatter
        kdesc, keep = _kplanes_desc(planes)
        gp = ((C.c_void_p * 3) * L.TN_KPLANES_MAX_SCALES)()
        for s in range(kdesc.n_scales):
            for p in range(3):
                gp[s][p] = _hwc(g_planes[3 * s + p]).data_ptr()
        L.call("tn_kplanes_bwd", dev, C.byref(kdesc), L.ptr(packed), C.c_int64(7), C.c_int64(n), L.ptr(g_feat), gp)
        grads = [None if in_place else g for (g, in_place) in bufs]
        return (None, None, None, None, None, None, None, None, None, None, *grads)


def supports(renderer) -> bool:
    from .models import KPlanesFeatureField, VanillaColorDecoder, VanillaOpacityDecoder
    fm, sd, cd = renderer.feature_module, renderer.sigma_decoder, renderer.rgb_decoder
    return (isinstance(fm, KPlanesFeatureField) and isinstance(sd, VanillaOpacityDecoder) and isinstance(cd, VanillaColorDecoder)
            and fm.dropout.p == 0.0 and type(sd) is VanillaOpacityDecoder and type(cd) is VanillaColorDecoder)


def render(renderer, packed: torch.Tensor, info: torch.Tensor, thr: float, accumulate_into_grad: bool = False) -> torch.Tensor:
    """Fused forward of ``renderer`` (a NerfRenderer with K-Planes field + Vanilla decoders) on packed samples."""
    fm, sd, cd = renderer.feature_module, renderer.sigma_decoder, renderer.rgb_decoder
    planes = fm.plane_tensors()
    sig_p, rgb_p = sd.net.params(), cd.net.params()
    bg = renderer._bg(packed.device)
    arena = None
    if getattr(renderer, "reuse_buffers", False):
        arena = renderer.__dict__.setdefault("_arena", Arena())
    return _RenderKPlanes.apply(packed.contiguous(), info.contiguous(), bg, float(thr), cd.pe.freqs, cd.n_freqs, len(planes),
                                len(sig_p), accumulate_into_grad, arena, *planes, *sig_p, *rgb_p)
