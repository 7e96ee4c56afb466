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
import os
from typing import Any, List, Optional, Sequence

import torch
from torch.autograd import Function

from . import _lib as L
from .arena import Arena
from .config import CONFIG
from .models import _empty_rows, _hwc, _kplanes_desc, _mlp_desc


# Heads behind a wide stack take their first-layer operands from the stack's workspace rows, and the stack stops writing the row-major
# copy (TN_MLP_ROWS_ONLY / TN_MLP_X_FROM_ROWS; f16x2 heads only).  TN_ROWS_HANDOFF=0: row-major as before (A/B runs, tests).
ROWS_HANDOFF = CONFIG.rows_handoff


def MATMUL_F16X2() -> bool:
    from . import models
    return ROWS_HANDOFF and models.MATMUL == "f16x2"


PAIR_FORWARD = True       # both heads' training forwards in one launch (tn_mlp_fwd_stash_pair)
FUSE_GATHER = True        # ... with the K-Planes gather inside that launch (tn_kplanes_mlp_fwd_pair)
FUSE_SCATTER = True       # backward: the plane scatter inside the data-gradient chain launch (tn_kplanes_mlp_bwd_pair)
PAIR_BACKWARD = True      # both heads' data gradients in one launch (tn_mlp_bwd_pair); False: one tn_mlp_bwd per head
# ... also behind the wide stacks (_RenderHeads; round 5): tn_mlp_bwd_pair takes both heads' first-layer weight gradients over the x
# columns in ONE launch (x rows read once) and, where both first layers fit LDS (128-wide stack), both data gradients in one pass; "0": off
HEADS_PAIR_BACKWARD = CONFIG.heads_pair
# round 5 (TN_MLP_SKIP_LAST): behind a wide stack whose last layer is a plain Linear (Vanilla 256 -> 256, Cobafa 128 -> 128: reference
# models.py:59-68,239-247) the heads' first layers are Linear too -- the harness merges the two (W_head W_last, W_head b_last + b_head: five
# small torch matmuls per step, differentiable, so the original parameters get their gradients by the chain rule) and the stack stops at its
# last hidden activation: one layer launch less in the forward pass, two less in the backward pass, the feature tensor never exists
MERGE_LAST = CONFIG.merge_last
# round 5 (TN_MLP_LEAN): the paired f16x2 training forward writes masks / pre-activations / feature rows only, the weight-gradient launches
# rebuild the hidden activations from the feature rows (csrc/mlp_wgrad_rc.hip) -- 2.7 GB less workspace traffic per K-Planes step.
# TN_KP_LEAN=0: the stash-everything form of rounds 1-4 (A/B runs; always taken by the fp32 / bf16x3 head forms)
KP_LEAN = CONFIG.kp_lean
_side_streams: dict = {}


def _side_stream(dev: torch.device) -> "torch.cuda.Stream":
    s = _side_streams.get(dev)
    if s is None:
        s = _side_streams[dev] = torch.cuda.Stream(dev)
    return s


def _alloc(arena: Optional[Arena], name: str, shape, dev: torch.device, dtype=torch.float32) -> torch.Tensor:
    return arena.get(name, shape, dev, dtype) if arena is not None else torch.empty(tuple(shape), device=dev, dtype=dtype)


def _workspace(desc: L.MlpDesc, n: int, dev: torch.device, arena: Optional[Arena] = None, name: str = "ws"):
    fn = L.lib().tn_mlp_bwd_workspace_bytes
    fn.restype = C.c_int64
    nbytes = int(fn(C.byref(desc), C.c_int64(n)))
    if not nbytes:
        return None, 0
    return _alloc(arena, name, (nbytes // 4,), dev), nbytes


def _ray_aux(packed: torch.Tensor, info: torch.Tensor, freqs: torch.Tensor, n_freqs: int, arena: Optional[Arena], hint: Optional[dict]):
    """Per-ray inputs of the colour head (models.py:87: cat[PE(d), d]): evaluated once per ray into a table
    (tn_dir_encode) that the MLP kernels index through the ray id of every sample, instead of 48 sin/cos per sample.
    Returns (table, ray_ids, stride, steps).  ``hint``: ray ids / steps / ray directions the sampler already wrote out
    for exactly this batch (run.Trainer.build_batch); otherwise they are rebuilt from (packed, info)."""
    dev = packed.device
    n, R = packed.size(0), info.size(0)
    stride = (6 * n_freqs + 3 + 7) & ~7
    table = _alloc(arena, "aux_table", (R, stride), dev)
    if hint is not None and hint.get("key") == (packed.data_ptr(), n, R):
        ray_ids, steps, dirs_ray = hint["ray_ids"], hint["steps"], hint["dirs"]
    else:
        ray_ids = _alloc(arena, "ray_ids", (n,), dev, torch.int32)
        steps = _alloc(arena, "steps", (n,), dev)
        dirs_ray = _alloc(arena, "dirs_ray", (R, 3), dev)
        L.call("tn_ray_aux", dev, L.ptr(packed), L.ptr(info), C.c_int64(R), L.ptr(ray_ids), L.ptr(steps), L.ptr(dirs_ray))
    if n > 0:
        L.call("tn_dir_encode", dev, L.ptr(dirs_ray), C.c_int64(R), L.ptr(freqs), C.c_int(n_freqs), L.ptr(table), C.c_int(stride))
    return table, ray_ids, stride, steps


def _gate_slot(hint: Optional[dict], wanted: bool) -> Optional[torch.Tensor]:
    """the trainer's "Empty iteration" flag slot for this batch (the weights kernel only ever RAISES it): fresh -- zeroed by the
    ring's lap -- for the first forward on a batch, cleared here for any further forward on the same batch (another threshold,
    other parameters), which would otherwise inherit the earlier forward's 1.0"""
    if not wanted or hint is None or hint.get("gate") is None:
        return None
    if hint.get("gate_used"):
        hint["gate"].zero_()
    hint["gate_used"] = True
    return hint["gate"]


def _upstream_is_gated(ctx: Any) -> bool:
    """The node's backward applies the gate itself unless the caller has DECLARED the upstream gradient gated
    (run.Trainer.step_on_batch sets stats["upstream_gated"] around tn_mse_grad_gated): any other loss on a trainer-built batch
    -- a test, a custom loop -- then still gets the reference's zero gradients in an all-masked step (core.py:251-254)"""
    return bool(ctx.gate_in_slot and ctx.stats is not None and ctx.stats.get("upstream_gated"))


INFER_PAIR = CONFIG.infer_pair       # (TN_INFER_PAIR=0: always the gated inference form -- A/B, debugging)
INFER_PAIR_MIN_LIVE = 0.6       # the pair form is taken at >= this live fraction ...
INFER_PAIR_HYSTERESIS = 0.15    # ... and kept until the fraction falls below MIN_LIVE - HYSTERESIS (chunks of an image alternate between background
                                # and object: without the band the form -- not the result -- would flip from chunk to chunk)


def _infer_prefers_pair(stats: Optional[dict]) -> bool:
    """live-sample fraction (w > 0) of the most recent inference call on this renderer whose measurement has LANDED, read without
    a host sync: the values travel to a small ring of pinned slots, each behind its own event.  Until one has landed the gated
    form runs: on a trained scene (most samples behind a terminated ray) it is 4 x faster than the pair form (48 against 188 ms
    per 800 x 800 image), on a field that is alive everywhere only 1.4 x slower -- the cheap mistake is the default."""
    st = None if stats is None else stats.get("infer_live")
    if st is None:
        return False
    for slot in st["slots"]:
        if slot["seq"] > st["seen"] and slot["event"].query():
            st["seen"], st["value"] = slot["seq"], float(slot["pinned"][0])
    if st["value"] is None:
        return False
    on = st.get("pair_on", False)
    on = st["value"] >= (INFER_PAIR_MIN_LIVE - INFER_PAIR_HYSTERESIS if on else INFER_PAIR_MIN_LIVE)
    st["pair_on"] = on
    return on


def _note_live_fraction(stats: dict, weights: torch.Tensor) -> None:
    st = stats.get("infer_live")
    if st is None:
        st = stats["infer_live"] = {"slots": [{"pinned": torch.ones(1, pin_memory=True), "event": torch.cuda.Event(), "seq": 0} for _ in range(4)],
                                    "seq": 0, "seen": 0, "value": None}
    if not weights.numel():
        return
    for slot in st["slots"]:                      # a slot whose previous measurement has landed (or that was never used)
        if slot["seq"] == 0 or (slot["seq"] <= st["seen"]) or slot["event"].query():
            if slot["seq"] > st["seen"]:          # landed but not read yet: read it before it is overwritten
                st["seen"], st["value"] = slot["seq"], float(slot["pinned"][0])
            st["seq"] += 1
            slot["seq"] = st["seq"]
            # (one reduction launch + one 4-byte copy; the count is divided on the host side of the pinned slot)
            slot["pinned"].copy_((torch.count_nonzero(weights) / float(weights.numel())).reshape(1).float(), non_blocking=True)
            slot["event"].record(torch.cuda.current_stream(weights.device))
            return
    # every slot still in flight: skip this measurement


class _RenderKPlanes(Function):
    @staticmethod
    def forward(ctx: Any, packed: torch.Tensor, info: torch.Tensor, bg: Optional[torch.Tensor], thr: float,
                freqs: torch.Tensor, n_freqs: int, n_planes: int, n_sigma: int, accumulate: bool, arena: Optional[Arena],
                train: bool, hint: Optional[dict], stats: Optional[dict], *params: torch.Tensor) -> torch.Tensor:  # type: ignore
        planes = list(params[:n_planes])
        sig_p = [p.contiguous() for p in params[n_planes:n_planes + n_sigma]]
        rgb_p = [p.contiguous() for p in params[n_planes + n_sigma:]]
        dev = L.require_cuda(packed, info, *sig_p, *rgb_p)
        n, R = packed.size(0), info.size(0)
        kdesc, keep = _kplanes_desc(planes)
        F = kdesc.n_scales * kdesc.channels
        feat = _alloc(arena, "feat", (n, F), dev)
        table, ray_ids, stride, steps = _ray_aux(packed, info, freqs, n_freqs, arena, hint)
        sdesc = _mlp_desc(sig_p, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None)
        rdesc = _mlp_desc(rgb_p, F, L.ENC_AUX_CAT, n_freqs, L.ACT_SIGMOID, freqs, 0, ray_ids, stride)
        ws_s = ws_r = None
        sb = rb = 0
        if train:          # training forward: activations go to the backward's workspace, nothing is recomputed
            ws_s, sb = _workspace(sdesc, n, dev, arena, "ws_sigma")
            ws_r, rb = _workspace(rdesc, n, dev, arena, "ws_rgb")
        sigma = _alloc(arena, "sigma", (n,), dev)
        rgbs = _alloc(arena, "rgbs", (n, 3), dev)
        pair = (ws_s is not None and ws_r is not None and PAIR_FORWARD and F % 4 == 0 and sig_p[0].size(0) == 64 and rgb_p[0].size(0) == 64)
        # TN_MLP_LEAN: no hidden activations in the workspace, the weight-gradient launches rebuild them (csrc/mlp_wgrad_rc.hip)
        lean = bool(pair and KP_LEAN and PAIR_BACKWARD and L.lib().tn_mlp_lean_supported(C.byref(rdesc), C.byref(sdesc)))
        if lean:
            rdesc.flags |= L.MLP_LEAN
            sdesc.flags |= L.MLP_LEAN
        gather_fused = pair and FUSE_GATHER and kdesc.n_scales == 3 and kdesc.channels == 32 and len(keep) == 9
        gather_sigma = (not train and FUSE_GATHER and F % 4 == 0 and sig_p[0].size(0) == 64 and kdesc.n_scales == 3 and kdesc.channels == 32
                        and len(keep) == 9)
        # inference, two forms with identical results (a sample with w == 0 contributes exactly 0 either way, core.py:243-249):
        #   gated: gather + sigma head -> weights -> colour head on the 32-sample tiles that hold a weight -> composite;
        #   pair:  gather + BOTH heads of every sample in one launch (no feature rows, nothing stashed) -> weights + composite.
        # The pair wins while most tiles are alive (an untrained or half-trained field: 0.6 against 0.9 ms per 2^20 samples), the
        # gated form once early termination has emptied most of them; the choice follows the live fraction of the previous call
        infer_pair = gather_sigma and INFER_PAIR and rgb_p[0].size(0) == 64 and len(rgb_p) == 10 and _infer_prefers_pair(stats)
        if infer_pair:
            L.call("tn_kplanes_mlp_fwd_pair", dev, C.byref(kdesc), L.ptr(packed), C.c_int64(7), C.byref(rdesc), C.byref(sdesc), L.ptr(table),
                   C.c_int64(n), L.ptr(feat), L.ptr(rgbs), L.ptr(sigma), C.c_void_p(None), C.c_int64(0), C.c_void_p(None), C.c_int64(0))
        elif gather_sigma:   # inference: gather + sigma head in one launch
            L.call("tn_kplanes_mlp_fwd", dev, C.byref(kdesc), L.ptr(packed), C.c_int64(7), C.byref(sdesc), C.c_int64(n), L.ptr(feat), L.ptr(sigma))
        elif gather_fused:   # gather + both heads in ONE launch: the feature rows go from the texel lines to the MFMA operands
            L.call("tn_kplanes_mlp_fwd_pair", dev, C.byref(kdesc), L.ptr(packed), C.c_int64(7), C.byref(rdesc), C.byref(sdesc), L.ptr(table),
                   C.c_int64(n), L.ptr(feat), L.ptr(rgbs), L.ptr(sigma), L.ptr(ws_r), C.c_int64(rb), L.ptr(ws_s), C.c_int64(sb))
        else:
            L.call("tn_kplanes_fwd", dev, C.byref(kdesc), L.ptr(packed), C.c_int64(7), C.c_int64(n), L.ptr(feat))
        if gather_fused or gather_sigma:
            pass
        elif pair:         # both heads in one launch: the feature rows are read from HBM once
            L.call("tn_mlp_fwd_stash_pair", dev, C.byref(rdesc), C.byref(sdesc), L.ptr(feat), L.ptr(table), C.c_int64(n), L.ptr(rgbs),
                   L.ptr(sigma), L.ptr(ws_r), C.c_int64(rb), L.ptr(ws_s), C.c_int64(sb))
        elif ws_s is not None:
            L.call("tn_mlp_fwd_stash", dev, C.byref(sdesc), L.ptr(feat), C.c_void_p(None), C.c_int64(n), L.ptr(sigma), L.ptr(ws_s), C.c_int64(sb))
        else:
            L.call("tn_mlp_fwd", dev, C.byref(sdesc), L.ptr(feat), C.c_void_p(None), C.c_int64(n), L.ptr(sigma), C.c_void_p(None))
        weights = _alloc(arena, "weights", (n,), dev)
        covered = hint is not None and hint.get("key") == (packed.data_ptr(), n, R)     # the trainer's sampler covers every sample
        if not covered:
            weights.zero_()          # cuda.cu:84 (zeros_like): samples outside every (start, count) keep weight 0
        # harness: a zeroed [1] slot that the weights kernel raises when any weight is > 0 (instead of a reduction launch), and an
        # upstream gradient that arrives gated (tn_mse_grad_gated) -- see the "Empty iteration" note below
        gate_slot = _gate_slot(hint, covered and train)
        out = torch.empty((R, 3), device=dev)
        if pair or infer_pair:
            # both heads are done: weights and composite of a ray in one launch (tn_render_rays_fwd, bit-identical to the two)
            L.call("tn_render_rays_fwd", dev, L.ptr(sigma), L.ptr(steps), L.ptr(rgbs), L.ptr(info), L.ptr(bg), C.c_float(thr), L.ptr(weights),
                   L.ptr(out), L.ptr(gate_slot), C.c_int64(n), C.c_int64(R))
        elif gate_slot is not None:
            L.call("tn_weights_fwd_gate", dev, L.ptr(sigma), L.ptr(steps), L.ptr(info), C.c_float(thr), L.ptr(weights), L.ptr(gate_slot),
                   C.c_int64(n), C.c_int64(R))
        else:
            L.call("tn_weights_fwd", dev, L.ptr(sigma), L.ptr(steps), L.ptr(info), C.c_float(thr), L.ptr(weights),
                   C.c_int64(n), C.c_int64(R))
        if pair or infer_pair:
            pass
        elif ws_r is not None:
            L.call("tn_mlp_fwd_stash", dev, C.byref(rdesc), L.ptr(feat), L.ptr(table), C.c_int64(n), L.ptr(rgbs), L.ptr(ws_r), C.c_int64(rb))
        else:              # inference: the colour head is only evaluated where the weight is not 0 (core.py:246-251), tile-wise
            rdesc.row_gate = weights.data_ptr()
            L.call("tn_mlp_fwd", dev, C.byref(rdesc), L.ptr(feat), L.ptr(table), C.c_int64(n), L.ptr(rgbs), C.c_void_p(None))
            rdesc.row_gate = None
        if not (pair or infer_pair):
            L.call("tn_composite_fwd", dev, L.ptr(rgbs), L.ptr(weights), L.ptr(info), L.ptr(bg), L.ptr(out), C.c_void_p(None),
                   C.c_int64(n), C.c_int64(R))
        if gather_sigma and INFER_PAIR and stats is not None:
            _note_live_fraction(stats, weights)
        # core.py:246-254: when EVERY sample is masked (w == 0 everywhere) the reference renders the background from constants
        # that carry no graph, i.e. no parameter receives a gradient from the image loss; here the upstream gradient is gated
        # (a [1] tensor kept outside save_for_backward: with N > 1 the trainer all-reduces it in place right after this forward
        # (run.Trainer.step_on_batch) -- the single-GPU step on the union of the ranks' rays is only "empty" when every
        # rank's is -- so the backward below already reads the all-rank value)
        ctx.gate_in_slot, ctx.stats = gate_slot is not None, stats
        ctx.gate = gate_slot if gate_slot is not None else (weights.amax().reshape(1) if train else None)
        if stats is not None:
            stats["gate"] = ctx.gate
            stats["pre_gated"] = ctx.gate_in_slot        # the caller MAY hand in a gated gradient (and must then say so)
            stats["upstream_gated"] = False
        ctx.save_for_backward(packed, info, bg, freqs, feat, sigma, steps, table, ray_ids, weights, rgbs, ws_s, ws_r, *params)
        ctx.cfg = (n_freqs, n_planes, n_sigma, accumulate, stride, sb, rb, covered)
        ctx.lean = lean
        ctx.arena = arena
        ctx.param_refs = params if accumulate else None
        ctx.planes_ready = hint.get("planes_ready") if (hint is not None and accumulate) else None
        return out

    @staticmethod
    def backward(ctx: Any, grad_out: torch.Tensor):  # type: ignore
        packed, info, bg, freqs, feat, sigma, steps, table, ray_ids, weights, rgbs, ws_s, ws_r, *params = ctx.saved_tensors
        n_freqs, n_planes, n_sigma, accumulate, stride, sb, rb, covered = ctx.cfg
        planes = list(params[:n_planes])
        sig_p = [p.contiguous() for p in params[n_planes:n_planes + n_sigma]]
        rgb_p = [p.contiguous() for p in params[n_planes + n_sigma:]]
        dev = packed.device
        n, R = packed.size(0), info.size(0)
        F = feat.size(1)
        g_out = grad_out.contiguous()
        if ctx.gate is not None and not _upstream_is_gated(ctx):
            g_out = g_out * (ctx.gate > 0).to(g_out.dtype)       # "Empty iteration": zero gradients, as on the module-by-module path

        def grad_buffer(p: torch.Tensor, ref: Optional[torch.Tensor]):
            if accumulate and ref is not None and ref.is_leaf and ref.grad is not None and ref.grad.stride() == p.stride():
                return ref.grad, True
            return torch.zeros_like(p), False

        refs: Sequence[Optional[torch.Tensor]] = ctx.param_refs if ctx.param_refs is not None else [None] * len(params)
        bufs = [grad_buffer(p, r) for p, r in zip(params, refs)]
        g_planes = [b[0] for b in bufs[:n_planes]]
        g_sig = [b[0] for b in bufs[n_planes:n_planes + n_sigma]]
        g_rgb = [b[0] for b in bufs[n_planes + n_sigma:]]

        arena = ctx.arena
        g_rgbs = _alloc(arena, "g_rgbs", (n, 3), dev)
        g_sigma = _alloc(arena, "g_sigma", (n,), dev)
        if not covered:              # samples no ray owns: zero gradient, not whatever the arena held (the kernel writes every
            g_rgbs.zero_(); g_sigma.zero_()      # sample a ray owns)
        # composite -> (rgbs, weights) and weights -> sigma as one launch per ray (the weights' gradient needs only the composite's)
        L.call("tn_render_rays_bwd", dev, L.ptr(sigma), L.ptr(steps), L.ptr(rgbs), L.ptr(info), L.ptr(bg), L.ptr(weights), L.ptr(g_out),
               L.ptr(g_rgbs), L.ptr(g_sigma), C.c_int64(n), C.c_int64(R))
        g_feat = _alloc(arena, "g_feat", (n, F), dev)
        nr, ns = len(rgb_p) // 2, len(sig_p) // 2
        gw_r = (C.c_void_p * nr)(*[g.data_ptr() for g in g_rgb[0::2]])
        gb_r = (C.c_void_p * nr)(*[g.data_ptr() for g in g_rgb[1::2]])
        gw_s = (C.c_void_p * ns)(*[g.data_ptr() for g in g_sig[0::2]])
        gb_s = (C.c_void_p * ns)(*[g.data_ptr() for g in g_sig[1::2]])
        kdesc, keep = _kplanes_desc(planes)
        gp = ((C.c_void_p * 3) * L.TN_KPLANES_MAX_SCALES)()
        for s in range(kdesc.n_scales):
            for p in range(3):
                gp[s][p] = _hwc(g_planes[3 * s + p]).data_ptr()
        scattered = False

        def scatter():
            L.call("tn_kplanes_bwd", dev, C.byref(kdesc), L.ptr(packed), C.c_int64(7), C.c_int64(n), L.ptr(g_feat), gp)
        if ws_r is not None and ws_s is not None and PAIR_BACKWARD and F % 32 == 0 and ns == 2 and sig_p[0].size(0) == 64 and rgb_p[0].size(0) == 64:
            # both heads in one data-gradient pass: d/d feat is written once as the sum of the two
            lean_bit = L.MLP_LEAN if ctx.lean else 0
            rdesc = _mlp_desc(rgb_p, F, L.ENC_AUX_CAT, n_freqs, L.ACT_SIGMOID, freqs, L.MLP_STASHED | lean_bit, ray_ids, stride)
            sdesc = _mlp_desc(sig_p, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None, L.MLP_STASHED | lean_bit)
            base_flags = rdesc.flags                    # (STASHED + the matrix-mode bit + LEAN: the phase bits are OR-ed in, nothing is dropped)
            pair_args = (C.byref(sdesc), L.ptr(feat), L.ptr(table), L.ptr(g_rgbs), L.ptr(g_sigma),
                         C.c_int64(n), gw_r, gb_r, gw_s, gb_s, L.ptr(g_feat), L.ptr(ws_r), C.c_int64(rb), L.ptr(ws_s), C.c_int64(sb))
            scatter_fused = FUSE_SCATTER and kdesc.n_scales == 3 and kdesc.channels == 32 and len(keep) == 9
            if scatter_fused:
                # data gradients of both heads AND the plane scatter in one launch: d loss / d features stays in registers
                def chain_and_weights(flags):
                    rdesc.flags = base_flags | flags
                    L.call("tn_kplanes_mlp_bwd_pair", dev, C.byref(kdesc), L.ptr(packed), C.c_int64(7), gp, C.byref(rdesc), C.byref(sdesc),
                           L.ptr(feat), L.ptr(table), L.ptr(g_rgbs), L.ptr(g_sigma), C.c_int64(n), gw_r, gb_r, gw_s, gb_s, C.c_void_p(None),
                           L.ptr(ws_r), C.c_int64(rb), L.ptr(ws_s), C.c_int64(sb))
                # two calls (chain + scatter kernel, then the weight-gradient kernels): the same launches as one call makes, but the
                # plane gradients are final in between -- with N > 1 their all-reduce starts there and travels under the
                # weight-gradient kernels -- and each half can be timed on its own (bench.py)
                chain_and_weights(L.MLP_CHAIN_ONLY)
                if ctx.planes_ready is not None:
                    ctx.planes_ready(g_planes)
                chain_and_weights(L.MLP_WGRAD_ONLY)
                scattered = True
            elif ctx.planes_ready is not None:
                # N > 1: data gradients -> plane scatter -> hand the finished plane gradients to the caller (it starts their
                # all-reduce) -> weight gradients of the heads, which run while the planes are on the wire
                rdesc.flags = base_flags | L.MLP_CHAIN_ONLY
                L.call("tn_mlp_bwd_pair", dev, C.byref(rdesc), *pair_args)
                scatter()
                scattered = True
                ctx.planes_ready(g_planes)
                rdesc.flags = base_flags | L.MLP_WGRAD_ONLY
                L.call("tn_mlp_bwd_pair", dev, C.byref(rdesc), *pair_args)
            else:
                L.call("tn_mlp_bwd_pair", dev, C.byref(rdesc), *pair_args)
        else:
            if ctx.lean:
                raise RuntimeError("tinynerf_amd: the forward ran with TN_MLP_LEAN (no hidden activations in the workspace) but this backward "
                                   "is not the paired form that rebuilds them")
            stashed = L.MLP_STASHED if ws_r is not None else 0
            rdesc = _mlp_desc(rgb_p, F, L.ENC_AUX_CAT, n_freqs, L.ACT_SIGMOID, freqs, stashed, ray_ids, stride)
            if ws_r is None:
                ws_r, rb = _workspace(rdesc, n, dev, arena, "ws_rgb")
            L.call("tn_mlp_bwd", dev, C.byref(rdesc), L.ptr(feat), L.ptr(table), L.ptr(g_rgbs), C.c_int64(n), gw_r, gb_r, L.ptr(g_feat),
                   L.ptr(ws_r), C.c_int64(rb))
            stashed = L.MLP_STASHED if ws_s is not None else 0
            sdesc = _mlp_desc(sig_p, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None, L.MLP_ACCUM_GRAD_X | stashed)   # g_feat += d sigma / d feat
            if ws_s is None:
                ws_s, sb = _workspace(sdesc, n, dev, arena, "ws_sigma")
            L.call("tn_mlp_bwd", dev, C.byref(sdesc), L.ptr(feat), C.c_void_p(None), L.ptr(g_sigma), C.c_int64(n), gw_s, gb_s,
                   L.ptr(g_feat), L.ptr(ws_s), C.c_int64(sb))
        if not scattered:      # plane scatter
            scatter()
        grads = [None if in_place else g for (g, in_place) in bufs]
        return (None, None, None, None, None, None, None, None, None, None, None, None, None, *grads)


class _MergeLast(Function):
    """(W_c[:, x] W_last, b_c + W_c[:, x] b_last, W_s W_last, b_s + W_s b_last) and the chain rule back to the six original tensors: one
    small launch each way (merge.hip) -- the same five matmuls through torch and autograd were ~35 launches, 0.26 ms per step."""

    @staticmethod
    def forward(ctx: Any, accumulate: bool, pe: int, wc: torch.Tensor, bc: torch.Tensor, ws: torch.Tensor, bs: torch.Tensor,
                w_last: torch.Tensor, b_last: torch.Tensor):  # type: ignore
        dev = L.require_cuda(wc, bc, ws, bs, w_last, b_last)
        ps = [t.contiguous() for t in (wc, bc, ws, bs, w_last, b_last)]
        F = w_last.size(0)
        outs = [torch.empty_like(t) for t in ps[:4]]
        heads = (L.MergeHead * 2)()
        for k, col0 in ((0, pe), (1, 0)):
            heads[k].weight, heads[k].bias = ps[2 * k].data_ptr(), ps[2 * k + 1].data_ptr()
            heads[k].rows, heads[k].ld, heads[k].col0 = ps[2 * k].size(0), ps[2 * k].size(1), col0
            heads[k].out_weight, heads[k].out_bias = outs[2 * k].data_ptr(), outs[2 * k + 1].data_ptr()
        L.call("tn_linear_merge_fwd", dev, C.c_int32(2), heads, L.ptr(ps[4]), L.ptr(ps[5]), C.c_int32(F))
        ctx.save_for_backward(*ps)
        ctx.cfg = (accumulate, pe)
        ctx.refs = (wc, bc, ws, bs, w_last, b_last) if accumulate else None
        return tuple(outs)

    @staticmethod
    def backward(ctx: Any, g_wc: torch.Tensor, g_bc: torch.Tensor, g_ws: torch.Tensor, g_bs: torch.Tensor):  # type: ignore
        ps = ctx.saved_tensors
        accumulate, pe = ctx.cfg
        dev = ps[0].device
        gm = [g.contiguous() if g is not None else torch.zeros_like(p) for g, p in zip((g_wc, g_bc, g_ws, g_bs), ps[:4])]
        refs = ctx.refs if ctx.refs is not None else [None] * 6
        in_place = [r is not None and r.is_leaf and r.grad is not None and r.grad.is_contiguous() and r.grad.dtype == p.dtype
                    for r, p in zip(refs, ps)]
        grads = [r.grad if ip else torch.zeros_like(p) for r, p, ip in zip(refs, ps, in_place)]
        heads = (L.MergeHead * 2)()
        for k, col0 in ((0, pe), (1, 0)):
            heads[k].weight, heads[k].bias = ps[2 * k].data_ptr(), ps[2 * k + 1].data_ptr()
            heads[k].rows, heads[k].ld, heads[k].col0 = ps[2 * k].size(0), ps[2 * k].size(1), col0
            heads[k].out_weight, heads[k].out_bias = grads[2 * k].data_ptr(), grads[2 * k + 1].data_ptr()
            heads[k].grad_merged_weight, heads[k].grad_merged_bias = gm[2 * k].data_ptr(), gm[2 * k + 1].data_ptr()
        L.call("tn_linear_merge_bwd", dev, C.c_int32(2), heads, L.ptr(ps[4]), L.ptr(ps[5]), C.c_int32(ps[4].size(0)), L.ptr(grads[4]), L.ptr(grads[5]))
        return (None, None, *[None if ip else g for g, ip in zip(grads, in_place)])


class _RenderHeads(Function):
    """The part of NerfRenderer.forward behind the feature module (core.py:239-267) for ANY field with the Vanilla decoders
    (Vanilla NeRF, Cobafa): sigma head -> weights scan -> colour head -> composite as one autograd node; ``feat`` is an
    ordinary differentiable input, so the field's own backward (width-256 / 128 stacks, grid scatters) follows through
    autograd.  Replaces the module-by-module path's ``mask.any()`` host sync, boolean gather / index_copy pair and their
    autograd counterparts: every sample goes through the colour head in place -- samples with w == 0 contribute exactly 0 to
    the composite in both directions, as in the reference (core.py:243-249)."""

    @staticmethod
    def forward(ctx: Any, feat: torch.Tensor, packed: torch.Tensor, info: torch.Tensor, bg: Optional[torch.Tensor], thr: float,
                freqs: torch.Tensor, n_freqs: int, n_sigma: int, accumulate: bool, arena: Optional[Arena], train: bool,
                hint: Optional[dict], stats: Optional[dict], link: Optional[dict], *params: torch.Tensor) -> torch.Tensor:  # type: ignore
        sig_p = [p.contiguous() for p in params[:n_sigma]]
        rgb_p = [p.contiguous() for p in params[n_sigma:]]
        feat = feat.contiguous()
        # `link`: row views of the wide stack that produced `feat` (models._FusedMLP.forward, harness only): feat^T as
        # [feature][32-sample] rows in that stack's workspace and the slot where it takes d loss / d feat in the same layout
        offered = link
        if not (link and train and link.get("n") == feat.size(0) and link.get("y_ptr") == feat.data_ptr() and link.get("width") == feat.size(1)
                and feat.size(1) in (128, 256)):
            link = None
        if link is None and offered and offered.get("rows_only"):
            raise RuntimeError("tinynerf_amd: the feature stack left its output as workspace rows only (TN_MLP_ROWS_ONLY) but this render "
                               "node cannot read them")
        ctx.link = link
        dev = L.require_cuda(feat, packed, info, *sig_p, *rgb_p)
        n, R = packed.size(0), info.size(0)
        F = feat.size(1)
        sdesc = _mlp_desc(sig_p, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None)
        if link is not None:
            # cat[PE(d), d] (models.py:87) once per RAY as a table the kernels index through the ray id of every sample
            # (TN_ENC_AUX_CAT, as in the K-Planes node): the colour head then takes the plain-column first layer.  Its weight
            # gradient over a 256-wide x needs the row-operand kernel, hence only with row views
            table, ray_ids, stride, steps = _ray_aux(packed, info, freqs, n_freqs, arena, hint)
            rdesc = _mlp_desc(rgb_p, F, L.ENC_AUX_CAT, n_freqs, L.ACT_SIGMOID, freqs, 0, ray_ids, stride)
        else:
            # cat[PE(d), d] is evaluated per sample inside the colour head's kernels (TN_ENC_DIR_CAT)
            if hint is not None and hint.get("key") == (packed.data_ptr(), n, R):
                steps = hint["steps"]
            else:
                steps = _alloc(arena, "steps", (n,), dev)
                steps.copy_(packed[:, 6])
            table = _alloc(arena, "dirs", (n, 3), dev)
            table.copy_(packed[:, 3:6])
            ray_ids, stride = None, 0
            rdesc = _mlp_desc(rgb_p, F, L.ENC_DIR_CAT, n_freqs, L.ACT_SIGMOID, freqs)
        ws_s = ws_r = None
        sb = rb = 0
        # f16x2 heads read their first-layer operands from the stack's row view (128-byte rows instead of 16 bytes per lane and
        # sample); once a forward has done so the stack stops writing the row-major feat (TN_MLP_ROWS_ONLY / TN_MLP_X_FROM_ROWS)
        rows_fwd = link is not None and MATMUL_F16X2()
        if train:
            ws_s, sb = _workspace(sdesc, n, dev, arena, "ws_sigma")
            ws_r, rb = _workspace(rdesc, n, dev, arena, "ws_rgb")
            if ws_s is None or ws_r is None:
                raise RuntimeError("tinynerf_amd: these decoder shapes are outside the fused render node (use renderer.fused = False)")
        sigma = _alloc(arena, "sigma", (n,), dev)
        rgbs = _alloc(arena, "rgbs", (n, 3), dev)
        if rows_fwd:
            for d in (rdesc, sdesc):
                d.x_rows, d.x_rows_tile_stride = link["y_rows"], link["stride"]
                if link.get("rows_only"):
                    d.flags |= L.MLP_X_FROM_ROWS
        if train:
            L.call("tn_mlp_fwd_stash", dev, C.byref(sdesc), L.ptr(feat), C.c_void_p(None), C.c_int64(n), L.ptr(sigma), L.ptr(ws_s), C.c_int64(sb))
        else:
            L.call("tn_mlp_fwd", dev, C.byref(sdesc), L.ptr(feat), C.c_void_p(None), C.c_int64(n), L.ptr(sigma), C.c_void_p(None))
        weights = _alloc(arena, "weights", (n,), dev)
        covered = hint is not None and hint.get("key") == (packed.data_ptr(), n, R)
        if not covered:
            weights.zero_()          # cuda.cu:84 (zeros_like): samples outside every (start, count) keep weight 0
        gate_slot = _gate_slot(hint, covered and train)       # (see _RenderKPlanes)
        out = torch.empty((R, 3), device=dev)
        if train:          # the colour head runs on every sample: weights and composite of a ray behind it, in one launch
            L.call("tn_mlp_fwd_stash", dev, C.byref(rdesc), L.ptr(feat), L.ptr(table), C.c_int64(n), L.ptr(rgbs), L.ptr(ws_r), C.c_int64(rb))
            L.call("tn_render_rays_fwd", dev, L.ptr(sigma), L.ptr(steps), L.ptr(rgbs), L.ptr(info), L.ptr(bg), C.c_float(thr), L.ptr(weights),
                   L.ptr(out), L.ptr(gate_slot), C.c_int64(n), C.c_int64(R))
        else:              # inference: the colour head is only evaluated where the weight is not 0 (core.py:246-251), tile-wise
            L.call("tn_weights_fwd", dev, L.ptr(sigma), L.ptr(steps), L.ptr(info), C.c_float(thr), L.ptr(weights), C.c_int64(n), C.c_int64(R))
            rdesc.row_gate = weights.data_ptr()
            L.call("tn_mlp_fwd", dev, C.byref(rdesc), L.ptr(feat), L.ptr(table), C.c_int64(n), L.ptr(rgbs), C.c_void_p(None))
            rdesc.row_gate = None
            L.call("tn_composite_fwd", dev, L.ptr(rgbs), L.ptr(weights), L.ptr(info), L.ptr(bg), L.ptr(out), C.c_void_p(None),
                   C.c_int64(n), C.c_int64(R))
        ctx.gate_in_slot, ctx.stats = gate_slot is not None, stats
        ctx.gate = gate_slot if gate_slot is not None else (weights.amax().reshape(1) if train else None)   # "Empty iteration", see _RenderKPlanes
        if stats is not None:
            stats["gate"] = ctx.gate
            stats["pre_gated"] = ctx.gate_in_slot
            stats["upstream_gated"] = False
        ctx.save_for_backward(feat, info, bg, freqs, sigma, steps, table, ray_ids, weights, rgbs, ws_s, ws_r, *params)
        ctx.cfg = (n_freqs, n_sigma, accumulate, stride, sb, rb, covered)
        ctx.arena = arena
        ctx.param_refs = params if accumulate else None
        return out

    @staticmethod
    def backward(ctx: Any, grad_out: torch.Tensor):  # type: ignore
        feat, info, bg, freqs, sigma, steps, table, ray_ids, weights, rgbs, ws_s, ws_r, *params = ctx.saved_tensors
        n_freqs, n_sigma, accumulate, stride, sb, rb, covered = ctx.cfg
        sig_p = [p.contiguous() for p in params[:n_sigma]]
        rgb_p = [p.contiguous() for p in params[n_sigma:]]
        dev = feat.device
        n, R, F = feat.size(0), info.size(0), feat.size(1)
        g_out = grad_out.contiguous()
        if ctx.gate is not None and not _upstream_is_gated(ctx):
            g_out = g_out * (ctx.gate > 0).to(g_out.dtype)
        refs: Sequence[Optional[torch.Tensor]] = ctx.param_refs if ctx.param_refs is not None else [None] * len(params)

        def grad_buffer(p: torch.Tensor, ref: Optional[torch.Tensor]):
            if accumulate and ref is not None and ref.is_leaf and ref.grad is not None and ref.grad.stride() == p.stride():
                return ref.grad, True
            return torch.zeros_like(p), False
        bufs = [grad_buffer(p, r) for p, r in zip(params, refs)]
        g_sig = [b[0] for b in bufs[:n_sigma]]
        g_rgb = [b[0] for b in bufs[n_sigma:]]
        arena = ctx.arena
        g_rgbs = _alloc(arena, "g_rgbs", (n, 3), dev)
        g_sigma = _alloc(arena, "g_sigma", (n,), dev)
        if not covered:              # (the kernel writes every sample a ray owns)
            g_rgbs.zero_(); g_sigma.zero_()
        L.call("tn_render_rays_bwd", dev, L.ptr(sigma), L.ptr(steps), L.ptr(rgbs), L.ptr(info), L.ptr(bg), L.ptr(weights), L.ptr(g_out),
               L.ptr(g_rgbs), L.ptr(g_sigma), C.c_int64(n), C.c_int64(R))
        link = ctx.link
        # with row views the heads write d loss / d feat straight into the feature stack's workspace (rows) and read feat^T from
        # there for their first layers' weight gradients; autograd gets a placeholder of the right shape (no memory behind it)
        g_feat = _empty_rows(n, F, dev) if link is None else None      # handed to autograd (the field's backward): not an arena view
        nr, ns = len(rgb_p) // 2, len(sig_p) // 2
        gw_r = (C.c_void_p * nr)(*[g.data_ptr() for g in g_rgb[0::2]])
        gb_r = (C.c_void_p * nr)(*[g.data_ptr() for g in g_rgb[1::2]])
        gw_s = (C.c_void_p * ns)(*[g.data_ptr() for g in g_sig[0::2]])
        gb_s = (C.c_void_p * ns)(*[g.data_ptr() for g in g_sig[1::2]])
        if link is not None:
            rdesc = _mlp_desc(rgb_p, F, L.ENC_AUX_CAT, n_freqs, L.ACT_SIGMOID, freqs, L.MLP_STASHED, ray_ids, stride)
        else:
            rdesc = _mlp_desc(rgb_p, F, L.ENC_DIR_CAT, n_freqs, L.ACT_SIGMOID, freqs, L.MLP_STASHED)
        sdesc = _mlp_desc(sig_p, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None, L.MLP_ACCUM_GRAD_X | L.MLP_STASHED)     # g_feat += d sigma / d feat
        if link is not None:
            for d in (rdesc, sdesc):
                d.x_rows, d.grad_x_rows = link["y_rows"], link["grad_rows"]
                d.x_rows_tile_stride = d.grad_x_rows_tile_stride = link["stride"]
                if link.get("skipped_last"):     # x is the producer's last HIDDEN activation: d / d (its pre-activation) = relu' * ...
                    d.grad_x_mask_rows, d.grad_x_mask_tile_stride = link["mask_rows"], link["stride"]
        if link is not None and HEADS_PAIR_BACKWARD and F % 64 == 0 and ns == 2 and nr == 5 and sig_p[0].size(0) == 64 and rgb_p[0].size(0) == 64:
            # one call for both heads: their first layers' x-column weight gradients share a launch (and the x rows), and behind the
            # 128-wide stack d loss / d feat is written once as the sum of the two data gradients (mlp_bwd2.hip, bwd_pair_common)
            sdesc.flags &= ~L.MLP_ACCUM_GRAD_X
            L.call("tn_mlp_bwd_pair", dev, C.byref(rdesc), C.byref(sdesc), L.ptr(feat), L.ptr(table), L.ptr(g_rgbs), L.ptr(g_sigma),
                   C.c_int64(n), gw_r, gb_r, gw_s, gb_s, L.ptr(g_feat), L.ptr(ws_r), C.c_int64(rb), L.ptr(ws_s), C.c_int64(sb))
        else:
            L.call("tn_mlp_bwd", dev, C.byref(rdesc), L.ptr(feat), L.ptr(table), L.ptr(g_rgbs), C.c_int64(n), gw_r, gb_r, L.ptr(g_feat),
                   L.ptr(ws_r), C.c_int64(rb))
            L.call("tn_mlp_bwd", dev, C.byref(sdesc), L.ptr(feat), C.c_void_p(None), L.ptr(g_sigma), C.c_int64(n), gw_s, gb_s, L.ptr(g_feat),
                   L.ptr(ws_s), C.c_int64(sb))
        if link is not None:
            link["delivered"] = True
            g_feat = torch.empty(1, device=dev).expand(n, F)
        grads = [None if in_place else g for (g, in_place) in bufs]
        return (g_feat, None, None, None, None, None, None, None, None, None, None, None, None, None, *grads)


def _mergeable(producer, sig_p, rgb_p) -> bool:
    """the producer stack's last layer is a square Linear without activation feeding nothing but the two heads' first layers"""
    ps = producer.params()
    if len(ps) < 6 or ps[-2].dim() != 2 or ps[-2].size(0) != ps[-2].size(1) or ps[-2].size(0) not in (128, 256):
        return False
    F = ps[-2].size(0)
    if not (sig_p[0].size(1) == F and rgb_p[0].size(1) > F and all(p.requires_grad for p in (ps[-2], ps[-1], sig_p[0], sig_p[1], rgb_p[0], rgb_p[1]))):
        return False
    # ... and the kernel side can actually stop this stack at its last hidden activation (slab-eligible f16x2 stack: positional encoding with
    # <= 64 slots or <= 64 plain inputs, out == H): asked of the library itself (host-only, no launch) with the descriptor of the
    # stack's previous forward -- shapes alone would arm TN_MLP_SKIP_LAST for stacks that then fail with TN_E_CONFIG mid-step
    import ctypes as C
    from . import _lib as L
    from .models import _mlp_desc
    sc = producer.__dict__.get("scratch")
    cfg = sc[4].get("last_cfg") if sc is not None and len(sc) > 4 else None
    if cfg is None:
        return False
    in_dim, encoding, n_freqs, out_act = cfg
    desc = _mlp_desc([p.contiguous() for p in ps], in_dim, encoding, n_freqs, out_act, None, flags=L.MLP_SKIP_LAST | L.MLP_ROWS_ONLY)
    a, b, c, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
    return L.lib().tn_mlp_rows_view_hidden(C.byref(desc), C.c_int64(32), C.byref(a), C.byref(b), C.byref(c), C.byref(d)) == 0


def _vanilla_decoders(renderer) -> bool:
    from .models import VanillaColorDecoder, VanillaOpacityDecoder
    sd, cd = renderer.sigma_decoder, renderer.rgb_decoder
    return type(sd) is VanillaOpacityDecoder and type(cd) is VanillaColorDecoder and sd.net.net[0].out_features == 64 and \
        cd.net.net[0].out_features == 64 and len(cd.net.params()) == 10 and sd.net.net[0].in_features % 4 == 0


def supports(renderer) -> bool:
    """K-Planes field (whole render as one node, gather / scatter inside the MLP launches) or any other field in front of the
    Vanilla decoders of run.py:133-134,138-139,149-150 (heads + scan + composite as one node)."""
    from .models import KPlanesFeatureField
    fm = renderer.feature_module
    if not _vanilla_decoders(renderer):
        return False
    if isinstance(fm, KPlanesFeatureField):
        return fm.dropout.p == 0.0
    return True


def render(renderer, packed: torch.Tensor, info: torch.Tensor, thr: float, accumulate_into_grad: bool = False) -> torch.Tensor:
    """Fused forward of ``renderer`` on packed samples (see supports())."""
    from .models import KPlanesFeatureField
    fm, sd, cd = renderer.feature_module, renderer.sigma_decoder, renderer.rgb_decoder
    sig_p, rgb_p = sd.net.params(), cd.net.params()
    bg = renderer._bg(packed.device)
    arena = None
    if getattr(renderer, "reuse_buffers", False):
        arena = renderer.__dict__.setdefault("_arena", Arena())
    hint, stats = getattr(renderer, "_batch_aux", None), renderer.__dict__.setdefault("_stats", {})
    if isinstance(fm, KPlanesFeatureField):
        planes = fm.plane_tensors()
        train = torch.is_grad_enabled() and any(p.requires_grad for p in (*planes, *sig_p, *rgb_p))
        return _RenderKPlanes.apply(packed.contiguous(), info.contiguous(), bg, float(thr), cd.pe.freqs, cd.n_freqs, len(planes),
                                    len(sig_p), accumulate_into_grad, arena, train, hint, stats, *planes, *sig_p, *rgb_p)
    # harness: the stack whose row view this node matched last time may leave its output as rows only (TN_MLP_ROWS_ONLY) -- armed for
    # exactly this forward; _RenderHeads fails loudly if it then cannot read the rows
    producer = renderer.__dict__.get("_rows_producer")
    armed = None
    if producer is not None and arena is not None and torch.is_grad_enabled() and MATMUL_F16X2():
        sc = producer.__dict__.get("scratch")
        if sc is not None and len(sc) > 4:
            sc[4]["rows_only"] = True
            armed = sc[4]
            if MERGE_LAST and _mergeable(producer, sig_p, rgb_p):
                sc[4]["skip_last"] = True
    try:
        feat = fm(packed[:, :3])
    finally:
        # the flag is for exactly THIS forward: if the producer stack did not run (an exception, a feature module that skipped it) it must
        # not stay armed for an unrelated forward, whose row-major output would then silently stay unwritten
        if armed is not None:
            armed.pop("skip_last", None)
        if armed is not None and armed.pop("rows_only", False):
            import warnings
            warnings.warn("tinynerf_amd.fused: TN_MLP_ROWS_ONLY was armed but the feature stack did not consume it; disarmed", RuntimeWarning)
    train = torch.is_grad_enabled() and (feat.requires_grad or any(p.requires_grad for p in (*sig_p, *rgb_p)))
    link = None
    if train and arena is not None:          # harness: the stack that produced `feat` may offer row views of its workspace
        from .models import MLP
        for mod in fm.modules():
            sc = mod.__dict__.get("scratch") if isinstance(mod, MLP) else None
            if sc is not None and len(sc) > 2 and sc[2].get("y_ptr") == feat.data_ptr():
                link = sc[2]
                if feat.size(1) in (128, 256) and sc[2].get("n") == feat.size(0):
                    renderer.__dict__["_rows_producer"] = mod
    if link is not None and link.get("skipped_last"):
        # the stack stopped at its last hidden activation h (rows): heads on h with the last layer folded into their first layers
        w_last, b_last = producer.params()[-2:]
        pe = rgb_p[0].size(1) - feat.size(1)                        # torch column order of the colour head: [PE(d), d, x] (models.py:87)
        wc, bc, ws_, bs_ = _MergeLast.apply(accumulate_into_grad, pe, rgb_p[0], rgb_p[1], sig_p[0], sig_p[1], w_last, b_last)
        rgb_p = [wc, bc, *rgb_p[2:]]
        sig_p = [ws_, bs_, *sig_p[2:]]
    return _RenderHeads.apply(feat, packed.contiguous(), info.contiguous(), bg, float(thr), cd.pe.freqs, cd.n_freqs, len(sig_p),
                              accumulate_into_grad, arena, train, hint, stats, link, *sig_p, *rgb_p)
