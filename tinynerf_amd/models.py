"""Field models with the API of the reference's ``src/models.py``, evaluated by HIP kernels.

Class names, constructor signatures, sub-module names and ``state_dict`` keys/shapes are the
reference's (a reference ``model.pt`` loads unchanged); ``forward`` of every head is
ONE launch of the fused fp32-MFMA MLP kernel (``csrc/mlp.hip``) with its encoding and output
activation fused, and the K-Planes field is one gather kernel (``csrc/kplanes.hip``).

Reference map: MLP models.py:7-28, PositionalEncoding :30-39, TruncatedExponential :42-55,
Vanilla* :59-89, KPlanes* :93-205, Cobafa* :209-266.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
from typing import Any, Callable, List, Optional, Sequence, Tuple, cast

import torch
from torch.autograd import Function

from . import _lib as L


# --------------------------------------------------------------------------------------------
# fused MLP plumbing
# --------------------------------------------------------------------------------------------
# Matrix products of the wide stacks (Vanilla 256 x 10, Cobafa 128 x 6): "bf16x3" = bf16 matrix cores with exact three-way
# operand splits (TN_MLP_BF16X3: results equal to fp32 rounding, 2.67 x the fp32 matrix rate), "fp32" = v_mfma_f32_32x32x2_f32.
# TN_MATMUL=fp32 / bf16x3 in the environment (or assigning here) selects them; the default is
# "f16x2" (TN_MLP_F16X2, round 4) = all three passes of the hidden layers on the fp16 matrix cores with two-term operand splits and
# power-of-two scales (three products instead of six; csrc/mlp_f2_layers.hip).
from .config import CONFIG as _CONFIG      # noqa: E402
MATMUL = _CONFIG.matmul


def _mlp_desc(params: Sequence[torch.Tensor], in_dim: int, encoding: int, n_freqs: int, out_act: int,
              freqs: Optional[torch.Tensor], flags: int = 0, aux_index: Optional[torch.Tensor] = None, aux_stride: int = 0) -> L.MlpDesc:
    n_layers = len(params) // 2
    if n_layers > L.TN_MLP_MAX_LAYERS:
        raise RuntimeError(f"MLP with {n_layers} layers exceeds the kernel limit {L.TN_MLP_MAX_LAYERS}")
    d = L.MlpDesc()
    d.n_layers = n_layers
    d.in_dim = in_dim
    d.encoding = encoding
    d.n_freqs = n_freqs
    d.out_activation = out_act
    d.flags = flags | {"bf16x3": L.MLP_BF16X3, "f16x2": L.MLP_F16X2}.get(MATMUL, 0)
    d.aux_index = aux_index.data_ptr() if aux_index is not None else None
    d.aux_stride = aux_stride
    d.freqs = freqs.data_ptr() if freqs is not None else None
    d.dims[0] = params[0].size(1)
    for l in range(n_layers):
        w, b = params[2 * l], params[2 * l + 1]
        d.dims[l + 1] = w.size(0)
        d.weights[l] = w.data_ptr()
        d.biases[l] = b.data_ptr()
    return d


def _bucket(n: int) -> int:
    """n rounded up to 1/8-octave steps (<= 12.5 % more).  Dynamic batches move N by a few percent every step; a workspace
    sized to N exactly (10 KB per sample for the Vanilla stack: 11 GB) gives the caching allocator a new size at every record
    high of N, each answered by a fresh hipMalloc in the middle of the step (measured: 240 ms stalls, 70 GiB reserved after 15
    steps) while the smaller blocks stay cached.  Bucketed sizes are re-used."""
    if n <= 64:
        return 64
    g = 1 << max(n.bit_length() - 4, 0)
    return (n + g - 1) // g * g


def _empty_rows(n: int, cols: int, dev: torch.device) -> torch.Tensor:
    """[n, cols] fp32 carved from a bucketed allocation: per-sample tensors of the wide stacks are ~1 GB each and N changes every
    step -- an exact-size request is a new block size for the caching allocator nearly every time (hipMalloc inside the step)"""
    return torch.empty((_bucket(n), cols), device=dev)[:n]


class _FusedMLP(Function):
    """y = act(MLP(enc(x, aux))) in one launch; in training the forward also writes the activation workspace the backward
    consumes (tn_mlp_fwd_stash), otherwise the backward recomputes the hidden activations."""

    stash_forward = True   # training forward writes the activation workspace (tn_mlp_fwd_stash); False: backward recomputes
    layerwise_inference = False   # True: tn_mlp_fwd_ws one launch per layer (TN_MLP_LAYERWISE) instead of the cross-layer persistent launch --
                                  # the parity partner of tests/test_hip_fused.py and the A side of scripts/fused_fwd_time.py
    layerwise_training = False    # ... the same for the training forward (tn_mlp_fwd_stash) and the data-gradient chain (tn_mlp_bwd)

    @staticmethod
    def forward(ctx: Any, x: torch.Tensor, aux: Optional[torch.Tensor], freqs: Optional[torch.Tensor], encoding: int,
                n_freqs: int, out_act: int, recording: bool, scratch: Optional[Tuple[Any, str]], *params: torch.Tensor) -> torch.Tensor:  # type: ignore
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.size(-1)).to(torch.float32).contiguous()
        aux2 = None if aux is None else aux.reshape(-1, aux.size(-1)).to(torch.float32).contiguous()
        ps = [p.contiguous() for p in params]
        dev = L.require_cuda(x2, aux2, *ps)
        n = x2.size(0)
        desc = _mlp_desc(ps, x2.size(1), encoding, n_freqs, out_act, freqs)
        if scratch is not None and len(scratch) > 4:
            scratch[4]["last_cfg"] = (x2.size(1), encoding, n_freqs, out_act)     # (fused._mergeable probes the library with it)
        y = _empty_rows(n, ps[-1].numel(), dev)
        # harness: fused.render_heads arms scratch[4]["rows_only"] for ONE forward when the render node it is about to run reads
        # this stack's output from the workspace rows (it has matched the row-view link before): the row-major y is then
        # allocated but never written
        rows_only = bool(scratch is not None and len(scratch) > 4 and scratch[4].pop("rows_only", False) and MATMUL == "f16x2")
        # ... and scratch[4]["skip_last"] when it has merged this stack's last (plain Linear) layer into the heads' first layers
        # (TN_MLP_SKIP_LAST): the stack then stops at its last hidden activation and offers THAT as rows
        skip_last = bool(scratch is not None and len(scratch) > 4 and scratch[4].pop("skip_last", False) and rows_only)
        # training: the forward writes the activations straight into the backward's workspace (nothing is recomputed)
        ws, ws_bytes = None, 0
        # `recording` = torch.is_grad_enabled() at the call site: inside torch.no_grad() (infer(), the occupancy refresh)
        # needs_input_grad still reports the parameters' flags, and inside Function.forward grad mode is always off
        if _FusedMLP.stash_forward and recording and any(ctx.needs_input_grad) and n > 0:
            wsfn = L.lib().tn_mlp_bwd_workspace_bytes
            wsfn.restype = C.c_int64
            ws_bytes = int(wsfn(C.byref(desc), C.c_int64(_bucket(n))))
        if ws_bytes:
            # harness: a capacity-based arena (arena.Arena) instead of the caching allocator -- the training loop runs each
            # forward's backward before the next forward of the module, so one buffer per module is enough
            ws = scratch[0].get(scratch[1], (ws_bytes // 4,), dev) if scratch is not None else torch.empty(ws_bytes // 4, device=dev)
            link = scratch[2] if scratch is not None and len(scratch) > 2 else None
            if _FusedMLP.layerwise_training:
                desc.flags |= L.MLP_LAYERWISE
            if rows_only and link is not None:
                desc.flags |= L.MLP_ROWS_ONLY
                if skip_last:
                    desc.flags |= L.MLP_SKIP_LAST
            L.call("tn_mlp_fwd_stash", dev, C.byref(desc), L.ptr(x2), L.ptr(aux2), C.c_int64(n), L.ptr(y), L.ptr(ws), C.c_int64(ws_bytes))
            if link is not None:
                # harness: wide stacks evaluated layer by layer keep y as [feature][32-sample] rows in their workspace and take
                # d loss / d y in that layout (tn_mlp_rows_view): the render node behind this stack (fused._RenderHeads) reads
                # / writes them there instead of going through row-major [n, 256] tensors
                link.clear()
                y_off, g_off, stride, m_off = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
                if desc.flags & L.MLP_SKIP_LAST:
                    L.call_plain("tn_mlp_rows_view_hidden", C.byref(desc), C.c_int64(n), C.byref(y_off), C.byref(g_off), C.byref(m_off), C.byref(stride))
                    link.update(ws=ws, y_rows=ws.data_ptr() + 4 * y_off.value, grad_rows=ws.data_ptr() + 4 * g_off.value,
                                mask_rows=ws.data_ptr() + 4 * m_off.value, stride=stride.value, n=n, width=y.size(1), y_ptr=y.data_ptr(),
                                delivered=False, rows_only=True, skipped_last=True)
                elif L.lib().tn_mlp_rows_view(C.byref(desc), C.c_int64(n), C.byref(y_off), C.byref(g_off), C.byref(stride)) == 0:
                    link.update(ws=ws, y_rows=ws.data_ptr() + 4 * y_off.value, grad_rows=ws.data_ptr() + 4 * g_off.value,
                                stride=stride.value, n=n, width=y.size(1), y_ptr=y.data_ptr(), delivered=False,
                                rows_only=bool(desc.flags & L.MLP_ROWS_ONLY))
                elif desc.flags & L.MLP_ROWS_ONLY:
                    raise RuntimeError("tinynerf_amd: a stack without row views ran with TN_MLP_ROWS_ONLY")
            ctx.link = link
        else:
            # inference (infer(), the occupancy refresh): wide stacks run through the layer kernels with two ping-pong row
            # buffers as scratch (tn_mlp_fwd_ws), in chunks of <= 2^22 samples so that the scratch stays below 10 GB
            fwd_ws = L.lib().tn_mlp_fwd_workspace_bytes
            fwd_ws.restype = C.c_int64
            chunk = 1 << 22
            if _FusedMLP.layerwise_inference:
                desc.flags |= L.MLP_LAYERWISE
            nbytes = int(fwd_ws(C.byref(desc), C.c_int64(min(n, chunk)))) if (n > 0 and aux2 is None) else 0
            if nbytes:
                wsi = torch.empty(nbytes // 4, device=dev)
                for k in range(0, n, chunk):
                    m_ = min(chunk, n - k)
                    L.call("tn_mlp_fwd_ws", dev, C.byref(desc), L.ptr(x2[k:k + m_]), C.c_void_p(None), C.c_int64(m_), L.ptr(y[k:k + m_]),
                           L.ptr(wsi), C.c_int64(nbytes))
            else:
                L.call("tn_mlp_fwd", dev, C.byref(desc), L.ptr(x2), L.ptr(aux2), C.c_int64(n), L.ptr(y), C.c_void_p(None))
        if not hasattr(ctx, "link"):
            ctx.link = None
        # harness (scratch[3]): weight gradients are added straight into param.grad where it exists (the optimizer pass zeroes it)
        ctx.param_refs = params if (scratch is not None and len(scratch) > 3 and scratch[3]) else None
        ctx.save_for_backward(x2, aux2, freqs, ws, *ps)
        ctx.skip_last = bool(ws_bytes and (desc.flags & L.MLP_SKIP_LAST))
        ctx.cfg = (encoding, n_freqs, out_act)
        ctx.x_shape = x.shape
        return y.reshape(*lead, y.size(-1))

    @staticmethod
    def backward(ctx: Any, grad_y: torch.Tensor):  # type: ignore
        x2, aux2, freqs, ws_fwd, *ps = ctx.saved_tensors
        encoding, n_freqs, out_act = ctx.cfg
        dev = x2.device
        n = x2.size(0)
        # d loss / d y already deposited as rows in the workspace by the consumer (see forward): grad_y is a placeholder
        delivered = ctx.link is not None and ctx.link.get("delivered") and ctx.link.get("ws") is ws_fwd
        if delivered:
            ctx.link["delivered"] = False
        gy = None if delivered else grad_y.reshape(n, -1).to(torch.float32).contiguous()
        desc = _mlp_desc(ps, x2.size(1), encoding, n_freqs, out_act, freqs,
                         (L.MLP_STASHED if ws_fwd is not None else 0) | (L.MLP_GRAD_Y_ROWS if delivered else 0) |
                         (L.MLP_SKIP_LAST if ctx.skip_last else 0) | (L.MLP_LAYERWISE if _FusedMLP.layerwise_training else 0))
        if ctx.skip_last and not delivered:
            raise RuntimeError("tinynerf_amd: this stack ran without its last layer (TN_MLP_SKIP_LAST) but its consumer did not deliver "
                               "d loss / d (hidden activation) as workspace rows")
        refs = ctx.param_refs if ctx.param_refs is not None else [None] * len(ps)
        in_place = [r is not None and r.grad is not None and r.grad.stride() == p.stride() and r.grad.dtype == p.dtype for r, p in zip(refs, ps)]
        grads = [r.grad if ip else torch.zeros_like(p) for r, p, ip in zip(refs, ps, in_place)]
        n_layers = len(ps) // 2
        gw = (C.c_void_p * n_layers)(*[g.data_ptr() for g in grads[0::2]])
        gb = (C.c_void_p * n_layers)(*[g.data_ptr() for g in grads[1::2]])
        want_gx = ctx.needs_input_grad[0] and encoding != L.ENC_POSENC
        gx = _empty_rows(n, x2.size(1), dev) if want_gx else None
        wsfn = L.lib().tn_mlp_bwd_workspace_bytes
        wsfn.restype = C.c_int64
        ws_bytes = int(wsfn(C.byref(desc), C.c_int64(_bucket(n))))
        ws = ws_fwd if ws_fwd is not None else (torch.empty(ws_bytes // 4, device=dev) if ws_bytes else None)
        L.call("tn_mlp_bwd", dev, C.byref(desc), L.ptr(x2), L.ptr(aux2), L.ptr(gy), C.c_int64(n), gw, gb, L.ptr(gx),
               L.ptr(ws), C.c_int64(ws_bytes))
        gx_out = gx.reshape(ctx.x_shape) if gx is not None else None
        out = [None if ip else g for g, ip in zip(grads, in_place)]
        if ctx.skip_last:                # the last layer's gradients arrive through the merged parameters (fused.render), not from here
            out[-2:] = [None, None]
        return (gx_out, None, None, None, None, None, None, None, *out)


def _linear_params(net: torch.nn.Sequential) -> List[torch.Tensor]:
    out: List[torch.Tensor] = []
    for m in net.modules():
        if isinstance(m, torch.nn.Linear):
            out += [m.weight, m.bias]
    return out


class MLP(torch.nn.Module):
    """Linear(in,h) ReLU [Linear(h,h) ReLU]*L Linear(h,out) -- models.py:7-28."""

    def __init__(
        self,
        in_features: int,
        hidden_features: int,
        hidden_layers: int,
        out_features: int | None = None,
        activation: Callable = torch.nn.ReLU,
    ):
        super().__init__()
        if activation is not torch.nn.ReLU:
            raise NotImplementedError("the fused MLP kernel implements ReLU hidden activations (the only one the reference uses)")
        out_features = out_features if out_features is not None else hidden_features
        self.net = torch.nn.Sequential(
            torch.nn.Linear(in_features, hidden_features),
            activation(),
            *[torch.nn.Sequential(
                torch.nn.Linear(hidden_features, hidden_features),
                activation()
            ) for _ in range(hidden_layers)],
            torch.nn.Linear(hidden_features, out_features)
        )

    def params(self) -> List[torch.Tensor]:
        return _linear_params(self.net)

    def fused(self, x: torch.Tensor, aux: Optional[torch.Tensor] = None, encoding: int = L.ENC_NONE, n_freqs: int = 0,
              out_act: int = L.ACT_NONE, freqs: Optional[torch.Tensor] = None) -> torch.Tensor:
        # `scratch` = (arena.Arena, buffer name, row-view link dict), set by the training harness (run.Trainer); None: allocate per call
        return _FusedMLP.apply(x, aux, freqs, encoding, n_freqs, out_act, torch.is_grad_enabled(), self.__dict__.get("scratch"),
                               *self.params())

    def forward(self, x: torch.Tensor):
        return self.fused(x)


class PositionalEncoding(torch.nn.Module):
    """[sin(x f_j), cos(x f_j)] per coordinate, f_j = 2^j pi -- models.py:30-39."""

    def __init__(self, n_freqs: int):
        super().__init__()
        self.freqs: torch.Tensor
        self.register_buffer("freqs", 2 ** torch.arange(0, n_freqs) * torch.pi)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x2 = x.reshape(-1, x.size(-1)).to(torch.float32).contiguous()
        fr = self.freqs.to(torch.float32).contiguous()
        dev = L.require_cuda(x2, fr)
        F = fr.numel()
        out = torch.empty((x2.size(0), x2.size(1) * 2 * F), device=dev)
        L.call("tn_posenc_fwd", dev, L.ptr(x2), C.c_int64(x2.size(0)), C.c_int(x2.size(1)), L.ptr(fr), C.c_int(F), L.ptr(out))
        return out.reshape(*x.shape[:-1], out.size(-1))


class _BasisDot(Function):
    """out[n,k] = act(sum_c f[n,c] * basis[n,k,c]) -- the product stage of the explicit K-Planes decoders (models.py:183-205)
    and, with C = K = 1 and f = 1, a plain element-wise activation (tn_basis_dot_fwd / _bwd)."""

    @staticmethod
    def forward(ctx: Any, f: torch.Tensor, basis: torch.Tensor, n_out: int, act: int) -> torch.Tensor:  # type: ignore
        f2 = f.reshape(-1, f.size(-1)).to(torch.float32).contiguous()
        b2 = basis.reshape(f2.size(0), -1).to(torch.float32).contiguous()
        dev = L.require_cuda(f2, b2)
        n, Cc = f2.shape
        if b2.size(1) != n_out * Cc:
            raise RuntimeError(f"basis must hold {n_out} x {Cc} values per sample")
        out = torch.empty((n, n_out), device=dev)
        L.call("tn_basis_dot_fwd", dev, L.ptr(f2), L.ptr(b2), C.c_int64(n), C.c_int32(Cc), C.c_int32(n_out), C.c_int32(act), L.ptr(out))
        ctx.save_for_backward(f2, b2)
        ctx.cfg = (n_out, act, f.shape, basis.shape)
        return out.reshape(*f.shape[:-1], n_out)

    @staticmethod
    def backward(ctx: Any, g: torch.Tensor):  # type: ignore
        f2, b2 = ctx.saved_tensors
        n_out, act, f_shape, b_shape = ctx.cfg
        n, Cc = f2.shape
        g = g.reshape(n, n_out).to(torch.float32).contiguous()
        g_f = torch.empty_like(f2)
        g_b = torch.empty_like(b2) if ctx.needs_input_grad[1] else None
        L.call("tn_basis_dot_bwd", f2.device, L.ptr(f2), L.ptr(b2), L.ptr(g), C.c_int64(n), C.c_int32(Cc), C.c_int32(n_out), C.c_int32(act),
               L.ptr(g_f), L.ptr(g_b), C.c_int32(0))
        return g_f.reshape(f_shape) if ctx.needs_input_grad[0] else None, None if g_b is None else g_b.reshape(b_shape), None, None


class TruncatedExponential(Function):  # pylint: disable=abstract-method
    """exp with a clamped backward (models.py:42-53): forward exp(x), backward g * exp(clamp(x, -15, 15)).  Inside the heads it
    is fused into the MLP kernel (TN_ACT_EXP_M1); on its own it is the element-wise case of tn_basis_dot_* (C = K = 1, f = 1,
    TN_ACT_EXP: x reaches expf() unmodified, as in torch.exp)."""

    @staticmethod
    def forward(ctx, x):  # pylint: disable=arguments-differ
        x = x.float().contiguous()
        dev = L.require_cuda(x)
        v = x.reshape(-1, 1)
        one = torch.ones((1, 1), device=dev).expand(v.size(0), 1).contiguous() if v.size(0) else v
        out = torch.empty_like(v)
        L.call("tn_basis_dot_fwd", dev, L.ptr(one), L.ptr(v), C.c_int64(v.size(0)), C.c_int32(1), C.c_int32(1), C.c_int32(L.ACT_EXP), L.ptr(out))
        ctx.save_for_backward(one, v)
        ctx.x_shape = x.shape
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):  # pylint: disable=arguments-differ
        one, v = ctx.saved_tensors
        g = g.float().contiguous().reshape(-1, 1)
        g_one, g_v = torch.empty_like(one), torch.empty_like(v)
        L.call("tn_basis_dot_bwd", v.device, L.ptr(one), L.ptr(v), L.ptr(g), C.c_int64(v.size(0)), C.c_int32(1), C.c_int32(1),
               C.c_int32(L.ACT_EXP), L.ptr(g_one), L.ptr(g_v), C.c_int32(0))
        return g_v.reshape(ctx.x_shape)


truncated_exp: Callable = TruncatedExponential.apply


# --------------------------------------------------------------------------------------------
# Vanilla NeRF (models.py:59-89)
# --------------------------------------------------------------------------------------------
class VanillaFeatureMLP(torch.nn.Module):
    def __init__(self, n_freqs: int, hidden_features: int, hidden_layers: int):
        super().__init__()
        in_features = n_freqs * 2 * 3
        self.encoding = PositionalEncoding(n_freqs=n_freqs)
        self.net = MLP(in_features, hidden_features, hidden_layers)
        self.feature_dim = hidden_features
        self.n_freqs = n_freqs

    def forward(self, x):
        # PE fused into the first layer of the MLP launch
        return self.net.fused(x, None, L.ENC_POSENC, self.n_freqs, L.ACT_NONE, self.encoding.freqs)


class VanillaOpacityDecoder(torch.nn.Module):
    def __init__(self, feature_dim):
        super().__init__()
        self.net = MLP(feature_dim, 64, 0, 1)
        self.activation = lambda x: truncated_exp(x - 1.)

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        # exp(y - 1) fused as the output activation (gradient clamp of truncated_exp kept in the backward kernel)
        return self.net.fused(features, None, L.ENC_NONE, 0, L.ACT_EXP_M1)


class VanillaColorDecoder(torch.nn.Module):
    def __init__(self, n_freqs: int, in_features: int, hidden_features: int, hidden_layers: int):
        super().__init__()
        self.pe = PositionalEncoding(n_freqs)
        total_features = in_features + n_freqs * 2 * 3 + 3
        self.net = MLP(total_features, hidden_features, hidden_layers, 3)
        self.activation = torch.nn.Sigmoid()
        self.n_freqs = n_freqs

    def forward(self, features: torch.Tensor, rays_d: torch.Tensor) -> torch.Tensor:
        # cat[PE(d), d, features] is formed in registers; sigmoid fused
        return self.net.fused(features, rays_d, L.ENC_DIR_CAT, self.n_freqs, L.ACT_SIGMOID, self.pe.freqs)


# --------------------------------------------------------------------------------------------
# K-Planes (models.py:93-205)
# --------------------------------------------------------------------------------------------
def _hwc(plane: torch.Tensor) -> torch.Tensor:
    """[1,C,H,W] channels_last parameter -> the [H,W,C] memory the kernels index (no copy)."""
    p = plane if plane.is_contiguous(memory_format=torch.channels_last) else plane.contiguous(memory_format=torch.channels_last)
    return p.permute(0, 2, 3, 1)[0]


def _kplanes_desc(planes: Sequence[Optional[torch.Tensor]]) -> Tuple[L.KPlanesDesc, List[torch.Tensor]]:
    n_scales = len(planes) // 3
    if n_scales > L.TN_KPLANES_MAX_SCALES:
        raise RuntimeError("too many K-Planes scales for the kernel")
    d = L.KPlanesDesc()
    d.n_scales = n_scales
    d.channels = planes[0].size(1)
    keep = []
    for s in range(n_scales):
        d.height[s] = planes[3 * s].size(2)
        d.width[s] = planes[3 * s].size(3)
        for p in range(3):
            if planes[3 * s + p] is None:           # absent plane = factor 1
                d.planes[s][p] = None
                continue
            hwc = _hwc(planes[3 * s + p])
            assert hwc.is_contiguous()
            keep.append(hwc)
            d.planes[s][p] = hwc.data_ptr()
    return d, keep


class _KPlanesFeatures(Function):
    @staticmethod
    def forward(ctx: Any, x: torch.Tensor, *planes: torch.Tensor) -> torch.Tensor:  # type: ignore
        lead = x.shape[:-1]
        x2 = x.reshape(-1, 3).to(torch.float32)
        if x2.stride(1) != 1 or (x2.size(0) > 1 and x2.stride(0) < 3):
            x2 = x2.contiguous()
        if not x2.is_cuda:
            raise RuntimeError("tinynerf_amd: tensor must be a CUDA (HIP) tensor -- there is no CPU path")
        dev = x2.device
        desc, keep = _kplanes_desc(planes)
        n = x2.size(0)
        stride = x2.stride(0) if n > 1 else 3
        feat = torch.empty((n, desc.n_scales * desc.channels), device=dev)
        L.call("tn_kplanes_fwd", dev, C.byref(desc), L.ptr(x2), C.c_int64(stride), C.c_int64(n), L.ptr(feat))
        ctx.present = [p is not None for p in planes]
        ctx.save_for_backward(x2, *[p if p is not None else x2.new_empty(0) for p in planes])
        return feat.reshape(*lead, feat.size(-1))

    @staticmethod
    def backward(ctx: Any, grad_feat: torch.Tensor):  # type: ignore
        x2, *planes = ctx.saved_tensors
        dev = x2.device
        desc, keep = _kplanes_desc([p if ctx.present[i] else None for i, p in enumerate(planes)])
        n = x2.size(0)
        stride = x2.stride(0) if n > 1 else 3
        g = grad_feat.reshape(n, -1).to(torch.float32).contiguous()
        planes = [p if ctx.present[i] else None for i, p in enumerate(planes)]
        grads = [None if p is None else torch.zeros_like(p, memory_format=torch.channels_last) for p in planes]
        gp = ((C.c_void_p * 3) * L.TN_KPLANES_MAX_SCALES)()
        for s in range(desc.n_scales):
            for p in range(3):
                gp[s][p] = None if grads[3 * s + p] is None else _hwc(grads[3 * s + p]).data_ptr()
        L.call("tn_kplanes_bwd", dev, C.byref(desc), L.ptr(x2), C.c_int64(stride), C.c_int64(n), L.ptr(g), gp)
        return (None, *grads)


class _PlaneRegulariser(Function):
    """w_tv * mean_planes(loss_tv) + w_l1 * mean_planes(loss_l1) over channel-last planes in two streaming
    kernels per plane (models.py:115-121,165-181); the gradient is added by a 5-point-stencil pass."""

    @staticmethod
    def forward(ctx: Any, w_tv: float, w_l1: float, accumulate: bool, *planes: torch.Tensor) -> torch.Tensor:  # type: ignore
        dev = planes[0].device
        ctx.refs = planes if accumulate else None
        if not planes[0].is_cuda:
            raise RuntimeError("tinynerf_amd: tensor must be a CUDA (HIP) tensor -- there is no CPU path")
        n = len(planes)
        sums = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        coef = torch.empty((n, 3), dtype=torch.float64)
        for i, p in enumerate(planes):
            _, Cc, H, W = p.shape
            L.call("tn_plane_reg_fwd", dev, L.ptr(_hwc(p)), C.c_int(H), C.c_int(W), C.c_int(Cc),
                   C.c_void_p(sums.data_ptr() + i * 24))
            coef[i, 0] = w_tv / (n * Cc * max(H - 1, 1) * W)       # mse over [C,H-1,W]
            coef[i, 1] = w_tv / (n * Cc * H * max(W - 1, 1))       # mse over [C,H,W-1]
            coef[i, 2] = w_l1 / (n * Cc * H * W)
        ctx.coef = coef
        ctx.save_for_backward(*planes)
        return (sums * coef.to(dev)).sum().to(torch.float32)

    @staticmethod
    def backward(ctx: Any, grad_out: torch.Tensor):  # type: ignore
        planes = ctx.saved_tensors
        dev = planes[0].device
        up = grad_out.reshape(1).to(torch.float32).contiguous()
        grads = []
        for i, p in enumerate(planes):
            _, Cc, H, W = p.shape
            ref = ctx.refs[i] if ctx.refs is not None else None
            in_place = ref is not None and ref.grad is not None and ref.grad.stride() == p.stride()
            g = ref.grad if in_place else torch.zeros_like(p, memory_format=torch.channels_last)
            cy, cx, cl1 = (float(v) for v in ctx.coef[i])
            L.call("tn_plane_reg_bwd", dev, L.ptr(_hwc(p)), C.c_int(H), C.c_int(W), C.c_int(Cc), C.c_float(cy), C.c_float(cx),
                   C.c_float(cl1), L.ptr(up), L.ptr(_hwc(g)))
            grads.append(None if in_place else g)
        return (None, None, None, *grads)


class KPlanesFeaturePlane(torch.nn.Module):
    def __init__(
        self,
        feature_dim: int = 8,
        resolution: Tuple[int, int] = (128, 128),
        init: Callable = torch.nn.init.uniform_,
    ):
        super().__init__()
        self.feature_dim = feature_dim
        # logical [1,C,H,W] like the reference; physically channel-last so a texel is one cache line.  The initialiser fills a
        # tensor in MEMORY order, so it runs on the reference's contiguous layout first: the same seed then gives the same logical
        # values as models.py:102-103 (golden G22 holds the constructors to the reference's, bit for bit)
        self.plane = torch.nn.Parameter(init(torch.empty(1, feature_dim, *resolution)).contiguous(memory_format=torch.channels_last))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (..., 2) -> (..., C).  A single plane is the field kernel with one scale whose other two
        planes are absent (factor 1)."""
        x3 = torch.cat([x.reshape(-1, 2).float(), torch.zeros((x.numel() // 2, 1), device=x.device)], -1)
        out = _KPlanesFeatures.apply(x3, self.plane, None, None)
        return out.view([*x.size()[:-1], self.feature_dim])

    def loss_tv(self) -> torch.Tensor:
        return _PlaneRegulariser.apply(1.0, 0.0, False, self.plane)

    def loss_l1(self) -> torch.Tensor:
        return _PlaneRegulariser.apply(0.0, 1.0, False, self.plane)


class KPlanesFeatureField(torch.nn.Module):
    def __init__(self, feature_dim: int = 32, resolutions: Sequence[int] = (128, 256, 512)):
        """``resolutions`` is an extension (the reference hard-codes 128/256/512, models.py:126-142); the default
        reproduces the reference."""
        super().__init__()
        self.planes = torch.nn.ModuleList([
            torch.nn.ModuleList([KPlanesFeaturePlane(feature_dim, resolution=(r, r)) for _ in range(3)])
            for r in resolutions
        ])
        self.dropout = torch.nn.Dropout(0.)
        # coordinate pairs, in this order (models.py:144-146); the kernel hard-codes the same order
        self.dimension_pairs = list(itertools.combinations(range(3), 2))
        self.feature_dim = 32 * len(self.planes)   # the reference hard-codes 32 here too (models.py:147)
        for plane_scale in self.planes:
            assert isinstance(plane_scale, torch.nn.ModuleList)
            assert len(plane_scale) == len(self.dimension_pairs)

    def plane_tensors(self) -> List[torch.Tensor]:
        return [cast(KPlanesFeaturePlane, p).plane for scale in self.planes for p in cast(torch.nn.ModuleList, scale)]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (..., 3) -> (..., C * n_scales): Hadamard product of the 3 planes per scale, scales concatenated."""
        return self.dropout(_KPlanesFeatures.apply(x, *self.plane_tensors()))

    def loss_tv(self) -> torch.Tensor:
        return _PlaneRegulariser.apply(1.0, 0.0, False, *self.plane_tensors())

    def loss_l1(self) -> torch.Tensor:
        return _PlaneRegulariser.apply(0.0, 1.0, False, *self.plane_tensors())

    def regulariser(self, w_tv: float, w_l1: float, accumulate_into_grad: bool = False) -> torch.Tensor:
        """w_tv * loss_tv() + w_l1 * loss_l1() in one pass over the planes (run.py:254-256).  With
        ``accumulate_into_grad`` the backward adds straight into ``plane.grad`` (harness option)."""
        return _PlaneRegulariser.apply(float(w_tv), float(w_l1), bool(accumulate_into_grad), *self.plane_tensors())

    def regulariser_spec(self, w_tv: float, w_l1: float):
        """[(plane, H, W, C, cy, cx, cl1)] and the [n,3] device tensor of the same coefficients: w_tv * loss_tv() + w_l1 *
        loss_l1() = sum_i cy_i * S_y,i + cx_i * S_x,i + cl1_i * S_1,i over the three per-plane sums of the kernels."""
        planes = self.plane_tensors()
        n = len(planes)
        dev = planes[0].device
        spec = []
        coef = torch.empty((n, 3), dtype=torch.float64)
        for i, p in enumerate(planes):
            _, Cc, H, W = p.shape
            cy = w_tv / (n * Cc * max(H - 1, 1) * W)
            cx = w_tv / (n * Cc * H * max(W - 1, 1))
            cl = w_l1 / (n * Cc * H * W)
            spec.append((p, H, W, Cc, cy, cx, cl))
            coef[i, 0], coef[i, 1], coef[i, 2] = cy, cx, cl
        key = (float(w_tv), float(w_l1), str(dev), n)
        if getattr(self, "_reg_coef_key", None) != key:
            self._reg_coef, self._reg_coef_key = coef.to(dev), key
        return spec, self._reg_coef

    @torch.no_grad()
    def regulariser_step(self, w_tv: float, w_l1: float, upstream: float, sums: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Harness form of ``regulariser``: value AND gradient in one launch over all planes (tn_plane_reg_multi).
        ``upstream`` = d(total loss)/d(regulariser) as a host scalar; ``upstream * d reg / d plane`` is added to every
        ``plane.grad`` (which must exist).  Returns the regulariser value (0-dim fp32 tensor, no graph), or -- with a
        caller-owned accumulator ``sums`` -- the [n,3] coefficients that turn the accumulated sums into the value."""
        spec, coef = self.regulariser_spec(w_tv, w_l1)
        n = len(spec)
        dev = spec[0][0].device
        if not spec[0][0].is_cuda:
            raise RuntimeError("tinynerf_amd: tensor must be a CUDA (HIP) tensor -- there is no CPU path")
        items = (L.PlaneRegItem * n)()
        for it, (p, H, W, Cc, cy, cx, cl) in zip(items, spec):
            if p.grad is None or p.grad.stride() != p.stride():
                raise RuntimeError("regulariser_step: every plane needs a .grad buffer with the plane's layout")
            it.plane, it.grad, it.H, it.W, it.C = _hwc(p).data_ptr(), _hwc(p.grad).data_ptr(), H, W, Cc
            it.cy, it.cx, it.cl1 = cy, cx, cl
        if sums is not None:        # caller-owned, zeroed fp64 accumulator [>= 3n]: returns the coefficients, value = (sums * coef).sum()
            L.call("tn_plane_reg_multi", dev, items, C.c_int32(n), C.c_float(upstream), L.ptr(sums))
            return coef
        sums = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        L.call("tn_plane_reg_multi", dev, items, C.c_int32(n), C.c_float(upstream), L.ptr(sums))
        return (sums * coef).sum().to(torch.float32)


class _Linear(Function):
    """y = x W^T + b on the fp32 MFMA (tn_linear_fwd / tn_linear_bwd, csrc/linear.hip) -- torch.nn.functional.linear for
    the one plain Linear of the reference's path (models.py:186)."""

    @staticmethod
    def forward(ctx: Any, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:  # type: ignore
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.size(-1)).to(torch.float32).contiguous()
        w = weight.to(torch.float32).contiguous()
        b = None if bias is None else bias.to(torch.float32).contiguous()
        dev = L.require_cuda(x2, w, b)
        if x2.size(1) != w.size(1):
            raise ValueError(f"linear: x has {x2.size(1)} features, weight expects {w.size(1)}")
        y = torch.empty((x2.size(0), w.size(0)), device=dev)
        L.call("tn_linear_fwd", dev, L.ptr(x2), L.ptr(w), L.ptr(b), C.c_int64(x2.size(0)), C.c_int32(w.size(1)), C.c_int32(w.size(0)), L.ptr(y))
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y.reshape(*lead, w.size(0))

    @staticmethod
    def backward(ctx: Any, grad_y: torch.Tensor):  # type: ignore
        x2, w = ctx.saved_tensors
        dev = x2.device
        gy = grad_y.reshape(-1, w.size(0)).to(torch.float32).contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gw = torch.zeros_like(w) if ctx.needs_input_grad[1] else None
        gb = torch.zeros(w.size(0), device=dev) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        L.call("tn_linear_bwd", dev, L.ptr(x2), L.ptr(w), L.ptr(gy), C.c_int64(x2.size(0)), C.c_int32(w.size(1)), C.c_int32(w.size(0)),
               L.ptr(gx), L.ptr(gw), L.ptr(gb))
        return (None if gx is None else gx.reshape(ctx.x_shape), gw, gb)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _Linear.apply(x, weight, bias)


class KPlanesExplicitOpacityDecoder(torch.nn.Module):
    """sigma = exp(<f, W f + b> - 1) (models.py:183-191).  The Linear is tn_linear_fwd / _bwd (fp32 MFMA, csrc/linear.hip; the
    module keeps a torch.nn.Linear for its parameters and state_dict keys); the per-sample dot product and the truncated
    exponential are one more launch (tn_basis_dot_fwd, backward with the clamp)."""

    def __init__(self, feature_dim):
        super().__init__()
        self.net = torch.nn.Linear(feature_dim, feature_dim)
        self.activation = lambda x: truncated_exp(x - 1.)

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        return _BasisDot.apply(features, linear(features, self.net.weight, self.net.bias), 1, L.ACT_EXP_M1)


class KPlanesExplicitColorDecoder(torch.nn.Module):
    """rgb_k = sigmoid(<f, B_k([PE(d), d, f])>) (models.py:193-205): the basis MLP is one fused MFMA launch, the three dot
    products and the sigmoid another."""

    def __init__(self, feature_dim, n_freqs=8, hidden_dim=128):
        super().__init__()
        self.pe = PositionalEncoding(n_freqs)
        self.feature_dim = feature_dim
        self.n_freqs = n_freqs
        in_dim = feature_dim + n_freqs * 2 * 3 + 3
        self.net = MLP(in_dim, hidden_dim, 3, 3 * feature_dim)

    def forward(self, features: torch.Tensor, rays_d: torch.Tensor) -> torch.Tensor:
        basis = self.net.fused(features, rays_d, L.ENC_DIR_CAT, self.n_freqs, L.ACT_NONE, self.pe.freqs)
        return _BasisDot.apply(features, basis, 3, L.ACT_SIGMOID)


# --------------------------------------------------------------------------------------------
# CoBaFa (models.py:209-266): coefficient grid x sawtooth-warped basis grids -> 128-wide MLP.
# All 7 trilinear lookups, the products and the concat are one launch (tn_cobafa_fwd / _bwd); the
# grids are channels_last_3d parameters with the reference's [1,C,D,H,W] logical shape.
# --------------------------------------------------------------------------------------------
def _dhwc(grid: torch.Tensor) -> torch.Tensor:
    """[1,C,D,H,W] channels_last_3d parameter -> the [D,H,W,C] memory the kernels index (no copy)."""
    g = grid if grid.is_contiguous(memory_format=torch.channels_last_3d) else grid.contiguous(memory_format=torch.channels_last_3d)
    return g.permute(0, 2, 3, 4, 1)


def _cobafa_desc(coef: torch.Tensor, basis: Sequence[torch.Tensor], freqs: Sequence[float]) -> Tuple[L.CobafaDesc, list]:
    d = L.CobafaDesc()
    n = len(basis)
    if n > L.TN_COBAFA_MAX_LEVELS or coef.size(1) != n:
        raise ValueError(f"Cobafa: {n} basis grids need a {n}-channel coefficient grid (max {L.TN_COBAFA_MAX_LEVELS})")
    keep = [_dhwc(coef)]
    d.n_levels = n
    d.coef = keep[0].data_ptr()
    for c in range(3):
        d.coef_res[c] = coef.size(2 + c)
    for i, (b, f) in enumerate(zip(basis, freqs)):
        v = _dhwc(b)
        keep.append(v)
        d.basis[i] = v.data_ptr()
        d.channels[i] = b.size(1)
        d.freqs[i] = float(f)
        for c in range(3):
            d.res[i][c] = b.size(2 + c)
    return d, keep


class _CobafaFeatures(Function):
    @staticmethod
    def forward(ctx: Any, x: torch.Tensor, freqs: Tuple[float, ...], accumulate: bool, coef: torch.Tensor, *basis: torch.Tensor) -> torch.Tensor:  # type: ignore
        # accumulate (harness switch, run.Trainer): the scatter adds straight into grid.grad where it exists
        ctx.param_refs = (coef, *basis) if accumulate else None
        x = x.contiguous()
        dev = L.require_cuda(x)
        if not all(g.is_cuda for g in (coef, *basis)):
            raise RuntimeError("tinynerf_amd: Cobafa grids must be CUDA (HIP) tensors -- there is no CPU path")
        desc, keep = _cobafa_desc(coef, basis, freqs)
        feat = torch.empty((x.size(0), sum(b.size(1) for b in basis)), device=dev)
        L.call("tn_cobafa_fwd", dev, C.byref(desc), L.ptr(x), C.c_int64(x.size(0)), L.ptr(feat))
        ctx.save_for_backward(x, coef, *basis)
        ctx.freqs = freqs
        return feat

    @staticmethod
    def backward(ctx: Any, g: torch.Tensor):  # type: ignore
        x, coef, *basis = ctx.saved_tensors
        desc, keep = _cobafa_desc(coef, basis, ctx.freqs)
        refs = ctx.param_refs if ctx.param_refs is not None else [None] * (1 + len(basis))
        out = []
        for r, p in zip(refs, (coef, *basis)):
            ip = (r is not None and r.requires_grad and r.grad is not None and r.grad.stride() == p.stride()
                  and p.is_contiguous(memory_format=torch.channels_last_3d))
            out.append((r.grad if ip else torch.zeros_like(p, memory_format=torch.channels_last_3d), ip))
        g_coef, g_basis = out[0][0], [o[0] for o in out[1:]]
        gb = (C.c_void_p * len(basis))(*[_dhwc(t).data_ptr() for t in g_basis])
        L.call("tn_cobafa_bwd", x.device, C.byref(desc), L.ptr(x), C.c_int64(x.size(0)), L.ptr(g.contiguous()),
               L.ptr(_dhwc(g_coef)), gb)
        return (None, None, None, *[None if ip else t for t, ip in out])


class SawtoothEncoding(torch.nn.Module):
    """models.py:209-215.  Inside CobafaFeatureField the warp happens in the gather kernel; the module itself
    is only evaluated when called on its own."""

    def __init__(self, f):
        super().__init__()
        self.f = f

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return 2. * torch.remainder(self.f * x, 1.) - 1.


class CobafaGrid(torch.nn.Module):
    """models.py:217-232: a dense [1,C,D,H,W] feature grid with trilinear lookup."""

    def __init__(self, res: int | Tuple[int, int, int], feature_dim: int, init: Callable = torch.nn.init.uniform_):
        super().__init__()
        resolution = (res, res, res) if isinstance(res, int) else tuple(res)
        if feature_dim > 8:
            raise ValueError("CobafaGrid: the gather kernel holds at most 8 channels per voxel")
        # (initialised in the reference's contiguous layout, then re-laid out: same seed -> same logical values as models.py:224-225)
        self.grid = torch.nn.Parameter(init(torch.empty(1, feature_dim, *resolution)).contiguous(memory_format=torch.channels_last_3d))
        self.feature_dim = feature_dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        one = torch.ones((1, 1, 1, 1, 1), device=x.device)
        out = _CobafaFeatures.apply(x.reshape(-1, 3), (0.0,), False, one, self.grid)
        return out.view(*x.size()[:-1], self.feature_dim)


class CobafaFeatureField(torch.nn.Module):
    """models.py:234-266."""

    def __init__(self, basis_res: List[int | Tuple[int, int, int]], coef_res: int | Tuple[int, int, int],
                 freqs: List[float], channels: List[int], mlp_hidden_dim: int):
        super().__init__()
        if not (len(basis_res) == len(freqs) == len(channels)):
            raise ValueError("Cobafa: basis_res, freqs and channels must have one entry per level")
        self.freqs = tuple(float(f) for f in freqs)
        self.basis_grids = torch.nn.ModuleList([CobafaGrid(r, c) for r, c in zip(basis_res, channels)])
        self.encoders = torch.nn.ModuleList([SawtoothEncoding(f) for f in freqs])
        self.coef_grid = CobafaGrid(coef_res, len(basis_res))
        self.dropout = torch.nn.Dropout(0.01)
        self.mlp = MLP(sum(channels), mlp_hidden_dim, 5)
        self.feature_dim = mlp_hidden_dim

    def features(self, x: torch.Tensor) -> torch.Tensor:
        return _CobafaFeatures.apply(x.reshape(-1, 3), self.freqs, bool(self.__dict__.get("accumulate_into_grad")), self.coef_grid.grid,
                                     *[b.grid for b in self.basis_grids])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.mlp(self.dropout(self.features(x)))
