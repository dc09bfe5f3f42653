"""Autograd tape entries of the bf16 channels-last ("CL16") data path (csrc/conv_cl16.hip, csrc/cl_elementwise.hip).

BASELINE.json configs[2] / [4] name 16-bit MFMA variants of the step; the reference itself is fp32-only, so this is a throughput path
with its own tolerance, never the default and never what a parity claim refers to.  Inside a model activations and their gradients
are bf16 tensors whose memory order is (n, d, h, w, c) — torch sees them as ordinary (N, C, [D,] H, W) tensors with permuted strides,
channel stride 1 and a pixel pitch of `pitch_of(C)` elements, padding channels zero — while parameters, BatchNorm statistics, weight
gradients and the optimiser stay fp32 and the module boundary (generator outputs, discriminator inputs and logits) stays fp32 NCDHW.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch.autograd import Function

from . import native as N
from .native import ACT_LEAKY, ACT_NONE, ConvGeom, check, dims5, lib, ptr, stream_ptr
import os

from .ops import _Opaque, _Out, _out_shape, _ws

_FUSE_BN_STATS = os.environ.get("DCV_NO_BN_FUSION") is None

BF16 = torch.bfloat16
_ENABLED = [False]
_HALF = [torch.bfloat16]          # element type of the path: bf16 (default) or fp16 (the library's dcv_clf16_* build of the same kernels; BASELINE configs[4] names fp16 MFMA)


def enable(on: bool = True, half: str = "bf16") -> None:
    """Process-wide switch: models built on dcvgan_amd.layers run their convolution / BatchNorm chains on the CL16 path.  `half`: "bf16" or "fp16" — one element type
    at a time (a tensor of the other one is refused by `_req`)."""
    _ENABLED[0] = bool(on)
    _HALF[0] = {"bf16": torch.bfloat16, "fp16": torch.float16}[half]


def half_dtype():
    return _HALF[0]


def _fn(name: str):
    """The library entry point `dcv_cl_<name>` (bf16) or `dcv_clf16_<name>` (fp16) for the element type in use."""
    return getattr(lib(), ("dcv_clf16_" if _HALF[0] is torch.float16 else "dcv_cl_") + name)


def active() -> bool:
    return _ENABLED[0]


def pitch_of(c: int) -> int:
    """Pixel pitch (elements) of a CL16 tensor with c channels: 8 for thin tensors (<= 8 channels: one 16-byte K granule per tap), else whole
    32-channel K blocks — what the gather kernel reads per (pixel, tap)."""
    return 8 if c <= 8 else (c + 31) // 32 * 32


_ZEROS = {}


def cl_empty(shape, device, pitch: Optional[int] = None, zero: bool = False) -> torch.Tensor:
    """(N, C, [D,] H, W) bf16 tensor in channels-last memory.  Every kernel that produces a CL16 tensor writes whole 8-channel groups (zeros past C), and
    every kernel that reads one stops at C rounded up to 8, so a fresh tensor needs no clearing; `zero=True` is for buffers whose channels are filled
    piecewise (a concatenation whose total is not a multiple of 8)."""
    n, c, sp = shape[0], shape[1], tuple(shape[2:])
    p = pitch if pitch is not None else pitch_of(c)
    store = torch.empty((n,) + sp + (p,), dtype=_HALF[0], device=device)
    if zero:      # a device-to-device copy of a cached block of zeros (torch.zeros would launch one of torch's fill kernels in every iteration)
        key = (str(device), _HALF[0], store.numel())
        z = _ZEROS.get(key)
        if z is None:
            z = _ZEROS[key] = torch.zeros(store.numel(), dtype=_HALF[0], device=device)
        store.view(-1).copy_(z)
    perm = (0, len(sp) + 1) + tuple(range(1, len(sp) + 1))
    return store.permute(*perm)[:, :c]


def _check_out_view(y: torch.Tensor, what: str) -> None:
    """Kernels that produce a CL16 tensor write whole 8-channel groups (zeros past C).  A destination VIEW whose channel count is not a multiple of 8 is therefore only
    legal when nothing live follows it inside its pixel: a fresh tensor, or the trailing member of a ConcatBuffer (which marks it).  Anything else — a middle slice of a
    hand-made buffer — would get zeros written over its neighbour's first channels, silently: refused here."""
    if y.shape[1] % 8 and not getattr(y, "_dcv_trailing", False):
        raise N.NativeError(f"{what}: out= view with {y.shape[1]} channels (not a multiple of 8) that is not the trailing slice of its buffer: the kernels write whole "
                            "8-channel groups and would overwrite the channels behind it")


def is_cl(t: torch.Tensor) -> bool:
    return t.dtype == torch.bfloat16 or t.dtype == torch.float16


def _req(t: torch.Tensor, what: str):
    if not t.is_cuda or t.dtype != _HALF[0]:
        raise N.NativeError(f"{what}: expected a {_HALF[0]} HIP tensor (the element type ops_cl.enable selected), got {t.dtype} on {t.device}")
    if t.dim() > 1 and t.shape[1] > 1 and t.stride(1) != 1:
        raise N.NativeError(f"{what}: expected channels-last memory (channel stride 1), got strides {tuple(t.stride())}")
    if t.device.index != torch.cuda.current_device():
        raise N.NativeError(f"{what}: tensor is on {t.device} but the current device is cuda:{torch.cuda.current_device()}")


def as_cl(t: torch.Tensor) -> torch.Tensor:
    """A gradient handed over by autograd: already channels-last in practice (sums of CL tensors keep their strides); anything else is re-laid."""
    if t.dtype == _HALF[0] and (t.shape[1] == 1 or t.stride(1) == 1) and all(s != 0 or n == 1 for s, n in zip(t.stride(), t.shape)):
        st = t.stride()
        if t.dim() < 3 or st[-1] % 8 == 0:
            return t
    out = cl_empty(t.shape, t.device, zero=bool(t.shape[1] % 8))
    out.copy_(t)
    return out


# --------------------------------------------------------------------------- #
# module boundary
# --------------------------------------------------------------------------- #
class _FromF32(Function):
    @staticmethod
    def forward(ctx, x, out=None):
        N._require(x, "from_f32 input")
        if out is not None:
            _check_out_view(out.t, "from_f32")
        y = cl_empty(x.shape, x.device) if out is None else out.t.detach()
        xd, yd = dims5(x), dims5(y)
        check(_fn("from_f32")(ptr(x), C.byref(xd), ptr(y), C.byref(yd), stream_ptr()), "dcv_cl_from_f32")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = as_cl(dy)
        dx = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
        dyd, dxd = dims5(dy), dims5(dx)
        check(_fn("to_f32")(ptr(dy), C.byref(dyd), ptr(dx), C.byref(dxd), 0, stream_ptr()), "dcv_cl_to_f32")
        return dx, None


# A module's fp32 output is the conversion of a CL16 tensor, and the next module converts it straight back (generator frames -> colour generator -> discriminators): the
# fp32 tensor carries its source as `_dcv_cl_twin`, `from_f32` hands that out instead of converting, and the gradient flows from the consumer into the producer's
# tape without the fp32 detour either (bf16 -> fp32 -> bf16 is the identity, so the forward values are the same bits).  Views made at the boundary (frames -> video)
# carry the twin through `carry_twin`.  The fp32 tensor itself stays what the caller sees and what slicing / fp32 ops read.  DCV_CL_NO_TWINS=1: always convert (A/B).
_TWINS = os.environ.get("DCV_CL_NO_TWINS") is None


def twin_of(x: torch.Tensor):
    """x's CL16 source, or None — also when x was written in place since the twin was attached (`clamp_`, `mul_`, a masked write between generator and
    discriminator): the twin then no longer holds x's values, so the consumer converts x itself and the gradient flows through the modification."""
    t = getattr(x, "_dcv_cl_twin", None) if _TWINS else None
    if t is None or t.dtype != _HALF[0] or tuple(t.shape) != tuple(x.shape):
        return None
    return t if getattr(x, "_dcv_cl_twin_ver", None) == (x._version, x.data_ptr()) else None


def carry_twin(dst: torch.Tensor, src: torch.Tensor, view) -> torch.Tensor:
    """dst = view(src) was made of an fp32 boundary tensor by pure view operations: give dst the same view of src's CL16 twin."""
    t = twin_of(src)
    if t is not None:
        v = view(t)
        if v.data_ptr() == t.data_ptr() and (v.shape[1] == 1 or v.stride(1) == 1):      # still a channels-last view of the same memory (torch made no copy)
            dst._dcv_cl_twin = v
            dst._dcv_cl_twin_ver = (dst._version, dst.data_ptr())      # a view shares its base's version counter
    return dst


def from_f32(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 (any strides) -> CL16; `out`: destination view (a channel slice of a concat buffer)."""
    if out is None:
        t = twin_of(x)
        if t is not None:
            return t
    return _FromF32.apply(x, None if out is None else _Out(out))


class _ToF32(Function):
    @staticmethod
    def forward(ctx, x):
        _req(x, "to_f32 input")
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        xd, yd = dims5(x), dims5(y)
        check(_fn("to_f32")(ptr(x), C.byref(xd), ptr(y), C.byref(yd), 0, stream_ptr()), "dcv_cl_to_f32")
        return y

    @staticmethod
    def backward(ctx, dy):
        if any(s == 0 and n > 1 for s, n in zip(dy.stride(), dy.shape)):
            dy = dy.contiguous()
        dx = cl_empty(dy.shape, dy.device)
        dyd, dxd = dims5(dy), dims5(dx)
        check(_fn("from_f32")(ptr(dy), C.byref(dyd), ptr(dx), C.byref(dxd), stream_ptr()), "dcv_cl_from_f32")
        return dx


def to_f32(x: torch.Tensor) -> torch.Tensor:
    """CL16 -> contiguous fp32 NCDHW (which remembers x: `from_f32` of it is x again)."""
    y = _ToF32.apply(x)
    if _TWINS:
        y._dcv_cl_twin = x
        y._dcv_cl_twin_ver = (y._version, y.data_ptr())
    return y


# --------------------------------------------------------------------------- #
# elementwise
# --------------------------------------------------------------------------- #
def _ew(kind, x, z, a=0.0, b=0.0, seed=0, offset=0, out=None):
    y = cl_empty(x.shape, x.device, pitch=None) if out is None else out
    xd, yd = dims5(x), dims5(y)
    zd = dims5(z) if z is not None else None
    check(_fn("elementwise")(kind, ptr(x), C.byref(xd), ptr(z), C.byref(zd) if z is not None else None, ptr(y), C.byref(yd), float(a), float(b),
                                   int(seed), int(offset), stream_ptr()), "dcv_cl_elementwise")
    return y


class _NoiseAddCl(Function):
    @staticmethod
    def forward(ctx, x, sigma, sample, seed, offset):
        _req(x, "noise input")
        if sample is not None:
            return _ew(1, x, sample, 1.0, sigma)
        return _ew(2, x, None, sigma, 0.0, seed, offset)

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None, None, None


def noise_add(x, sigma: float, sample=None, seed: int = 0, offset: int = 0):
    """x + sigma N(0,1) on a CL16 tensor; `sample`: an injected fp32 draw (converted) instead of the device Philox stream."""
    if sample is not None and sample.dtype != _HALF[0]:
        with torch.no_grad():
            sample = from_f32(sample)
    return _NoiseAddCl.apply(x, float(sigma), sample, int(seed), int(offset))


class _ActCl(Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        _req(x, "activation input")
        y = _ew(5 if act == ACT_LEAKY else 6, x, None, slope)
        ctx.cfg = (act, slope)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        _req(y, "activation backward: the saved output (element type switched since the forward?)")
        act, slope = ctx.cfg
        return _ew(3 if act == ACT_LEAKY else 4, as_cl(dy), y, slope), None, None


def act(x, kind: int, slope: float = 0.0):
    return _ActCl.apply(x, kind, float(slope))


# --------------------------------------------------------------------------- #
# convolution
# --------------------------------------------------------------------------- #
def _packed(w: torch.Tensor, which: int, g: ConvGeom, xd, yd, key_dims):
    """bf16 K-major tiles of `w` for (pass, geometry): repacked when the tensor's autograd version (bumped by every optimiser step) changes."""
    cache = getattr(w, "_dcv_clpack", None)
    if cache is None:
        cache = w._dcv_clpack = {}
    key = (which, g.key(), key_dims, _HALF[0])
    e = cache.get(key)
    stamp = (w._version, w.data_ptr())
    L = lib()
    if e is None:
        nb = _fn("packed_bytes")(C.byref(g), C.byref(xd), C.byref(yd), which)
        if nb == 0:
            raise N.NativeError("dcv_cl_packed_bytes: " + L.dcv_last_error().decode())
        e = cache[key] = [None, torch.empty(nb, dtype=torch.uint8, device=w.device), None, None]
    cur = torch.cuda.current_stream(w.device)
    if e[0] != stamp:
        if e[3] is not None and e[3].cuda_stream != cur.cuda_stream and e[2] is not None:
            cur.wait_event(e[2])      # (a stream that still reads the old tiles is not waited for: a repack follows an optimiser step, and every consumer of the old weights is ordered before that)
        check(_fn("pack_weights")(C.byref(g), C.byref(xd), C.byref(yd), which, ptr(w), ptr(e[1]), e[1].numel(), stream_ptr()), "dcv_cl_pack_weights")
        e[0] = stamp
        if e[2] is None:
            e[2] = torch.cuda.Event()
        e[2].record(cur)
        e[3] = cur
    elif e[3] is not None and e[3].cuda_stream != cur.cuda_stream:
        cur.wait_event(e[2])          # packed on another stream (a discriminator lane, the D phase's generator stream): this stream's kernels read the tiles behind that launch
    return e[1]


class _ConvCl(Function):
    @staticmethod
    def forward(ctx, x, w, g: ConvGeom, act: int, slope: float, out=None, grad_slot=None, bn_stats=None, act_slot=None):
        _req(x, "conv input"); N._require(w, "conv weight")
        ctx.grad_slot = grad_slot
        # this conv + (Leaky)ReLU writes a skip tensor (second slice of a concat buffer): the consumer's data gradient may apply the derivative (ops.GradSlot)
        ctx.act_slot = act_slot if (act_slot is not None and act == ACT_LEAKY) else None
        if ctx.act_slot is not None:
            ctx.act_slot.act, ctx.act_slot.act_applied = (act, slope), False
        if x.shape[1] != g.cin:
            raise N.NativeError(f"conv: input has {x.shape[1]} channels, module expects {g.cin}")
        shape = _out_shape(g, x)
        if out is not None:
            _check_out_view(out.t, "conv")
        y = cl_empty(shape, x.device) if out is None else out.t.detach()
        if tuple(y.shape) != tuple(shape):
            raise N.NativeError(f"out= has shape {tuple(y.shape)}, expected {tuple(shape)}")
        xd, yd = dims5(x), dims5(y)
        pk = _packed(w, 0, g, xd, yd, tuple(x.shape))
        wsp, wsn = _ws("clconv", _fn("conv_workspace_bytes")(C.byref(g), C.byref(xd), C.byref(yd), 0), x.device)
        sbytes = _fn("conv_stats_bytes")(C.byref(g), C.byref(xd), C.byref(yd)) if (bn_stats is not None and act == ACT_NONE and _FUSE_BN_STATS) else 0
        if sbytes:
            # conv -> BatchNorm pair: the epilogue leaves per-tile {sum, sum^2} of the stored bf16 values, the BatchNorm op skips its statistics pass over y
            stat = torch.empty(sbytes // 4, dtype=torch.float32, device=x.device)
            nparts, pitch = C.c_int(0), C.c_int(0)
            check(_fn("conv_forward_stats")(C.byref(g), ptr(x), C.byref(xd), ptr(pk), ptr(y), C.byref(yd), ptr(stat), sbytes, C.byref(nparts), C.byref(pitch),
                                                  wsp, wsn, stream_ptr()), "dcv_cl_conv_forward_stats")
            if nparts.value > 0:
                bn_stats.append((stat, nparts.value, pitch.value))
        else:
            check(_fn("conv_forward")(C.byref(g), ptr(x), C.byref(xd), ptr(pk), ptr(y), C.byref(yd), act, slope, wsp, wsn, stream_ptr()), "dcv_cl_conv_forward")
        ctx.g, ctx.act, ctx.slope = g, act, slope
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        _req(x, "conv backward: the saved input (ops_cl.enable(half=...) was switched between this layer's forward and its backward?)")
        g = ctx.g
        L = lib()
        dy = as_cl(dy)
        fused_away = ctx.act_slot is not None and ctx.act_slot.act_applied      # the consumer's data gradient already applied act' (gated epilogue)
        if ctx.act_slot is not None:
            ctx.act_slot.act_applied = False
        if ctx.act != ACT_NONE and not fused_away:
            dy = _ew(3 if ctx.act == ACT_LEAKY else 4, dy, y, ctx.slope)
        xd, dyd = dims5(x), dims5(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # x is a U-Net skip tensor whose other consumer (the concatenation) has already delivered its gradient slice: add this data gradient into that
            # slice in the GEMM epilogue and hand autograd nothing to sum (ops.GradSlot, as on the fp32 path)
            slot = ctx.grad_slot
            into = slot.take(x) if slot is not None else None
            if into is not None and (into.dtype != _HALF[0] or into.stride() != x.stride()):
                slot.g, into = into, None
            dx = into if into is not None else cl_empty(x.shape, x.device)
            dxd = dims5(dx)
            pk = _packed(w, 1, g, dxd, dyd, tuple(x.shape))
            wsp, wsn = _ws("clconv", _fn("conv_workspace_bytes")(C.byref(g), C.byref(dxd), C.byref(dyd), 1), x.device)
            rc = N.DCV_EUNSUPPORTED
            from . import ops as _o2
            if into is not None and slot.act is not None and _o2._GATED_DGRAD and tuple(x.stride()) == tuple(into.stride()):
                # ... and the derivative of the activation that produced x, read off x, in the same epilogue (Inconv -> DownBlock 0)
                xgd = dims5(x)
                rc = _fn("conv_backward_data_gated")(C.byref(g), ptr(dy), C.byref(dyd), ptr(pk), ptr(dx), C.byref(dxd), 1, ptr(x), C.byref(xgd),
                                                       slot.act[0], slot.act[1], wsp, wsn, stream_ptr())
                if rc == 0:
                    slot.act_applied = True
                elif rc != N.DCV_EUNSUPPORTED:
                    check(rc, "dcv_cl_conv_backward_data_gated")
            if rc == N.DCV_EUNSUPPORTED:
                check(_fn("conv_backward_data")(C.byref(g), ptr(dy), C.byref(dyd), ptr(pk), ptr(dx), C.byref(dxd), int(into is not None), wsp, wsn, stream_ptr()),
                      "dcv_cl_conv_backward_data")
            if into is not None:
                dx = None
        if ctx.needs_input_grad[1]:
            dw = _wgrad_on_side(g, x, xd, dy, dyd, w)
        return dx, dw, None, None, None, None, None, None, None


def _wgrad_on_side(g, x, xd, dy, dyd, w):
    """The layer's weight gradient: on the chain stream's companion where that is allowed (below), else in place."""
    need = _fn("wgrad_workspace_bytes")(C.byref(g), C.byref(xd), C.byref(dyd))
    if need == 0:
        raise N.NativeError("dcv_cl_wgrad_workspace_bytes: " + lib().dcv_last_error().decode())
    from . import ops as _o
    side = _o.wgrad_companion(x.device, w, _WGRAD_SIDE)
    if side is None:
        return _wgrad_cl(g, x, xd, dy, dyd, w, need)
    cur = torch.cuda.current_stream(x.device)
    side.wait_stream(cur)
    for t in (x, dy):
        t.record_stream(side)
    with torch.cuda.stream(side):
        dw = _wgrad_cl(g, x, xd, dy, dyd, w, need)
    _o.wgrad_join_at_end(x.device, cur, side)
    return dw


# Weight gradients off the chain: ops.wgrad_companion (the main stream's backward hands them to one companion stream; same kernels, bit-identical results).
_WGRAD_SIDE = os.environ.get("DCV_CL_NO_WGRAD_SIDE") is None      # (DCV_NO_WGRAD_SIDE=1 turns it off on both paths)


def _wgrad_cl(g, x, xd, dy, dyd, w, need):
    """The weight gradient of one CL16 convolution on the CURRENT stream: returns the tensor for autograd, or None when it was added into an existing target."""
    from . import ops as _o
    wsp, wsn = _ws("clconv", need, x.device)
    tgt = _o.grad_target(w) if _o._OWN_ACCUMULATION else None
    if tgt is not None:      # a later contribution to this parameter's gradient: added by the slab reduce (ops.grad_target), nothing for autograd to sum
        check(_fn("conv_backward_weight_acc")(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(dyd), C.c_void_p(tgt), 1, wsp, wsn, stream_ptr()),
              "dcv_cl_conv_backward_weight_acc")
        return None
    slot = getattr(w, "_dcv_grad_slot", None)       # data parallel: the parameter's slice of its bucket's flat buffer (optim.GradBucket), as on the fp32 path
    if slot is not None and w.grad is None and getattr(w, "_dcv_slot_epoch", None) is not _o._Conv._epoch[0] and _o._OWN_ACCUMULATION:
        w._dcv_slot_epoch = _o._Conv._epoch[0]
        b = getattr(w, "_dcv_bucket", None)
        if b is not None and b() is not None:
            b().before_slot_write(w)
        dw = slot.detach()
    else:
        dw = torch.empty(w.shape, dtype=torch.float32, device=w.device)
    check(_fn("conv_backward_weight")(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(dyd), ptr(dw), wsp, wsn, stream_ptr()), "dcv_cl_conv_backward_weight")
    if _o._OWN_ACCUMULATION:
        _o.note_first(w, dw)
    return dw


def conv(x, w, g: ConvGeom, act: int = ACT_NONE, slope: float = 0.0, out=None, grad_slot=None, bn_stats=None, act_slot=None):
    """y = act(conv(x, w)) on CL16 tensors; fp32 weights in torch layout; `out`: destination view (a channel slice of a concat buffer);
    `grad_slot`: ConcatBuffer.slot of the buffer whose second slice IS x (a skip connection); `bn_stats`: a list that receives (buffer, nparts, pitch) when the
    epilogue produced the following BatchNorm's sums (conv -> BatchNorm pairs in training mode, no activation in between)."""
    return _ConvCl.apply(x, w, g, act, float(slope), None if out is None else _Out(out), grad_slot, bn_stats, act_slot)


# --------------------------------------------------------------------------- #
# BatchNorm (+ Dropout2d mask) + activation
# --------------------------------------------------------------------------- #
class _BnActCl(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, mask, training, momentum, eps, act, slope, out=None, nbt=None, partials=None):
        _req(x, "bn input")
        L = lib()
        Cn = x.shape[1]
        if out is not None:
            _check_out_view(out.t, "bn_act")
        y = cl_empty(x.shape, x.device) if out is None else out.t.detach()
        stats = torch.empty((2, Cn), dtype=torch.float32, device=x.device)
        xd, yd = dims5(x), dims5(y)
        wsp, wsn = _ws("clbn", _fn("bn_workspace_bytes")(Cn), x.device)
        if partials is not None and training:
            stat, nparts, pitch = partials.v
            check(_fn("bn_act_forward_stats")(ptr(x), C.byref(xd), ptr(y), C.byref(yd), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                                ptr(nbt), ptr(stats[0]), ptr(stats[1]), ptr(mask), momentum, eps, act, slope, ptr(stat), nparts, pitch,
                                                wsp, wsn, stream_ptr()), "dcv_cl_bn_act_forward_stats")
        else:
            check(_fn("bn_act_forward")(ptr(x), C.byref(xd), ptr(y), C.byref(yd), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                          ptr(nbt) if training else None, ptr(stats[0]), ptr(stats[1]), ptr(mask), int(training), momentum, eps, act, slope,
                                          wsp, wsn, stream_ptr()), "dcv_cl_bn_act_forward")
        ctx.cfg = (bool(training), act, slope)
        ctx.save_for_backward(x, gamma, beta, stats, mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats, mask = ctx.saved_tensors
        _req(x, "BatchNorm backward: the saved input (ops_cl.enable(half=...) was switched between this layer's forward and its backward?)")
        training, act, slope = ctx.cfg
        L = lib()
        dy = as_cl(dy)
        Cn = x.shape[1]
        dx = cl_empty(x.shape, x.device)
        dgb = torch.empty((2, Cn), dtype=torch.float32, device=x.device)
        dyd, xd, dxd = dims5(dy), dims5(x), dims5(dx)
        wsp, wsn = _ws("clbn", _fn("bn_workspace_bytes")(Cn), x.device)
        check(_fn("bn_act_backward")(ptr(dy), C.byref(dyd), ptr(x), C.byref(xd), ptr(dx), C.byref(dxd), ptr(gamma), ptr(beta), ptr(stats[0]), ptr(stats[1]),
                                       ptr(mask), int(training), act, slope, ptr(dgb[0]), ptr(dgb[1]), wsp, wsn, stream_ptr()), "dcv_cl_bn_act_backward")
        from . import ops as _o
        if _o._OWN_ACCUMULATION:
            return dx, _o.deliver_small(gamma, dgb[0]), _o.deliver_small(beta, dgb[1]), None, None, None, None, None, None, None, None, None, None, None
        return dx, dgb[0], dgb[1], None, None, None, None, None, None, None, None, None, None, None


def bn_act(x, gamma, beta, running_mean, running_var, training: bool, act: int = ACT_NONE, slope: float = 0.0, mask=None, momentum: float = 0.1,
           eps: float = 1e-5, out=None, num_batches_tracked=None, partials=None):
    """`partials`: (buffer, nparts, pitch) left by the producing convolution's epilogue (conv(..., bn_stats=[]))."""
    return _BnActCl.apply(x, gamma, beta, running_mean, running_var, mask, training, float(momentum), float(eps), act, float(slope),
                          None if out is None else _Out(out), num_batches_tracked, None if partials is None else _Opaque(partials))


# --------------------------------------------------------------------------- #
# concatenation: both producers write into the two channel slices of one buffer
# --------------------------------------------------------------------------- #
class ConcatBuffer:
    def __init__(self, n, ca, cb, spatial, device):
        from .ops import _JoinSlices
        self._join = _JoinSlices
        if ca % 8:
            raise N.NativeError("ConcatBuffer: the first member's channel count must be a multiple of 8 (16-byte aligned second slice)")
        self.buf = cl_empty((n, ca + cb) + tuple(spatial), device, zero=bool((ca + cb) % 8))
        self.first, self.second = self.buf[:, :ca], self.buf[:, ca:]
        self.second._dcv_trailing = True          # nothing live behind it inside a pixel (only the buffer's own padding): may have any channel count (_check_out_view)
        from .ops import GradSlot, _SKIP_ACCUMULATE
        self.slot = GradSlot() if _SKIP_ACCUMULATE else None

    def join(self, a, b):
        return self._join.apply(a, b, _Out(self.buf), self.slot)
