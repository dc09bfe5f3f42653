"""Autograd tape entries whose forward/backward launch the gfx950 kernels.

PyTorch supplies the tape, device memory and the current stream; every tensor
computation of the G+D step below goes through libdcvgan_hip.so (native.py).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import native as N
from .native import ACT_LEAKY, ACT_NONE, ACT_TANH, ConvGeom, check, dims5, lib, ptr, stream_ptr

__all__ = ["invalidate_packed_weights", "conv", "bn_act", "act", "noise_add", "cat_channels", "temporal_diff", "gan_loss", "gru_sequence",
           "normal", "dropout2d_mask", "conv_geom", "ACT_NONE", "ACT_LEAKY", "ACT_TANH"]


def _dense(t: torch.Tensor) -> torch.Tensor:
    """Gradients can arrive as expanded (stride-0) views; give the kernels real memory."""
    if any(s == 0 and n > 1 for s, n in zip(t.stride(), t.shape)):
        return t.contiguous()
    return t


_POISON = bool(int(os.environ.get("DCV_DEBUG_POISON", "0")))


def _empty(shape, device) -> torch.Tensor:
    """Output allocation.  DCV_DEBUG_POISON=1 fills it with NaN so that an element a
    kernel fails to write cannot hide behind recycled memory."""
    if _POISON:
        return torch.full(tuple(shape), float("nan"), dtype=torch.float32, device=device)
    return torch.empty(tuple(shape), dtype=torch.float32, device=device)


def _ws(tag: str, nbytes: int, device) -> Tuple[C.c_void_p, int]:
    buf = N.scratch.get(tag, int(nbytes), device)
    return C.c_void_p(buf.data_ptr()), buf.numel()


# --------------------------------------------------------------------------- #
# convolution
# --------------------------------------------------------------------------- #
def conv_geom(weight: torch.Tensor, stride, padding, transposed: bool, precision=None) -> ConvGeom:
    """Geometry from a torch weight: (Cout,Cin,k..) or, transposed, (Cin,Cout,k..).  `precision`: None (process default), "fp32" or "bf16"."""
    k = list(weight.shape[2:])
    s, p = list(stride), list(padding)
    if len(k) == 2:
        k, s, p = [1] + k, [1] + s, [0] + p
    a, b = weight.shape[0], weight.shape[1]
    cin, cout = (a, b) if transposed else (b, a)
    return ConvGeom(k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2], int(transposed), cin, cout, N.PRECISION_CODE[precision])


def _out_shape(g: ConvGeom, x: torch.Tensor):
    sp = list(x.shape[2:])
    if len(sp) == 2:
        sp = [1] + sp
    ks, ss, ps = (g.kd, g.kh, g.kw), (g.sd, g.sh, g.sw), (g.pd, g.ph, g.pw)
    if g.transposed:
        o = [(sp[i] - 1) * ss[i] - 2 * ps[i] + ks[i] for i in range(3)]
    else:
        o = [(sp[i] + 2 * ps[i] - ks[i]) // ss[i] + 1 for i in range(3)]
    if x.dim() == 4:
        return (x.shape[0], g.cout, o[1], o[2])
    return (x.shape[0], g.cout, o[0], o[1], o[2])


class _Out:
    """Wrapper that keeps a destination view out of autograd's sight (it is plain memory to write
    into, e.g. a channel slice of a concat buffer)."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class _Opaque:
    """Non-tensor payload handed through Function.apply (autograd ignores it)."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


def _dest(out, shape, device):
    if out is None:
        return _empty(shape, device)
    t = out.t
    if tuple(t.shape) != tuple(shape):
        raise N.NativeError(f"out= has shape {tuple(t.shape)}, expected {tuple(shape)}")
    N._require(t, "out= destination")
    return t.detach()  # fresh tensor object sharing the memory; becomes the op's output


class _PackCache:
    """Packed (K-major) weights of one weight tensor, owned on this side of the ABI (dcv_wpack): one buffer per
    (pass, geometry, input layout), valid while the tensor's autograd version and storage stay what they were when it
    was packed.  Every in-place change bumps the version — load_state_dict / init through torch, the HIP Adam through
    torch.autograd.graph.increment_version (optim.Adam.step) — so a layer that runs 2-3 times between optimiser steps
    (D on real and fake batches, trainer.py:299-309,347-349) is packed once.  Lives in `weight._dcv_pack`: it dies
    with the tensor object, so recycled addresses cannot alias."""
    __slots__ = ("entries",)

    def __init__(self):
        self.entries = {}

    def get(self, w, which, g, xd_t, yd_t, xd, yd):
        prec = lib().dcv_conv_effective_precision(C.byref(g))                              # the packed FORMAT (and size) depends on it; the library checks the stamp too
        key = (which, g.key(), prec, tuple(xd_t.shape), tuple(xd_t.stride()), tuple(yd_t.shape), tuple(yd_t.stride()))   # the K order depends on the layout
        e = self.entries.get(key)
        stamp = (w._version, w.data_ptr(), prec)
        if _POISON:
            # debug builds of the tests: edits autograd cannot see (`p.data.normal_()`, raw-pointer writes) do not bump the version;
            # a checksum of the live weights in the stamp turns "silently convolving with stale packed weights" into a repack
            wd = w.detach().double()
            stamp = stamp + (float(wd.sum()), float(wd.abs().sum()))
        if e is None:
            nbytes = lib().dcv_conv_packed_bytes(C.byref(g), C.byref(xd), C.byref(yd), which)
            if nbytes == 0:
                return None
            e = self.entries[key] = [None, torch.empty(nbytes, dtype=torch.uint8, device=w.device), None, None]
        ready = e[0] == stamp
        if w.is_cuda:
            cur = torch.cuda.current_stream(w.device)
            if e[2] is not None and e[3] != cur.cuda_stream:
                # packed (or last repacked) by a launch on another stream — a discriminator lane, the D phase's generator stream: this stream's kernels read the
                # buffer behind that launch.  (A stream still READING old tiles is not waited for by a repack: a repack follows an optimiser step, and every
                # consumer of the old weights is ordered before that step.)
                cur.wait_event(e[2])
        e[0] = None                     # not valid again until the launch that (re)packs it has been accepted: see commit()
        pk = N.WPack(e[1].data_ptr(), e[1].numel(), int(ready), prec)
        pk._entry, pk._stamp, pk._packs = e, stamp, not ready
        return pk

    @staticmethod
    def commit(pk):
        """The conv call that consumed `pk` returned success: its packed buffer now holds the weights of `stamp`.  A call that
        fails (EWORKSPACE, a launch error) never gets here, so a retry packs again instead of reading an unpacked buffer."""
        if pk is not None:
            e = pk._entry
            e[0] = pk._stamp
            if pk._packs and e[1].is_cuda:      # this call wrote the buffer: other streams order themselves behind it (get())
                cur = torch.cuda.current_stream(e[1].device)
                if e[2] is None:
                    e[2] = torch.cuda.Event()
                e[2].record(cur)
                e[3] = cur.cuda_stream


_USE_PACK_CACHE = os.environ.get("DCV_NO_PACK_CACHE") is None
_SKIP_ACCUMULATE = os.environ.get("DCV_NO_SKIP_ACCUMULATE") is None
_GATED_DGRAD = os.environ.get("DCV_NO_GATED_DGRAD") is None


def _pack_of(w):
    if not _USE_PACK_CACHE:
        return None
    pc = getattr(w, "_dcv_pack", None)
    if pc is None:
        pc = w._dcv_pack = _PackCache()
    return pc


def invalidate_packed_weights(obj) -> None:
    """Forget the cached packed copies of a weight tensor / of every parameter of a module.  Needed only after an in-place
    edit that autograd cannot see (`p.data.normal_()`, a raw-pointer write): such edits do not bump `p._version`, the
    stamp the caches are validated with.  load_state_dict / copy_ / in-place ops on the parameter itself / torch and HIP
    optimiser steps / util.init_weights all bump it and need nothing."""
    ts = obj.parameters() if isinstance(obj, torch.nn.Module) else [obj]
    for t in ts:
        if getattr(t, "_dcv_pack", None) is not None:
            t._dcv_pack = None


# --------------------------------------------------------------------------- #
# parameter-gradient accumulation without torch's elementwise adds (round 5)
# --------------------------------------------------------------------------- #
# A parameter's gradient is the sum of several contributions whenever (a) the parameter is used twice in ONE backward — every discriminator parameter: D runs on the real
# and on the fake batch (trainer.py:299-309) — or (b) its .grad still holds an earlier backward's gradient — the discriminators in the G phase (trainer.py:356 after :319;
# they are only zeroed at :288-290).  Autograd forms these sums with at::add launches (118 per iteration in round 4's traces: torch's own kernels, built with packed-FP32
# code the library's build flag cannot reach).  Here the SECOND and later contributions are added by the library's own kernels — the weight gradient's slab
# reduce (dcv_conv_backward_weight_acc), dcv_axpby for the small BatchNorm / GRU gradients — straight into the tensor that already holds the first, and the backward node
# returns None for that input: same operands, same order, one rounding = bit-identical sums.
def _task_id() -> int:
    return torch._C._current_graph_task_id()


def grad_target(p):
    """Device pointer of the fp32 tensor this backward's NEXT gradient contribution of parameter `p` must be added to; None if this is the first one.
    (b) an existing .grad; (a) the tensor the first use in the running backward handed to autograd (noted by `note_first`; valid for that graph task only:
    the engine holds the tensor until the parameter's AccumulateGrad node has run, which is after every use)."""
    if not isinstance(p, torch.nn.Parameter):
        return None
    g = p.grad
    if g is not None:
        if g.dtype == torch.float32 and g.is_cuda and g.is_contiguous() and g.shape == p.shape:
            b = getattr(p, "_dcv_bucket", None)      # data parallel: autograd will not visit this parameter's AccumulateGrad node (nothing is returned for it), so
            b = b() if b is not None else None       # the bucket's "a backward has produced gradients" mark is set here instead of by its post-accumulate hook
            if b is not None:
                b.note_inplace(p)
            return g.data_ptr()
        return None
    a = getattr(p, "_dcv_acc", None)
    if a is not None and a[0] == _task_id() and a[0] >= 0:
        return a[1]
    return None


def note_first(p, t) -> None:
    """`t` is the first contribution to p's gradient in the running backward (only its address is kept: an extra reference would make AccumulateGrad clone it)."""
    if isinstance(p, torch.nn.Parameter) and t is not None and t.is_contiguous():
        p._dcv_acc = (_task_id(), t.data_ptr())


def deliver_small(p, t):
    """A small fp32 gradient `t` of parameter `p` (BatchNorm weight / bias, GRU weights): added in place by dcv_axpby when a target exists, else returned for autograd."""
    tgt = grad_target(p)
    if tgt is None or not t.is_contiguous():
        note_first(p, t if t.is_contiguous() else None)
        return t
    td = dims5(t.reshape(1, -1, 1, 1))
    check(lib().dcv_axpby(ptr(t), C.byref(td), 1.0, C.c_void_p(tgt), C.byref(td), 1.0, C.c_void_p(tgt), C.byref(td), stream_ptr()), "dcv_axpby (gradient accumulation)")
    return None


_OWN_ACCUMULATION = os.environ.get("DCV_TORCH_GRAD_ADDS") is None      # DCV_TORCH_GRAD_ADDS=1: leave the sums to autograd (A/B, bit-identity test)

# --------------------------------------------------------------------------- #
# Weight gradients off the chain (round 5).  Nothing in a backward pass reads a weight gradient, so it need not sit between a layer's data gradient and the next layer's
# BatchNorm backward on the same stream: the main stream's backward (the generators' chain, the longest serial one of the iteration) hands its weight gradients to ONE
# companion stream.  The companion waits for the chain up to the call (x, dy and earlier sums are complete), the chain never waits for the companion, and the engine's
# end-of-backward callback joins them (optimiser, collective and host reads then see complete gradients).  One companion = one order: the sums into a parameter stay in
# host order, the results are bit-identical.  The kernels are the same; they now run beside HBM-bound BatchNorm passes and under-filled deep layers instead of between them.
#   bf16 channels-last path, surreal-depth1 B = 100, same-box alternating runs on five boxes: 38.7 -> 37.7, 38.7 -> 37.75, 37.95 -> 37.55, 38.25 -> 37.45, 38.42 -> 37.35 ms;
#   isogd-depth 33.74 -> 32.66;   fp32 headline (isogd-depth B = 70): 109.63 -> 108.80 ms (profiles/r05_ab_cl16.txt, calls 26-33).
# Only with the library's own gradient sums (a sum autograd forms would be a kernel on the chain's stream reading the companion's result unordered; a FIRST contribution is
# handed to autograd, which takes it over without a kernel — it holds the only reference) and never with data-parallel buckets (their collectives are ordered on the
# chain's stream).  Measured on the 16-bit path and not shipped: the discriminator lanes' weight gradients on companions of their own (seven streams on the runtime's four
# hardware queues: 39.15 ms, slower than none), on the same companion (box-dependent: -0.15 ms on one, +0.7 on another), in the G phase only (neutral / +0.7); the
# companion's launch before the layer's data gradient instead of after it (38.3).  (Round 3 tried a LOW-PRIORITY side stream released by a per-layer event on the fp32 path and
# lost 0.3-1.2 ms; this form — normal priority, stream-wait on the chain, no per-layer join — wins there too.)
# DCV_NO_WGRAD_SIDE=1: in-stream, as before, on both paths (DCV_CL_NO_WGRAD_SIDE=1: on the 16-bit path only) — A/B.
# --------------------------------------------------------------------------- #
_WGRAD_SIDE = os.environ.get("DCV_NO_WGRAD_SIDE") is None
_side_streams = {}
_join_pending = set()


def wgrad_companion(device, w, enabled: bool = True):
    """The companion stream for parameter `w`'s weight gradient, or None: in-stream."""
    if not (_WGRAD_SIDE and enabled and _OWN_ACCUMULATION) or not isinstance(w, torch.nn.Parameter) or w._backward_hooks \
            or getattr(w, "_dcv_bucket", None) is not None or getattr(w, "_dcv_grad_slot", None) is not None:
        return None
    # The companion's result may only meet autograd where AccumulateGrad ADOPTS it without a kernel of its own (a kernel or hook on the chain's stream would read the
    # companion's data unordered): not under create_graph (it clones), not with post-accumulate hooks, not beside a .grad the in-place sum refuses (autograd would add).
    if torch.is_grad_enabled() or getattr(w, "_post_accumulate_grad_hooks", None):
        return None
    g = w.grad
    if g is not None and not (g.dtype == torch.float32 and g.is_cuda and g.is_contiguous() and g.shape == w.shape):
        return None
    cur = torch.cuda.current_stream(device)
    if cur.cuda_stream != torch.cuda.default_stream(device).cuda_stream:      # the discriminators' lanes already run beside one another
        return None
    s = _side_streams.get(device.index)
    if s is None:
        s = _side_streams[device.index] = torch.cuda.Stream(device)
    return s


def wgrad_join_at_end(device, cur, side) -> None:
    """Once per (backward pass, chain stream): when the engine has run the last node, the chain's stream — and the caller's — wait for the companion."""
    key = (device.index, cur.cuda_stream, _task_id())
    if key in _join_pending:
        return
    for k in [k for k in _join_pending if k[2] != key[2]]:      # a backward that raised never ran its callback: its keys go when the next pass registers
        _join_pending.discard(k)
    _join_pending.add(key)

    def join():
        # (the engine has already joined the leaf streams with the caller's ambient stream when the final callbacks run — under a guard that makes that ambient stream
        # current — so the caller's stream must wait for the companion itself, not only through the chain's stream)
        _join_pending.discard(key)
        cur.wait_stream(side)
        amb = torch.cuda.current_stream(device)
        if amb.cuda_stream != cur.cuda_stream:
            amb.wait_stream(side)
    torch.autograd.Variable._execution_engine.queue_callback(join)


_HEAD_BN_FUSION = os.environ.get("DCV_NO_HEAD_BN_FUSION") is None


class BnLink:
    """Ties the BatchNorm group that writes the FIRST channels of a concat buffer to the convolution that reads the buffer (the colour generator's last stage:
    UpBlock 5 -> cat with the stem's skip -> Outconv): `bn_act(..., link=)` notes what its backward needs, `conv(..., bn_link=)`'s backward then runs
    dcv_conv_backward_data_bn — the data gradient fused with that BatchNorm's backward — and leaves the BatchNorm's results here; the BatchNorm node of the SAME
    backward pass picks them up instead of reading its cotangent (whose memory the fused kernels never wrote).  Anything the fused entry point does not take
    (another geometry, eval mode, a dropout mask, the 16-bit path) leaves the link empty and both nodes run as they always did.  DCV_NO_HEAD_BN_FUSION=1: off (A/B)."""
    __slots__ = ("x", "gamma", "beta", "stats", "act", "slope", "task", "dx", "dgb", "stream", "defer", "deferred", "out", "_keep")

    def __init__(self, defer: bool = False):
        self.x = self.gamma = self.beta = self.stats = self.dx = self.dgb = self.out = None
        self.act, self.slope, self.task, self.stream = ACT_NONE, 0.0, -1, -1
        # `defer`: the BatchNorm group may leave its OUTPUT unwritten (statistics only) — the linked convolution's forward and weight gradient then read the BatchNorm
        # input and normalise + activate on load (dcv_conv_forward_bn / dcv_conv_backward_weight_bn); `deferred`: it did.  Only the caller knows that nothing else
        # reads that output (the generator: the up-path slice of the last concat buffer feeds the head alone)
        self.defer, self.deferred = bool(defer) and _HEAD_BN_FUSION and os.environ.get("DCV_NO_HEAD_BN_DEFER") is None, False

    def view_args(self):
        """The (cbn, bn_x, dims, gamma, beta, mean, invstd, act, slope) tail of the *_bn entry points."""
        bxd = dims5(self.x)
        self._keep = bxd
        return (self.x.shape[1], ptr(self.x), C.byref(bxd), ptr(self.gamma), ptr(self.beta), ptr(self.stats[0]), ptr(self.stats[1]), self.act, self.slope)


class _Conv(Function):
    # identity token of the current backward pass: a weight's gradient slot is handed out once per backward (new_backward_epoch() is called by
    # the optimiser wrapper's step(), i.e. between two backwards of the same bucket)
    _epoch = [object()]

    @staticmethod
    def forward(ctx, x, w, g: ConvGeom, act: int, slope: float, out=None, bn_stats=None, grad_slot=None, act_slot=None, bn_link=None, gate=None):
        N._require(x, "conv input"); N._require(w, "conv weight")
        ctx.grad_slot = grad_slot
        ctx.gate = gate if (_GATED_DGRAD and gate is not None and gate[1] is not None) else None
        ctx.bn_link = bn_link if _HEAD_BN_FUSION else None
        ctx.act_slot = act_slot if (act_slot is not None and act == ACT_LEAKY) else None
        if ctx.act_slot is not None:
            ctx.act_slot.act, ctx.act_slot.act_applied = (act, slope), False
        if x.shape[1] != g.cin:
            raise N.NativeError(f"conv: input has {x.shape[1]} channels, module expects {g.cin}")
        if not w.is_contiguous():
            raise N.NativeError("conv: weights must be contiguous (torch layout)")
        y = _dest(out, _out_shape(g, x), x.device)
        xd, yd = dims5(x), dims5(y)
        L = lib()
        need = L.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 0)
        if need == 0:
            raise N.NativeError("conv forward: " + L.dcv_last_error().decode())
        wsp, wsn = _ws("conv", need, x.device)
        ctx.pack = pc = _pack_of(w)
        pk = pc.get(w, 0, g, x, y, xd, yd) if pc is not None else None
        pkp = C.byref(pk) if pk is not None else None
        ctx.bn_deferred = False
        lk = ctx.bn_link
        if lk is not None and lk.deferred:
            # x's first channels were never written: the head's kernel reads the BatchNorm input and normalises on load — or, for a geometry it does not take,
            # the output is materialised now and everything proceeds as usual
            rc = L.dcv_conv_forward_bn(C.byref(g), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), act, slope, pkp, wsp, wsn, *lk.view_args(), stream_ptr())
            if rc == N.DCV_EUNSUPPORTED:
                od, bxd_ = dims5(lk.out), dims5(lk.x)
                check(L.dcv_bn_apply(ptr(lk.x), C.byref(bxd_), ptr(lk.out), C.byref(od), ptr(lk.gamma), ptr(lk.beta), ptr(lk.stats[0]), ptr(lk.stats[1]), None,
                                     lk.act, lk.slope, stream_ptr()), "dcv_bn_apply")
                lk.deferred = False
            else:
                check(rc, "dcv_conv_forward_bn")
                _PackCache.commit(pk)
                ctx.bn_deferred = True
                ctx.g, ctx.act, ctx.slope = g, act, slope
                ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
                return y
        sbytes = L.dcv_conv_stats_bytes(C.byref(g), C.byref(xd), C.byref(yd)) if (bn_stats is not None and act == ACT_NONE) else 0
        if sbytes:
            # conv -> BatchNorm pair: the epilogue leaves per-tile {sum, sum^2} of y, the BN op skips its pass over y
            stat = _empty((sbytes // 4,), x.device)     # poison runs turn a (class, tile) row no workgroup wrote into NaN statistics
            nparts, pitch = C.c_int(0), C.c_int(0)
            check(L.dcv_conv_forward_stats(C.byref(g), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), ptr(stat), sbytes,
                                           C.byref(nparts), C.byref(pitch), pkp, wsp, wsn, stream_ptr()), "dcv_conv_forward_stats")
            if nparts.value > 0:
                bn_stats.append((stat, nparts.value, pitch.value))
        else:
            check(L.dcv_conv_forward(C.byref(g), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), act, slope, pkp, wsp, wsn, stream_ptr()), "dcv_conv_forward")
        _PackCache.commit(pk)
        ctx.g, ctx.act, ctx.slope = g, act, slope
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        g = ctx.g
        L = lib()
        dy = _dense(dy)
        fused_away = ctx.act_slot is not None and bool(ctx.act_slot.act_applied)    # the consumer's data gradient already applied act'
        if ctx.act_slot is not None:
            ctx.act_slot.act_applied = max(0, int(ctx.act_slot.act_applied) - 1)      # (a count: a discriminator's two stems share one buffer and one consumer)
        if ctx.act != ACT_NONE and not fused_away:
            dz = _empty(y.shape, y.device)
            dyd, yd, dzd = dims5(dy), dims5(y), dims5(dz)   # y may be a strided concat-buffer slice; dz is dense
            check(L.dcv_act_backward(ptr(dy), C.byref(dyd), ptr(y), C.byref(yd), ptr(dz), C.byref(dzd), ctx.act, ctx.slope, stream_ptr()), "dcv_act_backward")
            dy = dz
        xd, dyd = dims5(x), dims5(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # x is a U-Net skip tensor whose OTHER consumer (the concat) has already delivered its gradient slice:
            # add this data gradient into that slice in the GEMM epilogue and hand autograd nothing to sum
            # (replaces a strided torch add over the tensor; the slice object is the one autograd holds)
            slot = ctx.grad_slot
            into = slot.take(x) if slot is not None else None
            dx = into if into is not None else _empty(x.shape, x.device)
            dxd = dims5(dx)
            need = L.dcv_conv_workspace_bytes(C.byref(g), C.byref(dxd), C.byref(dyd), 1)
            wsp, wsn = _ws("conv", need, x.device)
            pk = ctx.pack.get(w, 1, g, dx, dy, dxd, dyd) if ctx.pack is not None else None
            pkp = C.byref(pk) if pk is not None else None
            rc = N.DCV_EUNSUPPORTED
            link = ctx.bn_link
            if link is not None and link.x is not None and into is None and g.transposed and link.x.shape[0] == x.shape[0] and link.x.shape[2:] == x.shape[2:] \
                    and link.x.shape[1] < x.shape[1] and torch.cuda.current_stream(x.device).cuda_stream == link.stream:      # (the BatchNorm ran on this stream too)
                # the first channels of x are a BatchNorm group's output: the data gradient fused with that BatchNorm's backward (BnLink)
                bx = link.x
                cbn = bx.shape[1]
                bdx = _empty(bx.shape, bx.device)
                dgb = _empty((2, cbn), bx.device)
                bxd, bdxd = dims5(bx), dims5(bdx)
                need2 = L.dcv_conv_backward_data_bn_workspace_bytes(C.byref(dxd), cbn)
                ws2p, ws2n = _ws("headbn", need2, x.device)
                fused = C.c_int(0)
                rc = L.dcv_conv_backward_data_bn(C.byref(g), ptr(dy), C.byref(dyd), ptr(w), ptr(dx), C.byref(dxd), pkp, wsp, wsn, cbn, ptr(bx), C.byref(bxd),
                                                 ptr(link.gamma), ptr(link.beta), ptr(link.stats[0]), ptr(link.stats[1]), link.act, link.slope,
                                                 ptr(bdx), C.byref(bdxd), ptr(dgb[0]), ptr(dgb[1]), ws2p, ws2n, C.byref(fused), stream_ptr())
                if rc == N.DCV_EUNSUPPORTED:
                    pass                      # (nothing ran: the plain call below)
                else:
                    check(rc, "dcv_conv_backward_data_bn")
                    if fused.value:
                        link.dx, link.dgb, link.task = bdx, dgb, _task_id()
            if rc == N.DCV_EUNSUPPORTED and into is not None and slot.act is not None and _GATED_DGRAD and tuple(x.stride()) == tuple(into.stride()):
                # ... and the derivative of the activation that produced x, read off x, in the same epilogue
                xd_ = dims5(x)
                rc = L.dcv_conv_backward_data_gated(C.byref(g), ptr(dy), C.byref(dyd), ptr(w), ptr(dx), C.byref(dxd), 1, ptr(x), C.byref(xd_),
                                                    slot.act[0], slot.act[1], pkp, wsp, wsn, stream_ptr())
                if rc == 0:
                    slot.act_applied = True
                elif rc != N.DCV_EUNSUPPORTED:
                    check(rc, "dcv_conv_backward_data_gated")
            if rc == N.DCV_EUNSUPPORTED and ctx.gate is not None and into is None:
                # x is (a noisy copy of) a discriminator's concat buffer whose two halves are conv + LeakyReLU stems (discriminator.py:122-124): the stems'
                # activation derivative, read off the buffer, in this data gradient's epilogue — the stems then skip their own derivative passes
                gbuf, gslot = ctx.gate
                if gslot.act is not None and tuple(gbuf.shape) == tuple(dx.shape) and tuple(gbuf.stride()) == tuple(dx.stride()):
                    gd_ = dims5(gbuf)
                    rc = L.dcv_conv_backward_data_gated(C.byref(g), ptr(dy), C.byref(dyd), ptr(w), ptr(dx), C.byref(dxd), 0, ptr(gbuf), C.byref(gd_),
                                                        gslot.act[0], gslot.act[1], pkp, wsp, wsn, stream_ptr())
                    if rc == 0:
                        gslot.act_applied = 2
                    elif rc != N.DCV_EUNSUPPORTED:
                        check(rc, "dcv_conv_backward_data_gated")
            if rc == N.DCV_EUNSUPPORTED:
                check(L.dcv_conv_backward_data(C.byref(g), ptr(dy), C.byref(dyd), ptr(w), ptr(dx), C.byref(dxd), int(into is not None),
                                               pkp, wsp, wsn, stream_ptr()), "dcv_conv_backward_data")
            _PackCache.commit(pk)
            if into is not None:
                dx = None
        side = wgrad_companion(x.device, w) if ctx.needs_input_grad[1] else None
        if side is not None:      # the main chain's weight gradient on the companion stream (above)
            cur = torch.cuda.current_stream(x.device)
            side.wait_stream(cur)
            for t in (x, dy):
                t.record_stream(side)
            with torch.cuda.stream(side):
                need = L.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(dyd), 2)
                wsp, wsn = _ws("conv", need, x.device)
                tgt = grad_target(w)
                if ctx.bn_deferred:      # the operand's first channels exist only as the BatchNorm input: normalise on load
                    if tgt is None:
                        dw = _empty(w.shape, w.device)
                    check(L.dcv_conv_backward_weight_bn(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(dyd), C.c_void_p(tgt) if tgt is not None else ptr(dw),
                                                        int(tgt is not None), wsp, wsn, *ctx.bn_link.view_args(), stream_ptr()), "dcv_conv_backward_weight_bn")
                    if tgt is None:
                        note_first(w, dw)
                elif tgt is not None:
                    check(L.dcv_conv_backward_weight_acc(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(dyd), C.c_void_p(tgt), 1, wsp, wsn, stream_ptr()),
                          "dcv_conv_backward_weight_acc")
                else:
                    dw = _empty(w.shape, w.device)
                    check(L.dcv_conv_backward_weight(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(dyd), ptr(dw), wsp, wsn, stream_ptr()), "dcv_conv_backward_weight")
                    note_first(w, dw)
            wgrad_join_at_end(x.device, cur, side)
        elif ctx.needs_input_grad[1]:
            # data parallel: the parameter's slice of its bucket's flat gradient buffer (optim.GradBucket) — the first weight gradient of a
            # backward is written straight into it (autograd adopts the returned tensor as .grad); a second use of the same weight in one
            # backward (D on the real and the fake batch) gets a tensor of its own, which autograd adds
            need = L.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(dyd), 2)
            wsp, wsn = _ws("conv", need, x.device)
            tgt = grad_target(w) if _OWN_ACCUMULATION else None
            if ctx.bn_deferred:
                if tgt is None:
                    dw = _empty(w.shape, w.device)
                check(L.dcv_conv_backward_weight_bn(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(dyd), C.c_void_p(tgt) if tgt is not None else ptr(dw),
                                                    int(tgt is not None), wsp, wsn, *ctx.bn_link.view_args(), stream_ptr()), "dcv_conv_backward_weight_bn")
                if tgt is None and _OWN_ACCUMULATION:
                    note_first(w, dw)
            elif tgt is not None:
                # a later contribution (second use in this backward, or .grad from an earlier backward): the slab reduce adds into the tensor that holds the first
                check(L.dcv_conv_backward_weight_acc(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(dyd), C.c_void_p(tgt), 1, wsp, wsn, stream_ptr()),
                      "dcv_conv_backward_weight_acc")
            else:
                slot = getattr(w, "_dcv_grad_slot", None)
                if slot is not None and w.grad is None and getattr(w, "_dcv_slot_epoch", None) is not _Conv._epoch[0]:
                    w._dcv_slot_epoch = _Conv._epoch[0]
                    b = getattr(w, "_dcv_bucket", None)
                    if b is not None and b() is not None:
                        b().before_slot_write(w)      # a collective of the previous backward may still be reading this slice (GradBucket(overlap=True))
                    dw = slot.detach()
                else:
                    dw = _empty(w.shape, w.device)
                check(L.dcv_conv_backward_weight(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(dyd), ptr(dw), wsp, wsn, stream_ptr()), "dcv_conv_backward_weight")
                if _OWN_ACCUMULATION:
                    note_first(w, dw)
        return dx, dw, None, None, None, None, None, None, None, None, None


def new_backward_epoch():
    """Called when the gradients of the last backward have been consumed (optimiser step): weight-gradient slots may be handed out again."""
    _Conv._epoch[0] = object()


def conv(x, w, g: ConvGeom, act: int = ACT_NONE, slope: float = 0.0, out=None, bn_stats=None, grad_slot=None, act_slot=None, bn_link=None, gate=None):
    """y = act(conv(x, w)) for nn.Conv2d / nn.Conv3d / nn.ConvTranspose2d geometries.
    `out`: optional destination view (e.g. a channel slice of a concat buffer) to write into.
    `grad_slot`: ConcatBuffer.slot of the buffer whose second slice IS x (a skip connection), see GradSlot.
    `act_slot`: ConcatBuffer.slot of the buffer this conv + (Leaky)ReLU writes its output into (`out` is its second slice).
    `bn_link`: the BnLink of the BatchNorm group that produced x's first channels (see BnLink)."""
    return _Conv.apply(x, w, g, act, float(slope), None if out is None else _Out(out), bn_stats, grad_slot, act_slot, bn_link, gate)


# --------------------------------------------------------------------------- #
# BatchNorm (+ Dropout2d mask) + activation
# --------------------------------------------------------------------------- #
class _BnAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, mask, training: bool, momentum: float, eps: float, act: int, slope: float, out=None,
                partials=None, nbt=None, link=None):
        N._require(x, "bn input")
        ctx.link = link
        L = lib()
        Cn = x.shape[1]
        y = _dest(out, x.shape, x.device)
        stats = _empty((2, Cn), x.device)
        xd, yd = dims5(x), dims5(y)
        wsp, wsn = _ws("bn", L.dcv_bn_workspace_bytes(Cn), x.device)
        if link is not None:
            link.deferred = False
        if link is not None and link.defer and partials is not None and training and mask is None and act in (ACT_NONE, ACT_LEAKY) and x.dim() == 4 and x.is_contiguous() \
                and x.shape[3] == 64 and x.shape[1] % 32 == 0:
            # statistics only: the output slice stays unwritten, the linked head normalises on load (BnLink.defer)
            stat, nparts, pitch = partials.v
            check(L.dcv_bn_forward_stats_only(ptr(x), C.byref(xd), ptr(running_mean), ptr(running_var), ptr(nbt), ptr(stats[0]), ptr(stats[1]), momentum, eps,
                                              ptr(stat), nparts, pitch, stream_ptr()), "dcv_bn_forward_stats_only")
            link.deferred, link.out = True, y.detach()      # an alias WITHOUT this node as its grad_fn: link -> y -> grad_fn (this ctx) -> link would be a reference cycle that
                                                            # keeps x and y (1.17 GB each at B = 70) allocated until Python's cyclic collector happens to run (tools/cycle_probe.py)
        elif partials is not None and training:
            stat, nparts, pitch = partials.v
            check(L.dcv_bn_act_forward_stats(ptr(x), C.byref(xd), ptr(y), C.byref(yd), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(nbt),
                                             ptr(stats[0]), ptr(stats[1]), ptr(mask), momentum, eps, act, slope, ptr(stat), nparts, pitch,
                                             wsp, wsn, stream_ptr()), "dcv_bn_act_forward_stats")
        else:
            check(L.dcv_bn_act_forward(ptr(x), C.byref(xd), ptr(y), C.byref(yd), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                       ptr(nbt) if training else None, ptr(stats[0]), ptr(stats[1]), ptr(mask), int(training), momentum, eps, act, slope, wsp, wsn, stream_ptr()),
                  "dcv_bn_act_forward")
        ctx.cfg = (bool(training), act, slope)
        ctx.save_for_backward(x, gamma, beta, stats, mask)
        if link is not None:
            if training and mask is None and act in (ACT_NONE, ACT_LEAKY) and x.dim() == 4 and x.is_contiguous():
                link.x, link.gamma, link.beta, link.stats, link.act, link.slope = x, gamma, beta, stats, act, slope
                link.stream = torch.cuda.current_stream(x.device).cuda_stream
            else:
                link.x = None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats, mask = ctx.saved_tensors
        training, act, slope = ctx.cfg
        link = ctx.link
        if link is not None and link.dx is not None and link.task == _task_id():
            # the consumer convolution's backward has already done this node's work (dcv_conv_backward_data_bn); dy's memory was never written
            dx, dgb = link.dx, link.dgb
            link.dx = link.dgb = None
            if _OWN_ACCUMULATION:
                return (dx, deliver_small(gamma, dgb[0]), deliver_small(beta, dgb[1])) + (None,) * 12
            return (dx, dgb[0], dgb[1]) + (None,) * 12
        L = lib()
        dy = _dense(dy)
        Cn = x.shape[1]
        dx = _empty(x.shape, x.device)
        dgb = _empty((2, Cn), x.device)
        dyd, xd, dxd = dims5(dy), dims5(x), dims5(dx)
        wsp, wsn = _ws("bn", L.dcv_bn_workspace_bytes(Cn), x.device)
        check(L.dcv_bn_act_backward(ptr(dy), C.byref(dyd), ptr(x), C.byref(xd), ptr(dx), C.byref(dxd), ptr(gamma), ptr(beta),
                                    ptr(stats[0]), ptr(stats[1]), ptr(mask), int(training), act, slope, ptr(dgb[0]), ptr(dgb[1]), wsp, wsn, stream_ptr()),
              "dcv_bn_act_backward")
        if _OWN_ACCUMULATION:
            return (dx, deliver_small(gamma, dgb[0]), deliver_small(beta, dgb[1])) + (None,) * 12
        return (dx, dgb[0], dgb[1]) + (None,) * 12


def bn_act(x, gamma, beta, running_mean, running_var, training: bool, act: int = ACT_NONE, slope: float = 0.0,
           mask: Optional[torch.Tensor] = None, momentum: float = 0.1, eps: float = 1e-5, out=None, partials=None, num_batches_tracked=None, link=None):
    """y = act(mask * batch_norm(x)); running stats (and the int64 `num_batches_tracked` buffer, when given) are
    updated in place when training.
    `partials`: (buffer, nparts, pitch) left by the producing conv's epilogue (ops.conv(..., bn_stats=[]))."""
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise N.NativeError("bn_act: num_batches_tracked must be an int64 device tensor")
    return _BnAct.apply(x, gamma, beta, running_mean, running_var, mask, training, float(momentum), float(eps), act, float(slope),
                        None if out is None else _Out(out), None if partials is None else _Opaque(partials), num_batches_tracked, link)


# --------------------------------------------------------------------------- #
# activation
# --------------------------------------------------------------------------- #
class _Act(Function):
    @staticmethod
    def forward(ctx, x, act: int, slope: float):
        N._require(x, "activation input")
        y = _empty(x.shape, x.device)
        xd, yd = dims5(x), dims5(y)
        check(lib().dcv_act_forward(ptr(x), C.byref(xd), ptr(y), C.byref(yd), act, slope, stream_ptr()), "dcv_act_forward")
        ctx.cfg = (act, slope)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        act, slope = ctx.cfg
        dy = _dense(dy)
        dx = _empty(y.shape, y.device)
        dyd, yd, dxd = dims5(dy), dims5(y), dims5(dx)
        check(lib().dcv_act_backward(ptr(dy), C.byref(dyd), ptr(y), C.byref(yd), ptr(dx), C.byref(dxd), act, slope, stream_ptr()), "dcv_act_backward")
        return dx, None, None


def act(x, kind: int, slope: float = 0.0):
    return _Act.apply(x, kind, float(slope))


# --------------------------------------------------------------------------- #
# segmentation branch: softmax head, argmax -> {-1,+1} maps
# --------------------------------------------------------------------------- #
class _SoftmaxChannels(Function):
    @staticmethod
    def forward(ctx, x):
        N._require(x, "softmax input")
        y = _empty(x.shape, x.device)
        xd, yd = dims5(x), dims5(y)
        check(lib().dcv_softmax_channels_forward(ptr(x), C.byref(xd), ptr(y), C.byref(yd), stream_ptr()), "dcv_softmax_channels_forward")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = _empty(y.shape, y.device)
        dyd, yd, dxd = dims5(dy), dims5(y), dims5(dx)   # dy may be a strided view (the (B,T,C,H,W) video layout)
        check(lib().dcv_softmax_channels_backward(ptr(dy), C.byref(dyd), ptr(y), C.byref(yd), ptr(dx), C.byref(dxd), stream_ptr()), "dcv_softmax_channels_backward")
        return dx


def softmax_channels(x):
    """nn.Softmax(dim=1) of the geometry generator's segmentation head (generator.py:75-76)."""
    return _SoftmaxChannels.apply(x)


def segm_onehot(x):
    """One-hot / softmax maps -> {-1,+1} maps by channel argmax (generator.py:378-385); like the reference's
    argmax + scatter_ it passes no gradient."""
    N._require(x, "segmentation maps")
    with torch.no_grad():
        y = _empty(x.shape, x.device)
        xd, yd = dims5(x), dims5(y)
        check(lib().dcv_segm_onehot(ptr(x), C.byref(xd), ptr(y), C.byref(yd), stream_ptr()), "dcv_segm_onehot")
    return y


# --------------------------------------------------------------------------- #
# axpby-based pieces: noise add, channel concat, temporal difference
# --------------------------------------------------------------------------- #
def _axpby(x, a, z, b, out):
    xd, od = dims5(x), dims5(out)
    zd = dims5(z) if z is not None else None
    check(lib().dcv_axpby(ptr(x), C.byref(xd), float(a), ptr(z), C.byref(zd) if z is not None else None, float(b), ptr(out), C.byref(od), stream_ptr()), "dcv_axpby")
    return out


class _NoiseAdd(Function):
    @staticmethod
    def forward(ctx, x, sigma: float, sample, seed: int, offset: int):
        N._require(x, "noise input")
        y = _empty(x.shape, x.device)
        if sample is not None:  # injected draw (parity tests)
            _axpby(x, 1.0, sample, sigma, y)
        else:
            xd, yd = dims5(x), dims5(y)
            check(lib().dcv_noise_add(ptr(x), C.byref(xd), ptr(y), C.byref(yd), sigma, seed, offset, stream_ptr()), "dcv_noise_add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None, None, None


def noise_add(x, sigma: float, sample=None, seed: int = 0, offset: int = 0):
    """x + sigma * N(0,1): device Philox draw, or an injected `sample` tensor."""
    if x.dtype in (torch.bfloat16, torch.float16):       # 16-bit channels-last data path
        from . import ops_cl
        return ops_cl.noise_add(x, sigma, sample, seed, offset)
    return _NoiseAdd.apply(x, float(sigma), sample, int(seed), int(offset))


class _CatChannels(Function):
    @staticmethod
    def forward(ctx, a, b):
        N._require(a, "cat input"); N._require(b, "cat input")
        ca, cb = a.shape[1], b.shape[1]
        out = _empty((a.shape[0], ca + cb) + tuple(a.shape[2:]), a.device)
        _axpby(a, 1.0, None, 0.0, out[:, :ca])
        _axpby(b, 1.0, None, 0.0, out[:, ca:])
        ctx.ca = ca
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy[:, :ctx.ca], dy[:, ctx.ca:]


def cat_channels(a, b):
    """torch.cat([a, b], 1) as two strided copies into channel slices."""
    return _CatChannels.apply(a, b)


class _CopyInto(Function):
    @staticmethod
    def forward(ctx, x, holder):
        dst = holder.t
        _axpby(x, 1.0, None, 0.0, dst)
        return dst.detach()

    @staticmethod
    def backward(ctx, dy):
        return dy, None


def copy_into(x, dst):
    """Strided copy of x into the view dst (a concat-buffer slice); gradient passes through."""
    return _CopyInto.apply(x, _Out(dst))


class GradSlot:
    """Meeting point of the two gradients of a skip tensor (second slice of a ConcatBuffer): the concat's backward
    runs first (it was recorded later) and leaves its slice here; the data gradient of the tensor's other consumer
    then accumulates into that slice (`dcv_conv_backward_data(accumulate=1)`) instead of producing a second tensor
    for autograd to add.  If the order is ever different the slot is simply empty and nothing changes."""
    __slots__ = ("g", "act", "act_applied")

    def __init__(self):
        self.g = None
        # set by the conv + (Leaky)ReLU that PRODUCED the skip tensor (ops.conv(..., act_slot=)): its activation, whose
        # derivative the consumer's data gradient can apply from the tensor itself (dcv_conv_backward_data_gated) — the
        # producer then skips its own derivative pass (act_applied)
        self.act = None
        self.act_applied = False

    def take(self, x):
        g, self.g = self.g, None
        if g is None or tuple(g.shape) != tuple(x.shape) or g.device != x.device:
            return None
        return g


class _JoinSlices(Function):
    """cat([a, b], 1) when a and b were WRITTEN INTO adjacent channel slices of `buf` by their
    producers (conv / bn_act with out=): nothing to copy, the result is `buf` itself."""

    @staticmethod
    def forward(ctx, a, b, holder, slot=None):
        ctx.slot = slot
        buf = holder.t
        ca, cb = a.shape[1], b.shape[1]
        if buf.shape[1] != ca + cb or a.data_ptr() != buf.data_ptr() or b.data_ptr() != buf[:, ca:].data_ptr() \
                or a.stride() != buf.stride() or b.stride() != buf.stride():
            raise N.NativeError("join_slices: operands are not the two channel slices of the buffer")
        ctx.ca = ca
        return buf.detach()

    @staticmethod
    def backward(ctx, dy):
        second = dy[:, ctx.ca:]
        if ctx.slot is not None:
            ctx.slot.g = second
        return dy[:, :ctx.ca], second, None, None


class ConcatBuffer:
    """A (N, Ca+Cb, ...) buffer whose two channel slices are handed to the producers as out= views."""

    def __init__(self, n, ca, cb, spatial, device):
        self.buf = _empty((n, ca + cb) + tuple(spatial), device)
        self.first, self.second = self.buf[:, :ca], self.buf[:, ca:]
        self.slot = GradSlot() if _SKIP_ACCUMULATE else None

    def join(self, a, b):
        return _JoinSlices.apply(a, b, _Out(self.buf), self.slot)


class _TemporalDiff(Function):
    @staticmethod
    def forward(ctx, x):
        N._require(x, "temporal_diff input")
        L = x.shape[2]
        y = _empty((x.shape[0], x.shape[1], L - 1) + tuple(x.shape[3:]), x.device)
        _axpby(x[:, :, 1:L], 1.0, x[:, :, 0:L - 1], -1.0, y)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _dense(dy)
        L = ctx.shape[2]
        dx = _empty(ctx.shape, dy.device)
        _axpby(dy[:, :, 0:1], -1.0, None, 0.0, dx[:, :, 0:1])
        _axpby(dy[:, :, L - 2:L - 1], 1.0, None, 0.0, dx[:, :, L - 1:L])
        if L > 2:
            _axpby(dy[:, :, 0:L - 2], 1.0, dy[:, :, 1:L - 1], -1.0, dx[:, :, 1:L - 1])
        return dx


def temporal_diff(x):
    """x[:, :, 1:] - x[:, :, :-1]  (discriminator.py:330-331)."""
    return _TemporalDiff.apply(x)


# --------------------------------------------------------------------------- #
# GAN losses
# --------------------------------------------------------------------------- #
KIND_BCE_ONES, KIND_BCE_ZEROS, KIND_HINGE_REAL, KIND_HINGE_FAKE, KIND_SOFTPLUS_NEG = range(5)


class _GanLoss(Function):
    """Sum of GAN loss terms (one fused value + gradient launch per term, the value accumulated on the device: `l_real + l_fake` of loss.py:99,131,164 without
    a torch add); backward: each term's stored gradient times the upstream 0-d cotangent, read on the device (dcv_scale_dev)."""

    @staticmethod
    def forward(ctx, kinds, *ys):
        out = _empty((), ys[0].device)
        dys = []
        for i, (y, kind) in enumerate(zip(ys, kinds.v)):
            N._require(y, "loss input")
            yc = y.contiguous()
            dy = _empty(yc.shape, yc.device)
            check(lib().dcv_gan_loss(ptr(yc), yc.numel(), int(kind), ptr(out), int(i > 0), ptr(dy), stream_ptr()), "dcv_gan_loss")
            dys.append(dy)
        ctx.save_for_backward(*dys)
        ctx.shapes = [tuple(y.shape) for y in ys]
        return out

    @staticmethod
    def backward(ctx, g):
        if not (g.is_cuda and g.dtype == torch.float32):
            raise N.NativeError("gan_loss backward: the upstream cotangent must be an fp32 device tensor")
        g = g.reshape(())
        outs = []
        for dy, shape, need in zip(ctx.saved_tensors, ctx.shapes, ctx.needs_input_grad[1:]):
            if not need:
                outs.append(None)
                continue
            dx = _empty(dy.shape, dy.device)
            check(lib().dcv_scale_dev(ptr(dy), dy.numel(), ptr(g), ptr(dx), stream_ptr()), "dcv_scale_dev")
            outs.append(dx.view(shape))
        return (None,) + tuple(outs)


def gan_loss(y, kind: int):
    return _GanLoss.apply(_Opaque((kind,)), y)


def gan_loss_sum(terms):
    """sum_i loss(y_i, kind_i) for [(y, kind), ...], summed on the device in list order."""
    return _GanLoss.apply(_Opaque(tuple(k for _, k in terms)), *[y for y, _ in terms])


class _SumScalars(Function):
    """((x0 + x1) + x2) ... of 0-d tensors by dcv_axpby — `loss_idis + loss_vdis + loss_gdis` (trainer.py:315) without torch's add kernels; every term receives
    the upstream cotangent itself."""

    @staticmethod
    def forward(ctx, *xs):
        for x in xs:
            N._require(x, "sum_scalars operand")
        acc = xs[0]
        for x in xs[1:]:
            out = _empty((), x.device)      # (0-d and nobody's view: the trainer calls detach_() on the result, trainer.py:324)
            _axpby(acc.reshape(1, 1), 1.0, x.reshape(1, 1), 1.0, out.view(1, 1))
            acc = out
        ctx.n = len(xs)
        return acc if len(xs) > 1 else acc.clone()

    @staticmethod
    def backward(ctx, g):
        return (g,) * ctx.n


def sum_scalars(*xs):
    return _SumScalars.apply(*xs)


class _FanOut(Function):
    """One tensor read by several consumers — the fake clips feed the image discriminator (frame `t`), the video and the gradient discriminator, and the
    geometry clip the colour generator as well (trainer.py:303-309,346-349): the outputs are views of x (frame x[:, :, t] first, then `n_full` whole-tensor
    aliases), and the backward forms the ONE gradient of x with dcv_axpby — whole-tensor cotangents in output order, then the frame's into its slice — where
    autograd would launch a torch add per extra consumer plus the zero-fill and copy of the slice's backward."""

    @staticmethod
    def forward(ctx, x, t, n_full):
        N._require(x, "fan_out input")
        ctx.t, ctx.shape, ctx.strides = int(t), tuple(x.shape), tuple(x.stride())      # the generators' clips are (B, T, C, H, W) memory viewed as (B, C, T, H, W)
        order = sorted(range(x.dim()), key=lambda i: -ctx.strides[i])
        ctx.dense = x.permute(*order).is_contiguous()
        ctx.set_materialize_grads(False)
        return (x[:, :, ctx.t],) + tuple(x.view_as(x) for _ in range(int(n_full)))

    @staticmethod
    def backward(ctx, g_frame, *g_full):
        gs = [_dense(g) for g in g_full if g is not None]
        if not gs and g_frame is None:
            return None, None, None
        dev = (gs[0] if gs else g_frame).device
        # x's own layout when it is a dense permutation (the view chain back into the generator then needs no copy), else contiguous
        tot = torch.empty_strided(ctx.shape, ctx.strides, dtype=torch.float32, device=dev) if ctx.dense else _empty(ctx.shape, dev)
        if not gs:
            raise N.NativeError("fan_out: only the frame consumer delivered a gradient (no whole-tensor consumer): not a case of the training iteration")
        if len(gs) == 1:
            _axpby(gs[0], 1.0, None, 0.0, tot)
        else:
            _axpby(gs[0], 1.0, gs[1], 1.0, tot)
            for g in gs[2:]:
                _axpby(tot, 1.0, g, 1.0, tot)
        if g_frame is not None:
            sl = tot[:, :, ctx.t]
            _axpby(sl, 1.0, _dense(g_frame), 1.0, sl)
        return tot, None, None


def fan_out(x, t: int, n_full: int):
    """(x[:, :, t], x, x, ...) with `n_full` whole-tensor aliases; see _FanOut."""
    return _FanOut.apply(x, int(t), int(n_full))


def tile_rows(z, reps: int):
    """(B, C) -> (B * reps, C): every row `reps` times (`zc.repeat(1, T).view(B * T, C)`, generator.py:85-91,356-360) as ONE strided dcv_axpby read of a
    broadcast view — z carries no gradient (a latent draw)."""
    N._require(z, "tile_rows input")
    B, Cc = z.shape
    out = _empty((B * reps, Cc), z.device)
    _axpby(z.view(B, 1, 1, 1, Cc).expand(B, reps, 1, 1, Cc), 1.0, None, 0.0, out.view(B, reps, 1, 1, Cc))
    return out


# --------------------------------------------------------------------------- #
# GRU recurrence
# --------------------------------------------------------------------------- #
class _GruSeq(Function):
    @staticmethod
    def forward(ctx, e, h0, w_ih, w_hh, b_ih, b_hh):
        for t in (e, h0, w_ih, w_hh, b_ih, b_hh):
            N._require(t, "gru operand")
        T, B, dm = e.shape
        e, h0 = e.contiguous(), h0.contiguous()
        w_ih, w_hh, b_ih, b_hh = w_ih.contiguous(), w_hh.contiguous(), b_ih.contiguous(), b_hh.contiguous()
        out = _empty((B, T, dm), e.device)
        gates = _empty((T, B, 4 * dm), e.device)
        check(lib().dcv_gru_forward(ptr(e), ptr(h0), ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh), ptr(out), ptr(gates), T, B, dm, stream_ptr()), "dcv_gru_forward")
        ctx.save_for_backward(e, h0, out, gates, w_ih, w_hh, b_ih, b_hh)
        return out

    @staticmethod
    def backward(ctx, dout):
        e, h0, out, gates, w_ih, w_hh, b_ih, b_hh = ctx.saved_tensors
        T, B, dm = e.shape
        if not dout.is_contiguous():      # a channel slice of the latent's gradient (cat_channels): the library's strided copy, not torch's
            src = dout.reshape(T * B, dm)      # (a view: the slice is row-strided)
            dense = _empty((B, T, dm), e.device)
            _axpby(src, 1.0, None, 0.0, dense.view(T * B, dm))
            dout = dense
        dw_ih, dw_hh = _empty(w_ih.shape, w_ih.device), _empty(w_hh.shape, w_hh.device)
        db_ih = _empty((3 * dm,), e.device)
        db_hh = _empty((3 * dm,), e.device)
        L = lib()
        wsp, wsn = _ws("gru", L.dcv_gru_workspace_bytes(B, dm), e.device)
        check(L.dcv_gru_backward(ptr(dout), ptr(e), ptr(h0), ptr(out), ptr(gates), ptr(w_ih), ptr(w_hh), ptr(dw_ih), ptr(dw_hh), ptr(db_ih), ptr(db_hh),
                                 T, B, dm, wsp, wsn, stream_ptr()), "dcv_gru_backward")
        # second and later contributions (the dead D-phase backward adds to the last G phase's gradients) by dcv_axpby, not by autograd's torch add
        if not _OWN_ACCUMULATION:
            return None, None, dw_ih, dw_hh, db_ih, db_hh
        return (None, None) + tuple(deliver_small(p, t) for p, t in zip((w_ih, w_hh, b_ih, b_hh), (dw_ih, dw_hh, db_ih, db_hh)))


def gru_sequence(e, h0, w_ih, w_hh, b_ih, b_hh):
    """(T,B,dm) inputs, (B,dm) initial state -> (B,T,dm) hidden states (nn.GRUCell unrolled)."""
    return _GruSeq.apply(e, h0, w_ih, w_hh, b_ih, b_hh)


# --------------------------------------------------------------------------- #
# random draws on the device
# --------------------------------------------------------------------------- #
def normal(shape, device, seed: int, offset: int) -> torch.Tensor:
    out = _empty(tuple(shape), device)
    N._require(out, "normal() output")
    check(lib().dcv_normal_fill(ptr(out), out.numel(), int(seed), int(offset), stream_ptr()), "dcv_normal_fill")
    return out


def normal_many(count: int, shape, device, seed: int, offset: int) -> torch.Tensor:
    """(count, *shape): draw j = normal(shape, device, seed, offset + j), in one launch."""
    out = torch.empty((int(count),) + tuple(shape), dtype=torch.float32, device=device)
    if out.numel():
        check(lib().dcv_normal_fill_many(ptr(out), out.numel() // int(count), int(count), int(seed), int(offset), stream_ptr()), "dcv_normal_fill_many")
    return out


def dropout2d_mask(n: int, c: int, p: float, device, seed: int, offset: int) -> torch.Tensor:
    out = _empty((n, c, 1, 1), device)
    N._require(out, "dropout2d_mask() output")
    check(lib().dcv_dropout_mask(ptr(out), out.numel(), float(p), int(seed), int(offset), stream_ptr()), "dcv_dropout_mask")
    return out
