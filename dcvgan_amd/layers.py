"""Executes torch.nn containers on the HIP kernels.

The generators/discriminators keep their parameters in genuine nn.Conv2d /
nn.ConvTranspose2d / nn.Conv3d / nn.BatchNorm{2,3}d / nn.GRUCell objects inside
nn.Sequential containers, so ``state_dict`` keys, ``.apply(init_weights)``,
``.to()``, pickling and torch optimisers behave exactly as with the reference.
Only *execution* differs: ``run`` walks a Sequential and launches, per group of
layers, the fused HIP op that implements it:

    Noise                        -> noise_add   (Philox or injected sample)
    Conv* [+ (Leaky)ReLU | Tanh] -> conv        (activation fused in the epilogue)
    BatchNorm [+ Dropout2d] [+ (Leaky)ReLU] -> bn_act
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops, ops_cl

import os

_FUSE_BN_STATS = os.environ.get("DCV_NO_BN_FUSION") is None
# Test instrumentation: when a list is installed here, every fused (Leaky)ReLU appends the branch pattern `y > 0` of
# its output, in execution order — the pattern the backward kernels differentiate with (tests/test_fullwidth_gpu.py
# replays it in an fp64 evaluation of the same graph on the host).  None in production: nothing is recorded.
KINK_TAP = None
# Diagnostic of the 16-bit paths (bench.py's stress leg, tests/test_fp16_gpu.py): when a list is installed here, every convolution output that feeds a BatchNorm
# appends its largest magnitude (a 0-d device tensor) — the un-normalised sums that decide whether fp16's 65504 is enough.  None in production.
PREBN_TAP = None


def _tap(y, fused):
    if KINK_TAP is not None and fused is not None and fused[0] == ops.ACT_LEAKY:
        KINK_TAP.append((y.detach() > 0).cpu())
_CONVS = (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d)
_BNS = (nn.BatchNorm2d, nn.BatchNorm3d)


def _act_of(layer):
    """(code, slope) if `layer` is an activation the kernels can fuse, else None."""
    if isinstance(layer, nn.LeakyReLU):
        return ops.ACT_LEAKY, float(layer.negative_slope)
    if isinstance(layer, nn.ReLU):
        return ops.ACT_LEAKY, 0.0
    if isinstance(layer, nn.Tanh):
        return ops.ACT_TANH, 0.0
    return None


def geom_of(conv) -> "ops.ConvGeom":
    if conv.bias is not None:
        raise NotImplementedError("the DCVGAN layers are all bias-free")
    if any(d != 1 for d in conv.dilation) or conv.groups != 1:
        raise NotImplementedError("dilation/groups are not used by DCVGAN")
    if isinstance(conv, nn.ConvTranspose2d) and any(o != 0 for o in conv.output_padding):
        raise NotImplementedError("output_padding is not used by DCVGAN")
    return ops.conv_geom(conv.weight, conv.stride, conv.padding, isinstance(conv, nn.ConvTranspose2d), getattr(conv, "_dcv_precision", None))


def batch_norm(bn, x, rng, act=(ops.ACT_NONE, 0.0), dropout=None, out=None, partials=None, link=None):
    training = bn.training
    mask = None
    if dropout is not None and dropout.training:
        mask = rng.dropout2d_mask(x.shape[0], x.shape[1], dropout.p, x.device)
    if ops_cl.is_cl(x):     # bf16 channels-last data path (ops_cl): statistics from its own pass over the bf16 tensor
        return ops_cl.bn_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, act[0], act[1], mask,
                             bn.momentum if bn.momentum is not None else 0.1, bn.eps, out=out, num_batches_tracked=bn.num_batches_tracked if training else None,
                             partials=partials if training else None)
    # num_batches_tracked is bumped by the statistics kernel itself (one launch less per BatchNorm layer)
    return ops.bn_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, act[0], act[1], mask,
                      bn.momentum if bn.momentum is not None else 0.1, bn.eps, out=out, partials=partials if training else None,
                      num_batches_tracked=bn.num_batches_tracked if training else None, link=link)


def run(seq: nn.Sequential, x: torch.Tensor, rng, out=None, grad_slot=None, act_slot=None, bn_link=None, gate=None) -> torch.Tensor:
    """`out`: destination view for the sequence's LAST fused op (a concat-buffer slice), when that
    op is a conv(+act) or a BatchNorm group.  `grad_slot`: ops.GradSlot of the concat buffer that holds x
    (x is a skip tensor): passed to the FIRST convolution, whose data gradient then accumulates into it.
    `act_slot`: ops.GradSlot of the concat buffer `out` belongs to, for a conv + (Leaky)ReLU that ends the sequence.
    `bn_link`: ops.BnLink — given to the sequence's LAST BatchNorm group when it ends the sequence (the producer side), and to its FIRST convolution
    (the consumer side); fp32 path only.
    `gate`: (concat buffer, its ops.GradSlot) — given to the sequence's FIRST convolution when x is that buffer or a Noise layer's copy of it and the buffer's halves were
    written by conv + LeakyReLU stems with `act_slot=` that slot: the convolution's data gradient applies the stems' activation derivative (fp32 path only)."""
    layers = list(seq)
    first_conv = True
    i, n = 0, len(layers)
    pending = None   # BatchNorm partial sums left by the conv that produced x (conv -> BN pairs in training mode)
    while i < n:
        layer = layers[i]
        nxt = layers[i + 1] if i + 1 < n else None
        if isinstance(layer, _CONVS) and ops_cl.is_cl(x):
            # bf16 channels-last data path: the same grouping (conv [+ activation] in one launch), ops_cl's kernels
            fused = _act_of(nxt) if nxt is not None else None
            last = i + (2 if fused is not None else 1) >= n
            box = [] if (fused is None and isinstance(nxt, _BNS) and nxt.training and _FUSE_BN_STATS) else None
            x = ops_cl.conv(x, layer.weight, geom_of(layer), *(fused or (ops.ACT_NONE, 0.0)), out=out if last else None, grad_slot=grad_slot if i == 0 else None, bn_stats=box,
                            act_slot=act_slot if (last and fused is not None) else None)
            pending = box[0] if box else None
            if PREBN_TAP is not None and fused is None and isinstance(nxt, _BNS):
                PREBN_TAP.append(x.detach().float().abs().max())
            i += 2 if fused is not None else 1
        elif isinstance(layer, _CONVS):
            fused = _act_of(nxt) if nxt is not None else None
            if fused is not None:
                x = ops.conv(x, layer.weight, geom_of(layer), fused[0], fused[1], out=out if i + 2 >= n else None, grad_slot=grad_slot if i == 0 else None,
                             act_slot=act_slot if i + 2 >= n else None, bn_link=bn_link if i == 0 else None, gate=gate if first_conv else None)
                first_conv = False
                _tap(x, fused)
                i += 2
            else:
                box = [] if (isinstance(nxt, _BNS) and nxt.training and _FUSE_BN_STATS) else None
                x = ops.conv(x, layer.weight, geom_of(layer), out=out if i + 1 >= n else None, bn_stats=box, grad_slot=grad_slot if i == 0 else None,
                             gate=gate if first_conv else None)
                first_conv = False
                pending = box[0] if box else None
                i += 1
        elif isinstance(layer, _BNS):
            j = i + 1
            drop = None
            if j < n and isinstance(layers[j], nn.Dropout2d):
                drop = layers[j]
                j += 1
            fused = _act_of(layers[j]) if j < n else None
            if fused is not None:
                j += 1
            x = batch_norm(layer, x, rng, fused or (ops.ACT_NONE, 0.0), drop, out=out if j >= n else None, partials=pending,
                           link=bn_link if (j >= n and not ops_cl.is_cl(x)) else None)
            _tap(x, fused)
            pending = None
            i = j
        elif _act_of(layer) is not None:
            code, slope = _act_of(layer)
            x = ops_cl.act(x, code, slope) if ops_cl.is_cl(x) else ops.act(x, code, slope)
            _tap(x, (code, slope))
            i += 1
        elif hasattr(layer, "use_noise") and hasattr(layer, "sigma"):  # discriminator.Noise
            if layer.use_noise:
                x = rng.noise_add(x, layer.sigma)
            i += 1
        elif isinstance(layer, nn.Softmax):  # segmentation head (generator.py:75-76)
            if layer.dim != 1:
                raise NotImplementedError("only the channel softmax of the segmentation head is implemented")
            if ops_cl.is_cl(x):
                raise NotImplementedError("the segmentation head is not part of the bf16 channels-last path (no BASELINE config uses it there)")
            x = ops.softmax_channels(x)
            i += 1
        else:
            raise NotImplementedError(f"no HIP execution for layer {type(layer).__name__}")
    return x
