"""Device pick and weight init, same contract as the reference's util.py:16-28,186-195."""
from __future__ import annotations

import torch
import torch.nn as nn


def current_device() -> torch.device:
    """The process's current HIP device (cuda:0 unless torch.cuda.set_device was called — the
    data-parallel launcher sets it to LOCAL_RANK before any model is built), else cpu."""
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def init_weights(layer) -> None:
    """N(0, 0.02) for 2-D conv / transposed-conv weights, N(1, 0.02) / 0 for BatchNorm2d.
    The test is on the exact type, so Conv3d, BatchNorm3d and GRUCell keep PyTorch's
    default init — exactly what the reference does."""
    kind = type(layer)
    if kind is nn.Conv2d or kind is nn.ConvTranspose2d:
        nn.init.normal_(layer.weight.data, 0.0, 0.02)
        # `.data` writes bypass autograd's version counter, which the packed-weight caches are keyed on (ops._PackCache)
        torch.autograd.graph.increment_version(layer.weight)
    elif kind is nn.BatchNorm2d:
        nn.init.normal_(layer.weight.data, 1.0, 0.02)
        nn.init.constant_(layer.bias.data, 0.0)


def set_precision(module, mode=None):
    """Per-module switch of the MFMA product precision (the reference is fp32-only; this is the build's throughput option for BASELINE's bf16 / fp16
    configs): "bf16" = bf16 products with fp32 accumulation in this module's convolutions (forward, data and weight gradients), "fp32", or None = follow
    the process default (`dcvgan_amd.native.set_precision`).  Everything else of the module — tensors in HBM, BatchNorm statistics, optimiser — stays fp32."""
    import torch.nn as nn
    if mode not in (None, "fp32", "bf16", "f32x6"):
        raise ValueError(f"precision {mode!r}: expected None, 'fp32', 'bf16' or 'f32x6'")
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d)):
            if mode is None:
                if hasattr(m, "_dcv_precision"):
                    del m._dcv_precision
            else:
                m._dcv_precision = mode
    return module
