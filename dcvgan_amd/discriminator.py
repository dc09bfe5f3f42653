"""Discriminators with the reference's nn.Module surface, executed by HIP kernels.

Contract mirrored from /root/reference/src/discriminator.py:
``(ch1, ch2, use_noise=False, noise_sigma=0, ndf=64)`` constructors, raw-logit
outputs ``(B,4,4)`` / ``(B,4,4,4)`` / ``(B,3,4,4)`` (``.squeeze()`` included, so
B == 1 loses its batch dim as in the reference), state_dict keys
``conv_g.{1|0}``, ``conv_c.{1|0}``, ``main.{1,2,5,6,9[,10,13]}``.
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn

from . import layers, ops, ops_cl, util
from .rng import default_rng

_S3, _P3 = (1, 2, 2), (0, 1, 1)


_NO_STEM_GATE = os.environ.get("DCV_NO_STEM_GATE") is not None      # A/B: the stems run their own activation-derivative passes

class Noise(nn.Module):
    """x + sigma * N(0,1) when enabled — in train AND eval mode (discriminator.py:30-39)."""

    def __init__(self, use_noise: bool, sigma: float = 0.2):
        super().__init__()
        self.use_noise, self.sigma = use_noise, sigma
        self.device = util.current_device()

    def forward(self, x, rng=None):
        if not self.use_noise:
            return x
        return (rng if rng is not None else default_rng()).noise_add(x, self.sigma)


def _stage(noise, conv, bn_channels=None):
    seq = [noise, conv]
    if bn_channels is not None:
        bn = nn.BatchNorm2d if isinstance(conv, nn.Conv2d) else nn.BatchNorm3d
        seq += [bn(bn_channels), nn.LeakyReLU(0.2, inplace=True)]
    return seq


class _PairDiscriminator(nn.Module):
    """Shared shell: (geometry, colour) stems -> cat([colour, geometry]) -> trunk."""

    def __init__(self, ch1, ch2, use_noise, noise_sigma, ndf):
        super().__init__()
        self.ch1, self.ch2 = ch1, ch2
        self.use_noise, self.noise_sigma, self.ndf = use_noise, noise_sigma, ndf
        self._rng = None

    def _source(self):
        return self._rng if self._rng is not None else default_rng()

    def _describe(self, name):
        return json.dumps({name: {"ch_g": self.ch1, "ch_c": self.ch2, "ndf": self.ndf,
                                  "use_noise": self.use_noise, "noise_sigma": self.noise_sigma}})

    def forward(self, xg, xc):
        rng = self._source()
        # cat([hc, hg]) (colour first, discriminator.py:124,228): both stems write into one buffer
        conv = self.conv_g[-2]
        half = conv.out_channels
        sp = ops._out_shape(layers.geom_of(conv), xg)[2:]
        if ops_cl.active():   # bf16 channels-last data path inside the module; fp32 (strided views included) at its boundary
            cat = ops_cl.ConcatBuffer(xg.shape[0], half, half, sp, xg.device)
            hg = layers.run(self.conv_g, ops_cl.from_f32(xg), rng, out=cat.second)
            hc = layers.run(self.conv_c, ops_cl.from_f32(xc), rng, out=cat.first)
            return ops_cl.to_f32(layers.run(self.main, cat.join(hc, hg), rng)).squeeze()
        cat = ops.ConcatBuffer(xg.shape[0], half, half, sp, xg.device)
        hg = layers.run(self.conv_g, xg, rng, out=cat.second, act_slot=cat.slot)   # draw order: geometry stem first (discriminator.py:122-123)
        hc = layers.run(self.conv_c, xc, rng, out=cat.first, act_slot=cat.slot)
        # the trunk's first convolution applies the stems' LeakyReLU derivative in its data gradient's epilogue (read off the buffer; a Noise layer in between only adds)
        return layers.run(self.main, cat.join(hc, hg), rng, gate=None if _NO_STEM_GATE else (cat.buf, cat.slot)).squeeze()


class ImageDiscriminator(_PairDiscriminator):
    """Single frame pair -> (B, 4, 4) logits (discriminator.py:42-140)."""

    def __init__(self, ch1: int, ch2: int, use_noise: bool = False, noise_sigma: float = 0, ndf: int = 64):
        super().__init__(ch1, ch2, use_noise, noise_sigma, ndf)
        mk = lambda: Noise(use_noise, sigma=noise_sigma)
        c2 = lambda i, o: nn.Conv2d(i, o, 4, 2, 1, bias=False)
        self.conv_g = nn.Sequential(mk(), c2(ch1, ndf // 2), nn.LeakyReLU(0.2, inplace=True))
        self.conv_c = nn.Sequential(mk(), c2(ch2, ndf // 2), nn.LeakyReLU(0.2, inplace=True))
        self.main = nn.Sequential(*_stage(mk(), c2(ndf, ndf * 2), ndf * 2), *_stage(mk(), c2(ndf * 2, ndf * 4), ndf * 4),
                                  *_stage(mk(), c2(ndf * 4, 1)))
        self.device = util.current_device()

    def __str__(self, name: str = "idis") -> str:
        return self._describe(name)


def _c3(i, o):
    return nn.Conv3d(i, o, 4, stride=_S3, padding=_P3, bias=False)


class VideoDiscriminator(_PairDiscriminator):
    """Video pair -> (B, 4, 4, 4) logits; 4x4x4 convs, stride (1,2,2), no temporal padding;
    the two stems carry no Noise layer (discriminator.py:143-244)."""

    def __init__(self, ch1: int, ch2: int, use_noise: bool = False, noise_sigma: float = 0, ndf: int = 64):
        super().__init__(ch1, ch2, use_noise, noise_sigma, ndf)
        mk = lambda: Noise(use_noise, sigma=noise_sigma)
        self.conv_g = nn.Sequential(_c3(ch1, ndf // 2), nn.LeakyReLU(0.2, inplace=True))
        self.conv_c = nn.Sequential(_c3(ch2, ndf // 2), nn.LeakyReLU(0.2, inplace=True))
        self.main = nn.Sequential(*_stage(mk(), _c3(ndf, ndf * 2), ndf * 2), *_stage(mk(), _c3(ndf * 2, ndf * 4), ndf * 4),
                                  *_stage(mk(), _c3(ndf * 4, 1)))
        self.device = util.current_device()

    def __str__(self, name: str = "vdis") -> str:
        return self._describe(name)


class GradientDiscriminator(_PairDiscriminator):
    """Temporal difference of the geometry video -> (B, 3, 4, 4) logits; the colour
    input is ignored (discriminator.py:247-346)."""

    def __init__(self, ch1: int, ch2: int, use_noise: bool = False, noise_sigma: float = 0, ndf: int = 64):
        super().__init__(ch1, ch2, use_noise, noise_sigma, ndf)
        mk = lambda: Noise(use_noise, sigma=noise_sigma)
        self.main = nn.Sequential(*_stage(mk(), _c3(ch1, ndf), ndf), *_stage(mk(), _c3(ndf, ndf * 2), ndf * 2),
                                  *_stage(mk(), _c3(ndf * 2, ndf * 4), ndf * 4), *_stage(mk(), _c3(ndf * 4, 1)))
        self.device = util.current_device()

    def forward(self, xg, xc):
        if ops_cl.active():
            return ops_cl.to_f32(layers.run(self.main, ops_cl.from_f32(ops.temporal_diff(xg)), self._source())).squeeze()
        return layers.run(self.main, ops.temporal_diff(xg), self._source()).squeeze()

    def __str__(self, name: str = "vdis") -> str:  # the reference labels it "vdis" too (discriminator.py:335)
        return self._describe(name)
