"""Generators with the reference's nn.Module surface, executed by HIP kernels.

Contract mirrored from /root/reference/src/generator.py: class
names, positional constructor arguments, attributes, method names/returns and
``state_dict`` keys (``recurrent.*``, ``main.{0..12}``, ``inconv.main.0``,
``down_blocks.i.main.{0,1}``, ``up_blocks.i.main.{0,1}``, ``outconv.main.0``).
Execution is delegated to ``layers.run`` / ``ops`` — there is no torch.nn.functional
path and no CPU fallback.
"""
from __future__ import annotations

import os

import json

import torch
import torch.nn as nn

from . import layers, ops, ops_cl, util
from .rng import default_rng


def _convT(cin, cout, k, s, p):
    return nn.ConvTranspose2d(cin, cout, k, s, p, bias=False)


_NORMAL_MANY = os.environ.get("DCV_NO_NORMAL_MANY") is None      # A/B only: one launch for the per-frame motion noise

class GeometricVideoGenerator(nn.Module):
    """Noise -> geometry video (depth / flow / segmentation), generator.py:11-155.

    ``sample_videos(B)`` returns a (B, C, T, 64, 64) *view* of a (B, T, C, 64, 64)
    buffer, exactly like the reference (non-contiguous when C > 1)."""

    def __init__(self, dim_z_content: int, dim_z_motion: int, channel: int, geometric_info: str,
                 ngf: int = 64, video_length: int = 16):
        super().__init__()
        self.dim_z_content, self.dim_z_motion = dim_z_content, dim_z_motion
        self.channel, self.geometric_info = channel, geometric_info
        self.video_length, self.ngf = video_length, ngf
        self.dim_z = dim_z_motion + dim_z_content
        self.recurrent = nn.GRUCell(dim_z_motion, dim_z_motion)
        widths = [self.dim_z, ngf * 8, ngf * 4, ngf * 2, ngf]
        stack = []
        for i in range(4):  # 1x1 -> 4 -> 8 -> 16 -> 32
            stack += [_convT(widths[i], widths[i + 1], 4, 1 if i == 0 else 2, 0 if i == 0 else 1),
                      nn.BatchNorm2d(widths[i + 1]), nn.ReLU(inplace=True)]
        stack.append(_convT(ngf, channel, 4, 2, 1))  # -> 64
        stack.append(nn.Softmax(dim=1) if geometric_info == "segmentation" else nn.Tanh())
        self.main = nn.Sequential(*stack)
        self.device = util.current_device()
        self._rng = None  # None -> process-wide Philox stream; tests inject a replay source

    # -- random inputs (draw order: z_content, h0, e_1..e_T — generator.py:110-116) --
    def _source(self):
        return self._rng if self._rng is not None else default_rng()

    def get_gru_initial_state(self, batchsize: int) -> torch.Tensor:
        return self._source().normal((batchsize, self.dim_z_motion), self.device)

    def get_iteration_noise(self, batchsize: int) -> torch.Tensor:
        return self._source().normal((batchsize, self.dim_z_motion), self.device)

    def sample_z_m(self, batchsize: int) -> torch.Tensor:
        h0 = self.get_gru_initial_state(batchsize)
        # (the per-frame draws of get_iteration_noise, generator.py:57-62 of the reference, in one launch: same values, same stream position)
        src = self._source()
        if hasattr(src, "normal_many") and _NORMAL_MANY:
            e = src.normal_many(self.video_length, (batchsize, self.dim_z_motion), self.device)
        else:
            e = torch.stack([self.get_iteration_noise(batchsize) for _ in range(self.video_length)], 0)
        r = self.recurrent
        hs = ops.gru_sequence(e, h0, r.weight_ih, r.weight_hh, r.bias_ih, r.bias_hh)  # (B, T, dm)
        return hs.view(batchsize * self.video_length, self.dim_z_motion)

    def sample_z_content(self, batchsize: int) -> torch.Tensor:
        zc = self._source().normal((batchsize, self.dim_z_content), self.device)
        return ops.tile_rows(zc, self.video_length)      # every clip's content latent for each of its frames

    def sample_z_video(self, batchsize: int) -> torch.Tensor:
        zc = self.sample_z_content(batchsize)
        zm = self.sample_z_m(batchsize)
        return ops.cat_channels(zc, zm)

    def sample_videos(self, batchsize: int) -> torch.Tensor:
        z = self.sample_z_video(batchsize)
        if ops_cl.active():   # bf16 channels-last data path inside the module; fp32 at its boundary (latents in, frames out)
            frames = ops_cl.to_f32(layers.run(self.main, ops_cl.from_f32(z.view(-1, self.dim_z, 1, 1)), self._source()))
        else:
            frames = layers.run(self.main, z.view(-1, self.dim_z, 1, 1), self._source())
        to_video = lambda f: f.view(batchsize, self.video_length, self.channel, 64, 64).permute(0, 2, 1, 3, 4)
        return ops_cl.carry_twin(to_video(frames), frames, to_video) if ops_cl.active() else to_video(frames)

    def __str__(self, name: str = "ggen") -> str:
        return json.dumps({name: {"dim_zc": self.dim_z_content, "dim_zm": self.dim_z_motion, "channel": self.channel,
                                  "geometric_info": self.geometric_info, "vlen": self.video_length, "ngf": self.ngf}})


class Inconv(nn.Module):
    """3x3 conv + LeakyReLU with torch's DEFAULT slope 0.01 (generator.py:173-176)."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        self.main = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, 1, 1, bias=False), nn.LeakyReLU(inplace=True))


class DownBlock(nn.Module):
    """4x4 stride-2 conv, BN, LeakyReLU(0.2) (generator.py:203-213)."""

    def __init__(self, in_ch: int, out_ch: int, dropout: bool = False):
        super().__init__()
        seq = [nn.Conv2d(in_ch, out_ch, 4, 2, 1, bias=False), nn.BatchNorm2d(out_ch)]
        if dropout:
            seq.append(nn.Dropout2d(0.5, inplace=True))
        seq.append(nn.LeakyReLU(0.2, inplace=True))
        self.main = nn.Sequential(*seq)


class UpBlock(nn.Module):
    """4x4 stride-2 transposed conv, BN, [Dropout2d between BN and ReLU], ReLU (generator.py:238-250)."""

    def __init__(self, in_ch: int, out_ch: int, dropout: bool = False):
        super().__init__()
        seq = [_convT(in_ch, out_ch, 4, 2, 1), nn.BatchNorm2d(out_ch)]
        if dropout:
            seq.append(nn.Dropout2d(0.5, inplace=True))
        seq.append(nn.ReLU(inplace=True))
        self.main = nn.Sequential(*seq)


class Outconv(nn.Module):
    """3x3 transposed conv + Tanh (generator.py:272-277)."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        self.main = nn.Sequential(_convT(in_ch, out_ch, 3, 1, 1), nn.Tanh())


def _block_forward(self, x, rng=None, out=None, grad_slot=None, act_slot=None, bn_link=None):
    return layers.run(self.main, x, rng if rng is not None else default_rng(), out=out, grad_slot=grad_slot, act_slot=act_slot, bn_link=bn_link)


for _cls in (Inconv, DownBlock, UpBlock, Outconv):
    _cls.forward = _block_forward


class ColorVideoGenerator(nn.Module):
    """Geometry frames -> RGB frames, a 1+6 down / 6 up +1 U-Net (generator.py:285-448)."""

    _DOWN = ((1, 1), (1, 2), (2, 4), (4, 4), (4, 4), (4, 4))

    def __init__(self, in_ch: int, dim_z: int, geometric_info: str, ngf: int = 64, video_length: int = 16):
        super().__init__()
        self.in_ch, self.out_ch, self.dim_z, self.geometric_info = in_ch, 3, dim_z, geometric_info
        self.inconv = Inconv(in_ch, ngf)
        self.down_blocks = nn.ModuleList([DownBlock(ngf * a, ngf * b) for a, b in self._DOWN])
        ups = ((ngf * 4 + dim_z, ngf * 4, True), (ngf * 8, ngf * 4, True), (ngf * 8, ngf * 4, False),
               (ngf * 8, ngf * 2, False), (ngf * 4, ngf, False), (ngf * 2, ngf, False))
        self.up_blocks = nn.ModuleList([UpBlock(a, b, dropout=d) for a, b, d in ups])
        self.outconv = Outconv(ngf * 2, self.out_ch)
        self.n_down_blocks, self.n_up_blocks = len(self.down_blocks), len(self.up_blocks)
        self.device = util.current_device()
        self.channel, self.video_length = 3, video_length
        self._rng = None

    def _source(self):
        return self._rng if self._rng is not None else default_rng()

    def make_hidden(self, batchsize: int) -> torch.Tensor:
        return self._source().normal((batchsize, self.dim_z), self.device).view(batchsize, self.dim_z, 1, 1)

    def forward(self, x: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        rng = self._source()
        if self.geometric_info == "segmentation":  # one-hot -> {-1, +1} maps (generator.py:378-385); SURVEY §8(f).4
            x = ops.segm_onehot(x)
        if ops_cl.active():
            return self._forward_cl(x, z, rng)
        # Every torch.cat of the reference (generator.py:393-400) joins an up-path tensor with a skip:
        # both producers write straight into the two channel slices of one buffer, so no copy is made.
        nb = x.shape[0]
        widths = [self.inconv.main[0].out_channels] + [b.main[0].out_channels for b in self.down_blocks]   # skip channels
        ups = [b.main[0].out_channels for b in self.up_blocks]
        size = [x.shape[2] >> k for k in range(7)]                       # 64, 32, ..., 1
        # bufs[k] (k = 0..5): cat([up-path tensor at resolution size[k], skips[k]]); bufs[6]: cat([skips[6], z])
        bufs = [ops.ConcatBuffer(nb, ups[5 - k], widths[k], (size[k], size[k]), x.device) for k in range(6)]
        bufs.append(ops.ConcatBuffer(nb, widths[6], self.dim_z, (size[6], size[6]), x.device))
        skips = [self.inconv(x, rng, out=bufs[0].second, act_slot=bufs[0].slot)]
        for k, blk in enumerate(self.down_blocks):
            dst = bufs[k + 1].second if k + 1 < 6 else bufs[6].first
            skips.append(blk(skips[-1], rng, out=dst, grad_slot=bufs[k].slot))   # skips[k] lives in bufs[k].second
        zc = ops.copy_into(z, bufs[6].second)
        h = bufs[6].join(skips[6], zc)
        # the last stage — UpBlock 5's BatchNorm -> cat with the stem's skip -> Outconv — shares an ops.BnLink: in the backward the head's data gradient and that
        # BatchNorm's backward run as one fused pair of launches (dcv_conv_backward_data_bn)
        # ... and in the forward that BatchNorm's output is not written at all: the head reads the BatchNorm input and normalises on load (the slice feeds nothing else;
        # not while a test's activation-pattern tap is installed, which reads it)
        link = ops.BnLink(defer=layers.KINK_TAP is None)
        for i, blk in enumerate(self.up_blocks):
            h = blk(h, rng, out=bufs[5 - i].first, bn_link=link if i == 5 else None)
            h = bufs[5 - i].join(h, skips[5 - i])
        return self.outconv(h, rng, bn_link=link)

    def _forward_cl(self, x, z, rng):
        """The same U-Net on the bf16 channels-last data path: x, z fp32 in, RGB frames fp32 out; every concatenation is two channel ranges of
        one channels-last buffer (whose pixel pitch is rounded up to whole 32-channel K blocks: the 256 + dim_z bottleneck gets zero padding)."""
        nb = x.shape[0]
        widths = [self.inconv.main[0].out_channels] + [b.main[0].out_channels for b in self.down_blocks]
        ups = [b.main[0].out_channels for b in self.up_blocks]
        size = [x.shape[2] >> k for k in range(7)]
        bufs = [ops_cl.ConcatBuffer(nb, ups[5 - k], widths[k], (size[k], size[k]), x.device) for k in range(6)]
        bufs.append(ops_cl.ConcatBuffer(nb, widths[6], self.dim_z, (size[6], size[6]), x.device))
        skips = [self.inconv(ops_cl.from_f32(x), rng, out=bufs[0].second, act_slot=bufs[0].slot)]
        for k, blk in enumerate(self.down_blocks):
            dst = bufs[k + 1].second if k + 1 < 6 else bufs[6].first
            skips.append(blk(skips[-1], rng, out=dst, grad_slot=bufs[k].slot))      # skips[k] lives in bufs[k].second
        zc = ops_cl.from_f32(z, out=bufs[6].second)
        h = bufs[6].join(skips[6], zc)
        for i, blk in enumerate(self.up_blocks):
            h = blk(h, rng, out=bufs[5 - i].first)
            h = bufs[5 - i].join(h, skips[5 - i])
        return ops_cl.to_f32(self.outconv(h, rng))

    def forward_videos(self, xs: torch.Tensor) -> torch.Tensor:
        B, Cg, T, H, W = xs.shape
        z = self.make_hidden(B)
        zs = ops.tile_rows(z.view(B, self.dim_z), T).view(B * T, self.dim_z, 1, 1)
        to_frames = lambda v: v.permute(0, 2, 1, 3, 4).reshape(B * T, Cg, H, W)  # a view for generator outputs
        frames = to_frames(xs)
        if ops_cl.active() and frames.data_ptr() == xs.data_ptr():      # (a copy has no twin: its bits are its own)
            ops_cl.carry_twin(frames, xs, to_frames)
        ys = self(frames, zs)
        to_video = lambda f: f.view(B, T, 3, H, W).permute(0, 2, 1, 3, 4)
        return ops_cl.carry_twin(to_video(ys), ys, to_video) if ops_cl.active() else to_video(ys)

    def __str__(self, name: str = "cgen") -> str:
        return json.dumps({name: {"in_ch": self.in_ch, "out_ch": self.out_ch, "dim_z": self.dim_z,
                                  "n_down_blocks": self.n_down_blocks, "n_up_blocks": self.n_up_blocks}})
