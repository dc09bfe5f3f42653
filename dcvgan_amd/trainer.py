"""One G+D training iteration with the reference trainer's exact schedule.

Restates /root/reference/src/trainer.py:279-363 (the reference file itself cannot
ship: it imports evan / skvideo / colorlog / tensorboardX at module top, SURVEY
§0 D8) over objects with the reference's duck types — so it also drives the
reference's own classes, and a DCVGAN-style trainer drives ours.

Kept quirks: D-phase fakes are NOT detached (the dead generator backward runs),
``opt_ggen.step()`` is called twice, generators stay in whatever mode they were
left in until the first G phase, update gating by num_gen_update / num_dis_update.
Losses are returned as 0-d device tensors; ``sync_losses=True`` reproduces the
reference's four ``.cpu().item()`` host syncs per iteration.
"""
from __future__ import annotations

import collections
import os
from typing import Dict, Optional

import torch

from . import discriminator as D
from . import generator as G
from . import loss as Lm
from . import ops, optim, util
from .configs import StepConfig

MODEL_NAMES = ("ggen", "cgen", "idis", "vdis", "gdis")


def build_models(cfg: StepConfig, device=None) -> Dict[str, torch.nn.Module]:
    """train.py:117-165: positional constructor wiring + init_weights."""
    w = cfg.width
    ggen = G.GeometricVideoGenerator(cfg.dim_z_content, cfg.dim_z_motion, cfg.channel, cfg.geometric_info, w["ggen"], cfg.video_length)
    cgen = G.ColorVideoGenerator(ggen.channel, cfg.dim_z_color, cfg.geometric_info, w["cgen"], cfg.video_length)
    idis = D.ImageDiscriminator(ggen.channel, cgen.channel, cfg.use_noise["idis"], cfg.noise_sigma["idis"], w["idis"])
    vdis = D.VideoDiscriminator(ggen.channel, cgen.channel, cfg.use_noise["vdis"], cfg.noise_sigma["vdis"], w["vdis"])
    gdis = D.GradientDiscriminator(ggen.channel, cgen.channel, cfg.use_noise["gdis"], cfg.noise_sigma["gdis"], w["gdis"])
    models = dict(ggen=ggen, cgen=cgen, idis=idis, vdis=vdis, gdis=gdis)
    for m in models.values():
        m.apply(util.init_weights)
        m.to(device if device is not None else util.current_device())
    return models


def build_loss(cfg: StepConfig):
    return Lm.AdversarialLoss() if cfg.loss == "adversarial-loss" else Lm.HingeLoss()


def build_optimizers(cfg: StepConfig, models, data_parallel: bool = False, overlap: Optional[bool] = None):
    """train.py:169-176.  Data parallel: the optimisers stepped after the same backward share one gradient
    bucket — D phase (trainer.py:319-322) and G phase (trainer.py:356-359) — so an iteration has two collectives.
    `overlap` (default: the environment's DCV_DP_OVERLAP, else off): per-model chunks whose collectives start from the hook of the chunk's last gradient, on a
    communication stream, while the rest of the backward runs (optim.GradBucket(overlap=True)); off by default until an N > 1 run has measured it."""
    opts = {}
    if overlap is None:
        overlap = os.environ.get("DCV_DP_OVERLAP") is not None
    buckets = {"D": optim.GradBucket(overlap=overlap), "G": optim.GradBucket(overlap=overlap)} if data_parallel else None
    for name in MODEL_NAMES:
        o = optim.Adam(models[name].parameters(), lr=cfg.lr[name], betas=(0.5, 0.999), weight_decay=cfg.decay[name])
        opts[name] = optim.DataParallelAdam(o, buckets["D" if name.endswith("dis") else "G"]) if data_parallel else o
    return opts


class StepRunner:
    """`elide_dead_backward=True` builds the D-phase fakes without a tape (they are detached): the
    reference backpropagates `loss_dis` through cgen/ggen too (trainer.py:304-319, fakes not detached)
    and then throws those gradients away with `zero_grad()` at :340-341, so parameters, buffers and
    losses are identical either way; only ~23 % of the step's FLOPs disappear (BASELINE.md §4,
    "minimal" column).  Default False = the reference's as-written schedule."""

    def __init__(self, cfg: StepConfig, models, optimizers, loss, sync_losses: bool = False, elide_dead_backward: bool = False,
                 side_streams: Optional[bool] = None):
        self.cfg, self.models, self.opt, self.loss = cfg, models, optimizers, loss
        self.iteration = 0
        self.sync_losses = sync_losses
        self.elide_dead_backward = elide_dead_backward
        if side_streams is None:
            side_streams = os.environ.get("DCV_NO_SIDE_STREAMS") is None
        self._lanes = None
        dev = next(models["idis"].parameters()).device
        if side_streams and dev.type == "cuda":
            self._lanes = [torch.cuda.Stream(dev) for _ in range(3)]
        if cfg.start_in_eval:  # trainer.py:266-267: log_samples/evaluate leave the generators in eval()
            models["ggen"].eval(); models["cgen"].eval()
        # diagnostics (tools/phases.py): HIP events on the main stream at the phase boundaries of an iteration; None = off (no event is ever recorded)
        self.phase_marks = None
        self._ones = {}
        # How many iterations the host may enqueue ahead of the GPU.  The reference's loop reads the losses on the host every iteration (trainer.py:326-328,363) and is
        # never ahead; a caller that does not (bench.py, sync_losses=False) enqueues an iteration in about a third of the time the GPU takes to run it, and every tensor
        # that crossed streams is only returned to the allocator when the GPU reaches its last use — without a bound the memory in flight grows with the lead (soak at
        # B = 70: 37 GB after 5 iterations, 146 GB allocated / 191 GB reserved after 150).  Two iterations keep the GPU's queues full and the memory flat.
        self.max_ahead = max(1, int(os.environ.get("DCV_MAX_ITERATIONS_AHEAD", "2")))
        self._inflight = collections.deque()

    def _mark(self, name):
        if self.phase_marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            self.phase_marks.append((name, e))

    # The three discriminators are independent of one another (trainer.py:299-309, 347-349): each runs on its own HIP stream, so
    # their small layers (70-frame image discriminator, the 1-channel heads, BN reductions) and the tail rounds of the large ones
    # fill one another's idle CUs.  Autograd replays every tape entry on the stream of its forward and orders producer/consumer
    # streams itself, and joins the leaf streams with the caller's at the end of backward(); the host order of all calls — hence
    # every random draw and every result bit — is that of the single-stream schedule.
    def _on_lanes(self, dis, call, join: bool = True):
        if self._lanes is None:
            return tuple(call(d) for d in dis)
        main = torch.cuda.current_stream()
        outs = []
        for lane, d in zip(self._lanes, dis):
            lane.wait_stream(main)          # inputs (and the optimiser's updates) were produced on the main stream
            with torch.cuda.stream(lane):
                outs.append(call(d))
        if join:
            self._adopt(outs)
        return tuple(outs)

    def _adopt(self, outs):
        if self._lanes is None:
            return
        main = torch.cuda.current_stream()
        for lane, y in zip(self._lanes, outs):
            main.wait_stream(lane)
            y.record_stream(main)           # allocated on the lane, read by the loss kernels on the main stream

    # The fake clips are read by three discriminators — the image discriminator takes frame t_rand — and the geometry clip by the colour generator too
    # (trainer.py:303-309, 344-349).  ops.fan_out hands every consumer a view and forms the clip's ONE gradient with the library's kernels; left to autograd the fan-in
    # is a torch add per extra consumer plus the zero-fill + copy of each slice's backward (18 + ~25 torch launches per iteration in round 5's traces).
    def _colour(self, cgen, xg_fake, t_rand):
        """-> ((frame, for vdis, for gdis) of the geometry clip, the same of the colour clip cgen makes of it)"""
        if not (xg_fake.is_cuda and xg_fake.requires_grad):
            xc = cgen.forward_videos(xg_fake)
            return (xg_fake[:, :, t_rand], xg_fake, xg_fake), (xc[:, :, t_rand], xc, xc)
        g_i, g_c, g_v, g_g = ops.fan_out(xg_fake, t_rand, 3)
        c_i, c_v, c_g = ops.fan_out(cgen.forward_videos(g_c), t_rand, 2)
        return (g_i, g_v, g_g), (c_i, c_v, c_g)

    def _fakes_through(self, dis, idis, xg, xc, t_rand):
        which = {id(d): k for k, d in enumerate(dis)}
        return self._on_lanes(dis, lambda d: d(xg[which[id(d)]], xc[which[id(d)]]))

    def _root(self, loss):
        """d loss / d loss = 1 as a cached device scalar (backward() without an argument fills a fresh ones_like with a torch kernel)."""
        one = self._ones.get(loss.device)
        if one is None:
            one = self._ones[loss.device] = torch.ones((), dtype=loss.dtype, device=loss.device)
        return one

    def step(self, xc_real: torch.Tensor, xg_real: torch.Tensor, t_rand: int):
        c, m, o = self.cfg, self.models, self.opt
        ggen, cgen, idis, vdis, gdis = (m[k] for k in MODEL_NAMES)
        self.iteration += 1
        if xc_real.is_cuda:
            while len(self._inflight) >= self.max_ahead:
                self._inflight.popleft().synchronize()      # the host waits for iteration i - max_ahead to END: no kernel of the current queue is delayed
        # ---- discriminator phase (trainer.py:285-328) ----
        for d in (idis, vdis, gdis):
            d.train()
        for d in (idis, vdis, gdis):
            d.zero_grad()
        dis = (idis, vdis, gdis)
        self._mark("start")
        y_real = self._on_lanes(dis, lambda d: d(xg_real[:, :, t_rand], xc_real[:, :, t_rand]) if d is idis else d(xg_real, xc_real), join=False)
        with torch.set_grad_enabled(not self.elide_dead_backward):
            xg_fake = ggen.sample_videos(c.batchsize)     # on the main stream, beside the discriminators' real-batch passes
            xg_fake, xc_fake = self._colour(cgen, xg_fake, t_rand)
        self._mark("D: generators forward (beside D on the real batch)")
        y_fake = self._fakes_through(dis, idis, xg_fake, xc_fake, t_rand)
        self._adopt(y_real)
        self._mark("D: discriminators forward on the fakes")
        loss_idis = self.loss.compute_dis_loss(y_real[0], y_fake[0])
        loss_vdis = self.loss.compute_dis_loss(y_real[1], y_fake[1])
        loss_gdis = self.loss.compute_dis_loss(y_real[2], y_fake[2])
        loss_dis = ops.sum_scalars(loss_idis, loss_vdis, loss_gdis) if loss_idis.is_cuda else loss_idis + loss_vdis + loss_gdis      # trainer.py:315
        if self.iteration % c.num_gen_update == 0:
            loss_dis.backward(self._root(loss_dis))
            self._mark("D: backward (D lanes, then the generators' dead backward)")
            o["idis"].step(); o["vdis"].step(); o["gdis"].step()
            self._mark("D: Adam")
        else:
            loss_dis.detach_()
        if self.sync_losses:   # trainer.py:326-328 — on the loss objects themselves, as the reference reads them (they carry a host mirror: loss.HostMirroredLoss)
            out = {"loss_idis": loss_idis.cpu().item(), "loss_vdis": loss_vdis.cpu().item(), "loss_gdis": loss_gdis.cpu().item()}
        else:
            out = {"loss_idis": loss_idis.detach(), "loss_vdis": loss_vdis.detach(), "loss_gdis": loss_gdis.detach()}
        del y_real, y_fake, xg_fake, xc_fake, loss_dis
        # ---- generator phase (trainer.py:338-363) ----
        ggen.train(); cgen.train()
        ggen.zero_grad(); cgen.zero_grad()
        xg_fake = ggen.sample_videos(c.batchsize)
        xg_fake, xc_fake = self._colour(cgen, xg_fake, t_rand)
        self._mark("G: generators forward")
        y_fake = self._fakes_through(dis, idis, xg_fake, xc_fake, t_rand)
        loss_gen = self.loss.compute_gen_loss(*y_fake)
        self._mark("G: discriminators forward on the fakes")
        if self.iteration % c.num_dis_update == 0:
            loss_gen.backward(self._root(loss_gen))
            self._mark("G: backward (D lanes, then the generators)")
            o["ggen"].step(); o["cgen"].step(); o["ggen"].step()  # ggen twice — trainer.py:357-359
            self._mark("G: Adam")
        else:
            loss_gen.detach_()
        out["loss_gen"] = loss_gen.cpu().item() if self.sync_losses else loss_gen.detach()     # trainer.py:363
        if xc_real.is_cuda:
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream())      # (the lanes and the companion stream have joined the main stream by now)
            self._inflight.append(e)
        return out
