"""CPU oracle for the DCVGAN G+D training step.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``dcvgan_amd/`` imports or calls it.

It is a functional (state-dict + ``torch.nn.functional``) restatement, on the
CPU and in fp32, of the reference's hot path:

  * generators        /root/reference/src/generator.py:11-155, :158-448
  * discriminators    /root/reference/src/discriminator.py:11-346
  * losses            /root/reference/src/loss.py:64-193
  * weight init       /root/reference/src/util.py:186-195
  * one trainer step  /root/reference/src/trainer.py:279-363
  * Adam wiring       /root/reference/src/train.py:171-176

The arithmetic itself lives in PyTorch (the reference pins torch==1.2.0,
requirements.txt:15; the oracle runs on the torch in this image).  Parity is
PINNED: ``tests/golden/make_golden.py`` imports the real reference classes in
the build container and stores their inputs/outputs/gradients as fixtures;
``tests/test_oracle_golden.py`` checks this restatement against them.

Random draws: the reference pulls every latent / Noise / Dropout2d sample from
the global torch generator in program order.  ``TorchRng`` below draws in the
same order from the same generator (bit-identical on CPU for an equal seed) and
records every draw, so that a GPU run can *replay* the very same tensors
(``ReplayRng``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- #
# random sources
# --------------------------------------------------------------------------- #
class TorchRng:
    """Draws from the global CPU generator in the reference's call order and
    keeps a log ``[(kind, tensor), ...]`` of everything drawn."""

    def __init__(self, record: bool = True):
        self.record = record
        self.log: List[Tuple[str, torch.Tensor]] = []

    def normal(self, shape: Sequence[int]) -> torch.Tensor:
        # generator.py:85,88,104,356  discriminator.py:33-37 -> empty().normal_()
        t = torch.empty(tuple(shape)).normal_()
        if self.record:
            self.log.append(("normal", t.clone()))
        return t

    def dropout2d_mask(self, n: int, c: int, p: float) -> torch.Tensor:
        # nn.Dropout2d draws one Bernoulli(1-p) per (n, c) plane and rescales by
        # 1/(1-p) (generator.py:211,248).  Drawing it on a ones tensor of shape
        # (n, c, 1, 1) consumes the generator identically.
        m = F.dropout2d(torch.ones(n, c, 1, 1), p, True)
        if self.record:
            self.log.append(("dropout2d", m.clone()))
        return m


class ReplayRng:
    """Hands back a recorded draw log, in order (shape-checked)."""

    def __init__(self, log: List[Tuple[str, torch.Tensor]], device="cpu"):
        self.log = log
        self.pos = 0
        self.device = device

    def _next(self, kind: str, shape) -> torch.Tensor:
        k, t = self.log[self.pos]
        self.pos += 1
        assert k == kind, f"replay kind mismatch at {self.pos - 1}: {k} vs {kind}"
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.to(self.device)

    def normal(self, shape):
        return self._next("normal", tuple(shape))

    def dropout2d_mask(self, n, c, p):
        return self._next("dropout2d", (n, c, 1, 1))


# --------------------------------------------------------------------------- #
# tiny helpers
# --------------------------------------------------------------------------- #
class KinkTape:
    """Test instrumentation for the (Leaky)ReLU kinks.  Deep gradients of this network are discontinuous in the
    pre-activations: one that is within rounding of zero picks the other branch and every upstream gradient moves
    by ~1e-3.  To compare gradient ARITHMETIC without that lottery, an evaluation can be told which branch each
    element took elsewhere: with a tape installed (`with KinkTape(masks) as tape:`) the i-th (Leaky)ReLU call
    computes its forward value from its own input as usual, but differentiates with the i-th recorded pattern
    (True = identity branch).  `tape.mismatch` collects, per call, the number of elements whose own sign disagrees
    with the pattern and how far from zero the furthest of them is, relative to the call's rms input."""

    active = None

    def __init__(self, masks=None):
        """`masks=None`: RECORD this evaluation's own patterns into `self.recorded` instead of replaying."""
        self.masks, self.pos, self.mismatch = masks, 0, []
        self.recorded = [] if masks is None else None
        self.last_bn = None      # (state-dict prefix, per-channel |mean| / std of that BatchNorm's input): names the furthest mismatch

    def __enter__(self):
        KinkTape.active = self
        return self

    def __exit__(self, *a):
        KinkTape.active = None

    def next(self, x):
        if self.recorded is not None:
            self.recorded.append(x.detach() > 0)
            self.pos += 1
            return self.recorded[-1]
        m = self.masks[self.pos]
        self.pos += 1
        assert tuple(m.shape) == tuple(x.shape), (self.pos - 1, tuple(m.shape), tuple(x.shape))
        own = x.detach() > 0
        bad = own != m
        n = int(bad.sum())
        far = float(x.detach()[bad].abs().max() / x.detach().pow(2).mean().sqrt()) if n else 0.0
        where = None
        if n:   # which layer / channel the furthest mismatch sits in, and how ill-conditioned that BatchNorm channel is
            xa = torch.where(bad, x.detach().abs(), torch.zeros((), dtype=x.dtype))
            ch = int(xa.flatten(2).amax(dim=(0, 2)).argmax()) if x.dim() >= 3 else -1
            bn = self.last_bn
            if bn is not None and ch >= 0 and bn[1].numel() == x.shape[1]:
                where = (bn[0], ch, float(bn[1][ch]))
            else:
                where = ("(no BatchNorm in front)", ch, 0.0)
        self.last_bn = None
        self.mismatch.append((n, x.numel(), far, tuple(x.shape), where))
        return m


class _PatternLeaky(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, slope):
        ctx.save_for_backward(mask)
        ctx.slope = slope
        return torch.where(x > 0, x, x * slope)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, g, g * ctx.slope), None, None


def _lrelu(x: torch.Tensor, slope: float) -> torch.Tensor:
    """nn.ReLU (slope 0) / nn.LeakyReLU(slope)."""
    if KinkTape.active is not None:
        return _PatternLeaky.apply(x, KinkTape.active.next(x), slope)
    return F.relu(x) if slope == 0.0 else F.leaky_relu(x, slope)


def _bn(st: State, prefix: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """nn.BatchNorm{2,3}d forward incl. running-stat side effects."""
    rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    if KinkTape.active is not None and KinkTape.active.masks is not None:
        with torch.no_grad():
            xf = x.detach().transpose(0, 1).flatten(1).double()
            KinkTape.active.last_bn = (prefix, xf.mean(1).abs() / xf.std(1).clamp_min(1e-300))
    if training:
        st[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, st[prefix + ".weight"], st[prefix + ".bias"],
                        training, BN_MOMENTUM, BN_EPS)


def _noise(x: torch.Tensor, use: bool, sigma: float, rng) -> torch.Tensor:
    # discriminator.py:30-39 — active in train AND eval
    if not use:
        return x
    return x + sigma * rng.normal(x.shape)


# --------------------------------------------------------------------------- #
# geometric generator   (generator.py:11-155)
# --------------------------------------------------------------------------- #
def ggen_latent(st: State, B: int, T: int, dzc: int, dzm: int, rng) -> torch.Tensor:
    """sample_z_video (generator.py:90-116).  Draw order: z_c, h0, e_1..e_T."""
    zc = rng.normal((B, dzc))
    zc = zc.repeat(1, T).view(B * T, dzc)
    h = rng.normal((B, dzm))
    hs = []
    for _ in range(T):
        e = rng.normal((B, dzm))
        h = gru_cell(e, h, st["recurrent.weight_ih"], st["recurrent.weight_hh"],
                     st["recurrent.bias_ih"], st["recurrent.bias_hh"])
        hs.append(h)
    zm = torch.stack(hs, 1).view(B * T, dzm)
    return torch.cat([zc, zm], 1)


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """nn.GRUCell arithmetic (generator.py:58,94)."""
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return n + z * (h - n)


def ggen_decode(st: State, z: torch.Tensor, training: bool, segmentation=False) -> torch.Tensor:
    """self.main (generator.py:60-80): 5x ConvT2d with BN+ReLU between, Tanh head."""
    h = z.view(z.shape[0], z.shape[1], 1, 1)
    h = F.conv_transpose2d(h, st["main.0.weight"], None, 1, 0)
    h = _lrelu(_bn(st, "main.1", h, training), 0.0)
    for conv, bn in ((3, 4), (6, 7), (9, 10)):
        h = F.conv_transpose2d(h, st[f"main.{conv}.weight"], None, 2, 1)
        h = _lrelu(_bn(st, f"main.{bn}", h, training), 0.0)
    h = F.conv_transpose2d(h, st["main.12.weight"], None, 2, 1)
    return torch.softmax(h, 1) if segmentation else torch.tanh(h)


def ggen_sample_videos(st: State, B: int, T: int, dzc: int, dzm: int, channel: int,
                       rng, training: bool, segmentation=False) -> torch.Tensor:
    """sample_videos (generator.py:118-141) -> (B, C, T, 64, 64) permuted view."""
    z = ggen_latent(st, B, T, dzc, dzm, rng)
    h = ggen_decode(st, z, training, segmentation)
    return h.view(B, T, channel, 64, 64).permute(0, 2, 1, 3, 4)


# --------------------------------------------------------------------------- #
# colour generator   (generator.py:158-448)
# --------------------------------------------------------------------------- #
def cgen_forward(st: State, x: torch.Tensor, z: torch.Tensor, rng, training: bool,
                 segmentation=False) -> torch.Tensor:
    """ColorVideoGenerator.forward (generator.py:361-402)."""
    if segmentation:  # generator.py:378-385
        idx = torch.argmax(x, 1, keepdim=True)
        x = torch.full_like(x, -1.0).scatter_(1, idx, 1.0)
    # Inconv: conv3x3 + LeakyReLU(default slope 0.01)  (generator.py:173-176)
    hs = [_lrelu(F.conv2d(x, st["inconv.main.0.weight"], None, 1, 1), 0.01)]
    for i in range(6):  # DownBlock (generator.py:203-207)
        h = F.conv2d(hs[-1], st[f"down_blocks.{i}.main.0.weight"], None, 2, 1)
        h = _bn(st, f"down_blocks.{i}.main.1", h, training)
        hs.append(_lrelu(h, 0.2))
    h = torch.cat([hs[-1], z], 1)  # generator.py:393
    for i in range(6):  # UpBlock (generator.py:238-248)
        if i > 0:
            h = torch.cat([h, hs[-i - 1]], 1)
        h = F.conv_transpose2d(h, st[f"up_blocks.{i}.main.0.weight"], None, 2, 1)
        h = _bn(st, f"up_blocks.{i}.main.1", h, training)
        if i < 2 and training:  # Dropout2d(0.5) between BN and ReLU
            h = h * rng.dropout2d_mask(h.shape[0], h.shape[1], 0.5)
        h = _lrelu(h, 0.0)
    h = torch.cat([h, hs[0]], 1)
    h = F.conv_transpose2d(h, st["outconv.main.0.weight"], None, 1, 1)
    return torch.tanh(h)


def cgen_forward_videos(st: State, xs: torch.Tensor, dim_z: int, rng, training: bool,
                        segmentation=False) -> torch.Tensor:
    """forward_videos (generator.py:404-435)."""
    B, C, T, H, W = xs.shape
    z = rng.normal((B, dim_z)).view(B, dim_z, 1, 1)  # make_hidden :355-359
    zs = z.unsqueeze(1).repeat(1, T, 1, 1, 1).view(B * T, dim_z, 1, 1)
    x = xs.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    y = cgen_forward(st, x, zs, rng, training, segmentation)
    return y.view(B, T, 3, H, W).permute(0, 2, 1, 3, 4)


# --------------------------------------------------------------------------- #
# discriminators   (discriminator.py:42-346)
# --------------------------------------------------------------------------- #
def idis_forward(st: State, xg, xc, use_noise: bool, sigma: float, rng, training: bool):
    """ImageDiscriminator.forward (discriminator.py:107-127). Draw order: g, c, main x3."""
    hg = _lrelu(F.conv2d(_noise(xg, use_noise, sigma, rng), st["conv_g.1.weight"], None, 2, 1), 0.2)
    hc = _lrelu(F.conv2d(_noise(xc, use_noise, sigma, rng), st["conv_c.1.weight"], None, 2, 1), 0.2)
    h = torch.cat([hc, hg], 1)
    for conv, bn in ((1, 2), (5, 6)):
        h = F.conv2d(_noise(h, use_noise, sigma, rng), st[f"main.{conv}.weight"], None, 2, 1)
        h = _lrelu(_bn(st, f"main.{bn}", h, training), 0.2)
    h = F.conv2d(_noise(h, use_noise, sigma, rng), st["main.9.weight"], None, 2, 1)
    return h.squeeze()


_S3 = (1, 2, 2)
_P3 = (0, 1, 1)


def vdis_forward(st: State, xg, xc, use_noise: bool, sigma: float, rng, training: bool):
    """VideoDiscriminator.forward (discriminator.py:211-231); stems carry no Noise."""
    hg = _lrelu(F.conv3d(xg, st["conv_g.0.weight"], None, _S3, _P3), 0.2)
    hc = _lrelu(F.conv3d(xc, st["conv_c.0.weight"], None, _S3, _P3), 0.2)
    h = torch.cat([hc, hg], 1)
    for conv, bn in ((1, 2), (5, 6)):
        h = F.conv3d(_noise(h, use_noise, sigma, rng), st[f"main.{conv}.weight"], None, _S3, _P3)
        h = _lrelu(_bn(st, f"main.{bn}", h, training), 0.2)
    h = F.conv3d(_noise(h, use_noise, sigma, rng), st["main.9.weight"], None, _S3, _P3)
    return h.squeeze()


def gdis_forward(st: State, xg, xc, use_noise: bool, sigma: float, rng, training: bool):
    """GradientDiscriminator.forward (discriminator.py:310-333); xc is ignored."""
    L = xg.shape[2]
    h = xg[:, :, 1:L] - xg[:, :, 0:L - 1]
    for conv, bn in ((1, 2), (5, 6), (9, 10)):
        h = F.conv3d(_noise(h, use_noise, sigma, rng), st[f"main.{conv}.weight"], None, _S3, _P3)
        h = _lrelu(_bn(st, f"main.{bn}", h, training), 0.2)
    h = F.conv3d(_noise(h, use_noise, sigma, rng), st["main.13.weight"], None, _S3, _P3)
    return h.squeeze()


# --------------------------------------------------------------------------- #
# losses   (loss.py:64-193)
# --------------------------------------------------------------------------- #
def _bce_mean(y: torch.Tensor, target: float) -> torch.Tensor:
    t = torch.full_like(y, target)
    return F.binary_cross_entropy_with_logits(y, t, reduction="sum") / y.numel()


def dis_loss(kind: str, y_real, y_fake):
    if kind == "adversarial-loss":  # loss.py:91-99
        return _bce_mean(y_real, 1.0) + _bce_mean(y_fake, 0.0)
    if kind == "hinge-loss":  # loss.py:163-164
        return torch.mean(F.relu(1.0 - y_real)) + torch.mean(F.relu(1.0 + y_fake))
    raise ValueError(kind)


def gen_loss(kind: str, y_i, y_v, y_g):
    if kind == "adversarial-loss":  # loss.py:123-131
        return _bce_mean(y_i, 1.0) + _bce_mean(y_v, 1.0) + _bce_mean(y_g, 1.0)
    if kind == "hinge-loss":  # loss.py:190-191 — y_g unused
        return torch.mean(F.softplus(-y_i)) + torch.mean(F.softplus(-y_v))
    raise ValueError(kind)


# --------------------------------------------------------------------------- #
# state construction   (PyTorch default init + util.init_weights :186-195)
# --------------------------------------------------------------------------- #
def _conv_default_(w: torch.Tensor) -> torch.Tensor:
    # nn.Conv*/ConvTranspose* reset_parameters: kaiming_uniform_(a=sqrt(5))
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    return w


def _bn_entries(st: State, prefix: str, c: int, two_d: bool):
    st[prefix + ".weight"] = torch.ones(c)
    st[prefix + ".bias"] = torch.zeros(c)
    st[prefix + ".running_mean"] = torch.zeros(c)
    st[prefix + ".running_var"] = torch.ones(c)
    st[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    if two_d:  # init_weights touches BatchNorm2d only
        st[prefix + ".weight"].normal_(1.0, 0.02)
        st[prefix + ".bias"].fill_(0.0)


def make_state(model: str, **kw) -> State:
    """Fresh parameters/buffers with the reference's key names and shapes
    (SURVEY §8(b)).  Values follow the same *distributions* as the reference's
    init; bit-equal init is not needed because parity runs load a fixture."""
    st: State = {}
    if model == "ggen":
        dz, dm, ngf, ch = kw["dim_z_content"] + kw["dim_z_motion"], kw["dim_z_motion"], kw["ngf"], kw["channel"]
        k = 1.0 / math.sqrt(dm)
        for n, shp in (("weight_ih", (3 * dm, dm)), ("weight_hh", (3 * dm, dm)), ("bias_ih", (3 * dm,)), ("bias_hh", (3 * dm,))):
            st["recurrent." + n] = torch.empty(shp).uniform_(-k, k)
        chans = [dz, ngf * 8, ngf * 4, ngf * 2, ngf, ch]
        for li, idx in enumerate((0, 3, 6, 9, 12)):
            st[f"main.{idx}.weight"] = torch.empty(chans[li], chans[li + 1], 4, 4).normal_(0, 0.02)
            if li < 4:
                _bn_entries(st, f"main.{idx + 1}", chans[li + 1], True)
    elif model == "cgen":
        ngf, cin, dz = kw["ngf"], kw["in_ch"], kw["dim_z"]
        st["inconv.main.0.weight"] = torch.empty(ngf, cin, 3, 3).normal_(0, 0.02)
        down = [(1, 1), (1, 2), (2, 4), (4, 4), (4, 4), (4, 4)]
        for i, (a, b) in enumerate(down):
            st[f"down_blocks.{i}.main.0.weight"] = torch.empty(ngf * b, ngf * a, 4, 4).normal_(0, 0.02)
            _bn_entries(st, f"down_blocks.{i}.main.1", ngf * b, True)
        up = [(ngf * 4 + dz, ngf * 4), (ngf * 8, ngf * 4), (ngf * 8, ngf * 4), (ngf * 8, ngf * 2), (ngf * 4, ngf), (ngf * 2, ngf)]
        for i, (a, b) in enumerate(up):
            st[f"up_blocks.{i}.main.0.weight"] = torch.empty(a, b, 4, 4).normal_(0, 0.02)
            _bn_entries(st, f"up_blocks.{i}.main.1", b, True)
        st["outconv.main.0.weight"] = torch.empty(ngf * 2, 3, 3, 3).normal_(0, 0.02)
    elif model == "idis":
        ndf, c1, c2 = kw["ndf"], kw["ch1"], kw["ch2"]
        st["conv_g.1.weight"] = torch.empty(ndf // 2, c1, 4, 4).normal_(0, 0.02)
        st["conv_c.1.weight"] = torch.empty(ndf // 2, c2, 4, 4).normal_(0, 0.02)
        st["main.1.weight"] = torch.empty(ndf * 2, ndf, 4, 4).normal_(0, 0.02)
        _bn_entries(st, "main.2", ndf * 2, True)
        st["main.5.weight"] = torch.empty(ndf * 4, ndf * 2, 4, 4).normal_(0, 0.02)
        _bn_entries(st, "main.6", ndf * 4, True)
        st["main.9.weight"] = torch.empty(1, ndf * 4, 4, 4).normal_(0, 0.02)
    elif model == "vdis":
        ndf, c1, c2 = kw["ndf"], kw["ch1"], kw["ch2"]
        st["conv_g.0.weight"] = _conv_default_(torch.empty(ndf // 2, c1, 4, 4, 4))
        st["conv_c.0.weight"] = _conv_default_(torch.empty(ndf // 2, c2, 4, 4, 4))
        st["main.1.weight"] = _conv_default_(torch.empty(ndf * 2, ndf, 4, 4, 4))
        _bn_entries(st, "main.2", ndf * 2, False)
        st["main.5.weight"] = _conv_default_(torch.empty(ndf * 4, ndf * 2, 4, 4, 4))
        _bn_entries(st, "main.6", ndf * 4, False)
        st["main.9.weight"] = _conv_default_(torch.empty(1, ndf * 4, 4, 4, 4))
    elif model == "gdis":
        ndf, c1 = kw["ndf"], kw["ch1"]
        chans = [c1, ndf, ndf * 2, ndf * 4]
        for li, idx in enumerate((1, 5, 9)):
            st[f"main.{idx}.weight"] = _conv_default_(torch.empty(chans[li + 1], chans[li], 4, 4, 4))
            _bn_entries(st, f"main.{idx + 1}", chans[li + 1], False)
        st["main.13.weight"] = _conv_default_(torch.empty(1, ndf * 4, 4, 4, 4))
    else:
        raise ValueError(model)
    return st


def trainable(st: State) -> List[torch.Tensor]:
    """Parameters in nn.Module.parameters() order == insertion order minus buffers."""
    return [v for k, v in st.items()
            if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def require_grad(st: State) -> State:
    for p in trainable(st):
        p.requires_grad_(True)
    return st


# --------------------------------------------------------------------------- #
# one trainer iteration   (trainer.py:279-363)
# --------------------------------------------------------------------------- #
class StepOracle:
    """Holds the five states + five torch Adam optimisers (train.py:171-176) and
    advances the reference's per-iteration schedule."""

    def __init__(self, cfg, states: Dict[str, State], rng=None):
        self.cfg = cfg
        self.st = {k: require_grad(v) for k, v in states.items()}
        self.rng = rng or TorchRng()
        self.opt = {
            name: torch.optim.Adam(trainable(self.st[name]), lr=cfg.lr[name],
                                   betas=(0.5, 0.999), weight_decay=cfg.decay[name])
            for name in ("ggen", "cgen", "idis", "vdis", "gdis")
        }
        self.iteration = 0
        # trainer.py:266-267 leaves ggen/cgen in eval() until the first G phase.
        self.gen_training = not getattr(cfg, "start_in_eval", False)

    # -- pieces ------------------------------------------------------------- #
    def fakes(self):
        c = self.cfg
        seg = c.geometric_info == "segmentation"
        xg = ggen_sample_videos(self.st["ggen"], c.batchsize, c.video_length, c.dim_z_content,
                                c.dim_z_motion, c.channel, self.rng, self.gen_training, seg)
        xc = cgen_forward_videos(self.st["cgen"], xg, c.dim_z_color, self.rng, self.gen_training, seg)
        return xg, xc

    def dis_all(self, xg, xc, t):
        c = self.cfg
        yi = idis_forward(self.st["idis"], xg[:, :, t], xc[:, :, t], c.use_noise["idis"], c.noise_sigma["idis"], self.rng, True)
        yv = vdis_forward(self.st["vdis"], xg, xc, c.use_noise["vdis"], c.noise_sigma["vdis"], self.rng, True)
        yg = gdis_forward(self.st["gdis"], xg, xc, c.use_noise["gdis"], c.noise_sigma["gdis"], self.rng, True)
        return yi, yv, yg

    def _zero(self, names):
        for n in names:
            for p in trainable(self.st[n]):
                p.grad = None

    # -- the iteration ------------------------------------------------------ #
    def step(self, xc_real: torch.Tensor, xg_real: torch.Tensor, t_rand: int) -> Dict[str, float]:
        c = self.cfg
        self.iteration += 1
        # D phase (trainer.py:285-328)
        self._zero(("idis", "vdis", "gdis"))
        yr = self.dis_all(xg_real, xc_real, t_rand)
        xg_f, xc_f = self.fakes()  # NOT detached (:304-305)
        yf = self.dis_all(xg_f, xc_f, t_rand)
        l_i = dis_loss(c.loss, yr[0], yf[0])
        l_v = dis_loss(c.loss, yr[1], yf[1])
        l_g = dis_loss(c.loss, yr[2], yf[2])
        l_d = l_i + l_v + l_g
        if self.iteration % c.num_gen_update == 0:
            l_d.backward()
            self.opt["idis"].step(); self.opt["vdis"].step(); self.opt["gdis"].step()
        # G phase (trainer.py:338-363)
        self.gen_training = True
        self._zero(("ggen", "cgen"))
        xg_f, xc_f = self.fakes()
        yf = self.dis_all(xg_f, xc_f, t_rand)
        l_gen = gen_loss(c.loss, *yf)
        if self.iteration % c.num_dis_update == 0:
            l_gen.backward()
            self.opt["ggen"].step(); self.opt["cgen"].step(); self.opt["ggen"].step()  # :357-359
        return {"loss_idis": l_i.item(), "loss_vdis": l_v.item(), "loss_gdis": l_g.item(), "loss_gen": l_gen.item()}


# --------------------------------------------------------------------------- #
# sampling path   (util.py:31-79, 198-322)
# --------------------------------------------------------------------------- #
def videos_to_uint8(x: torch.Tensor):
    """util.videos_to_numpy (util.py:58-79): clip, (x+1)/2*255 in fp32, astype(uint8)."""
    import numpy as np
    v = np.clip(x.detach().cpu().numpy(), -1, 1)
    return ((v + 1) / 2 * 255).astype("uint8")


def depth_to_color(xg: torch.Tensor):
    """util.generate_samples :306-308 + geometric_info_in_color_format :219-222 (depth)."""
    import numpy as np
    v = np.clip(xg.detach().cpu().numpy(), -1, 1)
    v = np.tile(v, (1, 3, 1, 1, 1))
    return ((v + 1) / 2 * 255).astype("uint8")


def segmentation_to_color(xg, palette_u8):
    """util.geometric_info_in_color_format, segmentation branch (util.py:236-246): argmax over the part
    channels of (B,25,T,H,W) (first maximum), part colour from the palette; uint8 (B,3,T,H,W)."""
    import numpy as np
    idx = np.argmax(np.asarray(xg), axis=1)                       # (B,T,H,W)
    return np.asarray(palette_u8, dtype=np.uint8)[idx].transpose(0, 4, 1, 2, 3)


def segmentation_one_hot(labels, num_parts: int = 25):
    """dataset.py:176-181: label frames (T,H,W) -> one-hot float32 (25,T,H,W)."""
    import numpy as np
    return np.eye(num_parts, dtype=np.float32)[np.asarray(labels)].transpose(3, 0, 1, 2)


def generate_samples_depth(st_g: State, st_c: State, num: int, batchsize: int, T: int, dzc: int, dzm: int, dzcol: int, rng):
    """util.generate_samples (util.py:251-322) for depth geometry: eval mode, no_grad, truncation to num."""
    import numpy as np
    xgs, xcs = [], []
    with torch.no_grad():
        for _ in range(0, num, batchsize):
            xg = ggen_sample_videos(st_g, batchsize, T, dzc, dzm, 1, rng, False)
            xc = cgen_forward_videos(st_c, xg, dzcol, rng, False)
            xgs.append(depth_to_color(xg)); xcs.append(videos_to_uint8(xc))
    return np.concatenate(xgs)[:num], np.concatenate(xcs)[:num]
