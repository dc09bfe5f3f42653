"""The reference's training configurations, restated as literal tables.

Source: /root/reference/config/{debug-isogd-depth,isogd-depth,surreal-depth1,
isogd-flow}.yml (the four configs BASELINE.json names).  YAML parsing itself is
out of scope; only the values that reach the hot path are kept:
constructor arguments (train.py:117-156), Adam hyper-parameters
(train.py:171-176) and the update gating (trainer.py:318,355).

``surreal-depth1.yml`` ships no ``gdis:`` block although train.py:150-156
indexes it unconditionally; isogd's block is injected.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict


@dataclass
class StepConfig:
    name: str
    batchsize: int
    seed: int
    geometric_info: str          # "depth" | "optical-flow" | "segmentation"
    channel: int                 # geometry channels Cg
    loss: str                    # "adversarial-loss" | "hinge-loss"
    num_gen_update: int = 1
    num_dis_update: int = 1
    video_length: int = 16
    image_size: int = 64
    dim_z_content: int = 40
    dim_z_motion: int = 10
    dim_z_color: int = 10
    width: Dict[str, int] = field(default_factory=dict)        # ngf / ndf per model
    use_noise: Dict[str, bool] = field(default_factory=dict)
    noise_sigma: Dict[str, float] = field(default_factory=dict)
    lr: Dict[str, float] = field(default_factory=dict)
    decay: Dict[str, float] = field(default_factory=dict)
    start_in_eval: bool = False  # trainer.py:266-267 quirk (log_samples before iteration 1)

    def scaled(self, batchsize=None, width_div=1, **kw) -> "StepConfig":
        """Reduced copy for parity tests (same topology, thinner layers)."""
        w = {k: max(2, v // width_div) for k, v in self.width.items()}
        return replace(self, batchsize=batchsize or self.batchsize, width=w, **kw)


_DECAY = {m: 1e-5 for m in ("ggen", "cgen", "idis", "vdis", "gdis")}

CONFIGS: Dict[str, StepConfig] = {
    # config/debug-isogd-depth.yml
    "debug-isogd-depth": StepConfig(
        name="debug-isogd-depth", batchsize=2, seed=10, geometric_info="depth", channel=1,
        loss="adversarial-loss",
        width=dict(ggen=64, cgen=64, idis=64, vdis=32, gdis=32),
        use_noise=dict(idis=True, vdis=True, gdis=False),
        noise_sigma=dict(idis=0.1, vdis=0.1, gdis=0.2),
        lr=dict(ggen=2e-4, cgen=2e-4, idis=2e-4, vdis=2e-4, gdis=2e-4), decay=dict(_DECAY)),
    # config/isogd-depth.yml  (the headline config)
    "isogd-depth": StepConfig(
        name="isogd-depth", batchsize=70, seed=15, geometric_info="depth", channel=1,
        loss="adversarial-loss",
        width=dict(ggen=64, cgen=64, idis=64, vdis=64, gdis=32),
        use_noise=dict(idis=True, vdis=True, gdis=False),
        noise_sigma=dict(idis=0.1, vdis=0.1, gdis=0.2),
        lr=dict(ggen=2e-4, cgen=2e-4, idis=5e-4, vdis=5e-4, gdis=2e-4), decay=dict(_DECAY)),
    # config/surreal-depth1.yml (+ isogd's gdis block)
    "surreal-depth1": StepConfig(
        name="surreal-depth1", batchsize=100, seed=15, geometric_info="depth", channel=1,
        loss="hinge-loss", num_gen_update=2,
        width=dict(ggen=96, cgen=64, idis=64, vdis=64, gdis=32),
        use_noise=dict(idis=False, vdis=False, gdis=False),
        noise_sigma=dict(idis=0.2, vdis=0.2, gdis=0.2),
        lr=dict(ggen=2e-4, cgen=2e-4, idis=2e-4, vdis=2e-4, gdis=2e-4), decay=dict(_DECAY)),
    # config/surreal-segm.yml (25 body-part maps; the yml has no gdis section: trainer.py reads one that is
    # absent, so as with surreal-depth1 the gdis settings are injected — ndf 32, no noise).  SURVEY §8(f).4
    "surreal-segm": StepConfig(
        name="surreal-segm", batchsize=60, seed=15, geometric_info="segmentation", channel=25,
        loss="adversarial-loss", num_gen_update=2,
        width=dict(ggen=96, cgen=64, idis=64, vdis=48, gdis=32),
        use_noise=dict(idis=True, vdis=True, gdis=False),
        noise_sigma=dict(idis=0.2, vdis=0.2, gdis=0.2),
        lr=dict(ggen=2e-4, cgen=2e-4, idis=2e-4, vdis=2e-4, gdis=2e-4), decay=dict(_DECAY)),
    # config/isogd-flow.yml (as shipped: 16 x 64 x 64, two flow channels)
    "isogd-flow": StepConfig(
        name="isogd-flow", batchsize=100, seed=15, geometric_info="optical-flow", channel=2,
        loss="hinge-loss",
        width=dict(ggen=64, cgen=64, idis=64, vdis=64, gdis=32),
        use_noise=dict(idis=True, vdis=True, gdis=False),
        noise_sigma=dict(idis=0.2, vdis=0.2, gdis=0.2),
        lr=dict(ggen=2e-4, cgen=2e-4, idis=2e-4, vdis=2e-4, gdis=2e-4), decay=dict(_DECAY)),
}

# Conv / conv-transpose / GRU FLOPs (2*MACs) of ONE forward pass of each model per video, in GFLOP (SURVEY §8(d),
# BASELINE.md §4; torch.utils.flop_counter on the CPU restatement reproduces them): g = ggen, c = cgen, i / v / d = the
# image / video / gradient discriminators, d1 = the first convolution(s) of the three discriminators together (they
# have no data gradient when the input is a real batch).
FORWARD_GFLOP = {
    "debug-isogd-depth": dict(g=3.268, c=15.495, i=0.139, v=1.251, d=1.058, d1=0.170),
    "isogd-depth": dict(g=3.268, c=15.495, i=0.139, v=4.784, d=1.058, d1=0.287),
    "surreal-depth1": dict(g=7.318, c=15.495, i=0.139, v=4.784, d=1.058, d1=0.287),
    "isogd-flow": dict(g=3.301, c=15.571, i=0.140, v=4.838, d=1.108, d1=0.380),
    "surreal-segm": dict(g=7.6, c=18.23, i=0.16, v=3.1, d=1.9, d1=2.2),
}


def flops_per_video_iteration(cfg: "StepConfig", elide_dead_backward: bool = False) -> float:
    """Algorithmic FLOPs of one trainer iteration (trainer.py:279-363) per video, averaged over the update-gating cycle.

    forward            : ggen x2, cgen x2, every discriminator x3                           2G + 3D
    D-phase backward   : D(real) wgrad + dgrad past the first convs, D(fake) wgrad + dgrad,   (2D - d1) + 2D + 2G
                         and — the fakes are not detached — the generators' wgrad + dgrad     [every num_gen_update-th iteration]
    G-phase backward   : the discriminators the loss uses (hinge ignores gdis), generators    2 D_g + 2G   [every num_dis_update-th]
    `elide_dead_backward` drops the D phase's generator backward and the data gradient of the fakes' first convs."""
    f = FORWARD_GFLOP[cfg.name]
    G, D = f["g"] + f["c"], f["i"] + f["v"] + f["d"]
    Dg = D if cfg.loss == "adversarial-loss" else f["i"] + f["v"]
    fwd = 2 * G + 3 * D
    bwd_d = (2 * D - f["d1"]) + 2 * D + 2 * G
    if elide_dead_backward:
        bwd_d -= 2 * G + f["d1"]
    bwd_g = 2 * Dg + 2 * G
    return (fwd + bwd_d / cfg.num_gen_update + bwd_g / cfg.num_dis_update) * 1e9


# one forward of every model touches ~0.158 GB of conv tensors per video (11.04 GB at B = 70, fp32); the iteration's
# multiplicities make that ~1.06 GB per video: the algorithmic HBM floor the step's traffic is held against
ALGORITHMIC_HBM_GB_PER_VIDEO_ITERATION = 1.06
