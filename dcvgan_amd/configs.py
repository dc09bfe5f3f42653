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

# Conv / conv-transpose / GRU FLOPs (2*MACs) for ONE video through ONE as-written
# G+D step, and with the dead D-phase generator backward elided.
FLOPS_PER_VIDEO_STEP = {
    "debug-isogd-depth": (134.44e9, 96.75e9),
    "isogd-depth": (166.12e9, 128.33e9),
    "surreal-depth1": (188.31e9, 142.41e9),
    "isogd-flow": (165.41e9, 127.29e9),
    # torch.utils.flop_counter on a CPU restatement of the step (reproduces 166.12e9 for isogd-depth); minimal = as-written
    # minus the dead generator backward (2 x 25.83e9 generator forward)
    "surreal-segm": (204.08e9, 152.4e9),
}
