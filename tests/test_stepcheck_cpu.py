"""CPU: the composed-iteration checker (oracle/stepcheck.py) applied to the REFERENCE's own arithmetic — the pinned fp32 torch-CPU
oracle plays the checked implementation.  Shows (a) the checker's bars are ones the reference's fp32 path itself meets, so they are a
statement about arithmetic, not about which fp32 lottery ticket a run drew, and (b) it fails when something is actually wrong (a wrong
learning rate, a skipped step).  Test widths (ngf = ndf = 6), 3 iterations, adversarial loss + Noise, and hinge with num_gen_update 2."""
import types

import pytest
import torch

from oracle import dcvgan_oracle as O
from oracle import stepcheck as SC
from tests import goldenio as G


class _Fp32Runner:
    """trainer.StepRunner's duck type over the fp32 StepOracle; `layers`-like tap for its branch patterns."""

    def __init__(self, cfg, states):
        self.so = O.StepOracle(cfg, states)
        self.KINK_TAP = None
        self.models = {n: types.SimpleNamespace(state_dict=lambda n=n: self.so.st[n], training=True) for n in SC.MODELS}
        self.opts = {n: types.SimpleNamespace(params=O.trainable(self.so.st[n]), state=self.so.opt[n].state) for n in SC.MODELS}
        gen_mode = not getattr(cfg, "start_in_eval", False)
        self.models["ggen"].training = self.models["cgen"].training = gen_mode

    @property
    def iteration(self):
        return self.so.iteration

    def step(self, xc, xg, t):
        with O.KinkTape() as tape:
            out = self.so.step(xc, xg, t)
        if self.KINK_TAP is not None:
            self.KINK_TAP.extend(tape.recorded)
        self.models["ggen"].training = self.models["cgen"].training = True
        return out


def _setup(fixture):
    fx = G.load(fixture)
    cfg = G.cfg_of(fx, loss=str(fx["meta/loss"]), num_gen_update=int(fx["meta/num_gen_update"]), num_dis_update=int(fx["meta/num_dis_update"]),
                   start_in_eval=bool(fx["meta/start_in_eval"]))
    B = cfg.batchsize
    gd = torch.Generator().manual_seed(int(fx["meta/seed_data"]))
    lo, hi = (-0.5, 0.5) if cfg.channel == 2 else (-1.0, 1.0)
    xc = torch.rand(B, 3, 16, 64, 64, generator=gd) * 2 - 1
    xg = torch.rand(B, cfg.channel, 16, 64, 64, generator=gd) * (hi - lo) + lo
    return fx, cfg, xc, xg


@pytest.mark.parametrize("fixture", ["step_depth_adv_g1.npz", "step_depth_adv_g1_evalstart.npz", "step_flow_hinge_g2.npz"])
def test_reference_arithmetic_meets_the_step_bars(fixture):
    fx, cfg, xc, xg = _setup(fixture)
    torch.manual_seed(int(fx["meta/seed_run"]))
    run = _Fp32Runner(cfg, G.states(fx))
    forced = SC.ForcedStepOracle(cfg, run.so.rng.log)       # the log grows as the fp32 run draws; the replay reads behind it
    for it in range(int(fx["meta/iters"])):
        res = SC.checked_iteration(run, run.models, run.opts, forced, run, xc, xg, xc, xg, int(fx["meta/t_rands"][it]), cfg.lr)
        assert [res["losses"][k] for k in ("loss_idis", "loss_vdis", "loss_gdis", "loss_gen")] == pytest.approx(list(fx["losses"][it]), rel=1e-6)  # it IS the reference's run
        SC.assert_iteration(res, cfg.lr, f"{fixture} it {it + 1}")
    assert forced.rng.pos == len(run.so.rng.log)


def test_checker_catches_a_wrong_learning_rate_and_a_skipped_step():
    fx, cfg, xc, xg = _setup("step_depth_adv_g1.npz")
    torch.manual_seed(int(fx["meta/seed_run"]))
    run = _Fp32Runner(cfg, G.states(fx))
    for g in run.so.opt["cgen"].param_groups:
        g["lr"] *= 1.01                                      # 1 % off: far inside the old 5 % norm / cos > 0.9 bars
    forced = SC.ForcedStepOracle(cfg, run.so.rng.log)
    res = SC.checked_iteration(run, run.models, run.opts, forced, run, xc, xg, xc, xg, 3, cfg.lr)
    with pytest.raises(AssertionError):
        SC.assert_iteration(res, cfg.lr)
    bad = [r for r in res["rows"] if r["rel_l2"] > SC.UPDATE_TOL]
    assert bad and all(r["model"] == "cgen" for r in bad)
    # the double ggen step taken once
    torch.manual_seed(int(fx["meta/seed_run"]))
    run = _Fp32Runner(cfg, G.states(fx))
    inner, n = run.so.opt["ggen"].step, [0]

    def once():
        n[0] += 1
        if n[0] == 1:
            inner()
    run.so.opt["ggen"].step = once
    forced = SC.ForcedStepOracle(cfg, run.so.rng.log)
    res = SC.checked_iteration(run, run.models, run.opts, forced, run, xc, xg, xc, xg, 3, cfg.lr)
    with pytest.raises(AssertionError):
        SC.assert_iteration(res, cfg.lr)
