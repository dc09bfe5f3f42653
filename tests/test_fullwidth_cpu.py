"""CPU: the oracle at the REAL channel widths of the three GPU configs (isogd-depth, surreal-depth1 with ggen ngf 96
and hinge, isogd-flow with two flow channels) and on the 32 x 128 x 128 discriminator stress shape, against
fixtures from the reference classes (tests/golden/make_golden.py: fullwidth_fixture, step_fullwidth_fixture,
stress_d_fixture).  Pins the oracle where the -m gpu tests use it as the full-tensor reference."""
import numpy as np
import pytest
import torch

from oracle import dcvgan_oracle as O
from tests import fullwidth as FW
from tests import goldenio as G

FIX = ["fullwidth_isogd_depth.npz", "fullwidth_surreal_depth1.npz", "fullwidth_isogd_flow.npz"]


@pytest.mark.parametrize("fixture", FIX)
def test_fullwidth_generator_pass(fixture):
    fx = G.load(fixture)
    cfg, models = FW.same_seed_models(fx)
    r = FW.oracle_gen_pass(cfg, models, int(fx["meta/seed_run"]), int(fx["meta/t_rand"]))
    assert tuple(r["xg"].stride()) == tuple(fx["xg_stride"]) and tuple(r["xc"].stride()) == tuple(fx["xc_stride"])
    assert np.allclose(G.summ(r["xg"]), fx["xg_sum"], rtol=1e-5) and np.allclose(G.summ(r["xc"]), fx["xc_sum"], rtol=1e-5)
    for k in ("yi", "yv", "yg"):
        assert G.relerr(r[k].numpy(), fx[k]) < 2e-5, k
    assert abs(r["loss"].item() - float(fx["loss_gen"])) < 2e-6 * abs(float(fx["loss_gen"])) + 1e-7
    n = 0
    for (m, k), g in r["grads"].items():
        if f"gradnone/{m}/{k}" in fx:
            assert g is None, (m, k)      # hinge: gdis receives no gradient (loss.py:190-191)
            continue
        assert abs(g.double().norm().item() - float(fx[f"gradnorm/{m}/{k}"])) < 1e-4 * float(fx[f"gradnorm/{m}/{k}"]) + 1e-12, (m, k)
        assert G.relerr(FW.gsub(g), fx[f"gradsub/{m}/{k}"]) < 1e-4, (m, k)
        n += 1
    assert n > 60


@pytest.mark.parametrize("fixture", ["step_fullwidth_isogd_depth.npz", "step_fullwidth_surreal_depth1.npz", "step_fullwidth_isogd_flow.npz"])
def test_fullwidth_training_step(fixture):
    """Two iterations of trainer.py:279-363 at full width (isogd-depth: BCE + Noise, the headline config; hinge for the other two; surreal: the D
    update runs every 2nd iteration)."""
    fx = G.load(fixture)
    cfg, models = FW.same_seed_models(fx)
    cfg.num_gen_update = int(fx["meta/num_gen_update"])
    cfg.lr = {m: float(fx[f"meta/lr/{m}"]) for m in G.MODELS}
    B = cfg.batchsize
    gd = torch.Generator().manual_seed(int(fx["meta/seed_data"]))
    lo, hi = (-0.5, 0.5) if cfg.channel == 2 else (-1.0, 1.0)
    xc_real = torch.rand(B, 3, 16, 64, 64, generator=gd) * 2 - 1
    xg_real = torch.rand(B, cfg.channel, 16, 64, 64, generator=gd) * (hi - lo) + lo
    torch.manual_seed(int(fx["meta/seed_run"]))
    so = O.StepOracle(cfg, FW.states_of(models))
    for it in range(1, int(fx["meta/iters"]) + 1):
        before = {n: {k: v.detach().clone() for k, v in so.st[n].items()} for n in G.MODELS}
        r = so.step(xc_real, xg_real, int(fx["meta/t_rands"][it - 1]))
        got = [r["loss_idis"], r["loss_vdis"], r["loss_gdis"], r["loss_gen"]]
        assert np.allclose(got, fx["losses"][it - 1], rtol=2e-5, atol=1e-6), (it, got, fx["losses"][it - 1])
        for n in G.MODELS:
            for k, v in so.st[n].items():
                if f"delta{it}/{n}/{k}/norm" not in fx:
                    continue
                d = v.detach() - before[n][k]
                ref = float(fx[f"delta{it}/{n}/{k}/norm"])
                assert abs(d.double().norm().item() - ref) <= 2e-3 * ref + 1e-12, (it, n, k)
                if it % cfg.num_gen_update != 0 and n.endswith("dis"):
                    assert ref == 0.0        # gated off: the discriminators did not move
                else:
                    assert ref > 0.0
                    assert G.relerr(FW.gsub(d), fx[f"delta{it}/{n}/{k}/sub"]) < 2e-2, (it, n, k)


def test_stress_shape_discriminators():
    """32 x 128 x 128 clips straight into vdis / gdis, B = 1."""
    fx = G.load("stress_d_32x128x128.npz")
    from dcvgan_amd import discriminator as D, util
    torch.manual_seed(int(fx["meta/seed_init"]))
    vdis = D.VideoDiscriminator(2, 3, True, 0.2, 64); gdis = D.GradientDiscriminator(2, 3, False, 0.2, 32)
    st = {}
    for n, m in (("vdis", vdis), ("gdis", gdis)):
        m.apply(util.init_weights)
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                assert np.allclose(G.summ(v), fx[f"init_sum/{n}/{k}"], rtol=1e-6, atol=1e-6), (n, k)
        st[n] = O.require_grad({k: v.detach().clone() for k, v in m.state_dict().items()})
    g = torch.Generator().manual_seed(int(fx["meta/seed_inputs"]))
    xg = (torch.rand(1, 32, 2, 128, 128, generator=g) - 0.5).permute(0, 2, 1, 3, 4).requires_grad_(True)
    xc = (torch.rand(1, 32, 3, 128, 128, generator=g) * 2 - 1).permute(0, 2, 1, 3, 4).requires_grad_(True)
    torch.manual_seed(int(fx["meta/seed_fwd"]))
    rng = O.TorchRng()
    yv = O.vdis_forward(st["vdis"], xg, xc, True, 0.2, rng, True)
    yg = O.gdis_forward(st["gdis"], xg, xc, False, 0.2, rng, True)
    assert tuple(yv.shape) == (20, 8, 8) and tuple(yg.shape) == (19, 8, 8)
    assert G.relerr(yv.detach().numpy(), fx["yv"]) < 2e-5 and G.relerr(yg.detach().numpy(), fx["yg"]) < 2e-5
    ((yv * torch.linspace(1, -1, yv.numel()).view(yv.shape)).sum() + (yg * torch.linspace(-0.5, 1.5, yg.numel()).view(yg.shape)).sum()).backward()
    assert G.relerr(G.sub(xg.grad, 101), fx["grad_xg_sub"]) < 1e-4 and G.relerr(G.sub(xc.grad, 101), fx["grad_xc_sub"]) < 1e-4
    for n in ("vdis", "gdis"):
        for k, p in st[n].items():
            if p.requires_grad:
                assert G.relerr(FW.gsub(p.grad), fx[f"gradsub/{n}/{k}"]) < 1e-4, (n, k)
