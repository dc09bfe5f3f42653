"""Shared by the full-width parity tests: same-seed model construction checked against the fixture's
init checksums, and one generator-loss forward/backward of the oracle in fp32 or fp64 with the SAME draws."""
import numpy as np
import torch

from oracle import dcvgan_oracle as O
from tests import goldenio as G

MODELS = G.MODELS


def same_seed_models(fx, B=None):
    """The reference builds its five models in a fixed order under one seed (train.py:117-165); the HIP-backed
    classes consume the init stream identically, so the fixture only stores checksums of the initial tensors."""
    from dcvgan_amd import trainer
    cfg = G.cfg_of(fx, B=B, loss=str(fx["meta/loss"]) if "meta/loss" in fx else "adversarial-loss")
    torch.manual_seed(int(fx["meta/seed_init"]))
    models = trainer.build_models(cfg, torch.device("cpu"))
    for n, m in models.items():
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                assert np.allclose(G.summ(v), fx[f"init_sum/{n}/{k}"], rtol=1e-6, atol=1e-6), (n, k)
    return cfg, models


def states_of(models, dtype=torch.float32):
    st = {}
    for n, m in models.items():
        st[n] = {k: (v.detach().cpu().clone().to(dtype) if v.dtype.is_floating_point else v.detach().cpu().clone()) for k, v in m.state_dict().items()}
        O.require_grad(st[n])
    return st


class Rng64(O.TorchRng):
    """The fp32 draws of TorchRng, widened: the fp64 graph sees exactly the same random numbers."""

    def normal(self, shape):
        return super().normal(shape).double()


def oracle_gen_pass(cfg, models, seed_run, t, dtype=torch.float32, kinks=None):
    """ggen -> cgen -> idis/vdis/gdis -> compute_gen_loss -> backward (the G phase of trainer.py:344-356).
    `kinks`: (Leaky)ReLU branch patterns recorded from another evaluation (oracle.KinkTape) to differentiate with."""
    if kinks is not None:
        with O.KinkTape(kinks) as tape:
            r = oracle_gen_pass(cfg, models, seed_run, t, dtype)
        assert tape.pos == len(kinks), (tape.pos, len(kinks))
        r["kink_mismatch"] = tape.mismatch
        return r
    st = states_of(models, dtype)
    torch.manual_seed(seed_run)
    rng = O.TorchRng() if dtype == torch.float32 else Rng64()
    B = cfg.batchsize
    seg = cfg.geometric_info == "segmentation"
    xg = O.ggen_sample_videos(st["ggen"], B, 16, cfg.dim_z_content, cfg.dim_z_motion, cfg.channel, rng, True, seg)
    xc = O.cgen_forward_videos(st["cgen"], xg, cfg.dim_z_color, rng, True, seg)
    yi = O.idis_forward(st["idis"], xg[:, :, t], xc[:, :, t], cfg.use_noise["idis"], cfg.noise_sigma["idis"], rng, True)
    yv = O.vdis_forward(st["vdis"], xg, xc, cfg.use_noise["vdis"], cfg.noise_sigma["vdis"], rng, True)
    yg = O.gdis_forward(st["gdis"], xg, xc, cfg.use_noise["gdis"], cfg.noise_sigma["gdis"], rng, True)
    loss = O.gen_loss(cfg.loss, yi, yv, yg)
    loss.backward()
    grads = {}
    for n in MODELS:
        for k, p in st[n].items():
            if p.requires_grad:
                grads[(n, k)] = None if p.grad is None else p.grad.detach()
    return dict(xg=xg.detach(), xc=xc.detach(), yi=yi.detach(), yv=yv.detach(), yg=yg.detach(), loss=loss.detach(), grads=grads, log=rng.log, st=st)


def oracle_dis_pass(cfg, models, seed_run, t, xc_real, xg_real, dtype=torch.float32, kinks=None):
    """The D phase of trainer.py:285-319: D(real), fakes WITH their graph (not detached), D(fake), the three discriminator losses,
    one backward — gradients of all five models (the generators' are the 'dead' ones the reference computes and discards)."""
    if kinks is not None:
        with O.KinkTape(kinks) as tape:
            r = oracle_dis_pass(cfg, models, seed_run, t, xc_real, xg_real, dtype)
        assert tape.pos == len(kinks), (tape.pos, len(kinks))
        r["kink_mismatch"] = tape.mismatch
        return r
    st = states_of(models, dtype)
    torch.manual_seed(seed_run)
    so = O.StepOracle(cfg, st, O.TorchRng() if dtype == torch.float32 else Rng64())
    xc_r, xg_r = xc_real.to(dtype), xg_real.to(dtype)
    yr = so.dis_all(xg_r, xc_r, t)
    xg_f, xc_f = so.fakes()
    yf = so.dis_all(xg_f, xc_f, t)
    losses = [O.dis_loss(cfg.loss, a, b) for a, b in zip(yr, yf)]
    (losses[0] + losses[1] + losses[2]).backward()
    grads = {(n, k): (None if p.grad is None else p.grad.detach()) for n in MODELS for k, p in st[n].items() if p.requires_grad}
    return dict(losses=[l.detach() for l in losses], grads=grads, log=so.rng.log)


def hip_dis_pass(cfg, models, log, t, xc_real, xg_real, dev):
    from dcvgan_amd import layers, trainer
    from dcvgan_amd.rng import InjectedRng
    to_device(models, dev)
    for m in models.values():
        m.zero_grad()
        m.train()
    r = InjectedRng(log)
    for m in models.values():
        m._rng = r
    layers.KINK_TAP = kinks = []
    xc_r, xg_r = xc_real.to(dev), xg_real.to(dev)
    B = cfg.batchsize
    y_real = (models["idis"](xg_r[:, :, t], xc_r[:, :, t]), models["vdis"](xg_r, xc_r), models["gdis"](xg_r, xc_r))
    xg = models["ggen"].sample_videos(B); xc = models["cgen"].forward_videos(xg)
    y_fake = (models["idis"](xg[:, :, t], xc[:, :, t]), models["vdis"](xg, xc), models["gdis"](xg, xc))
    layers.KINK_TAP = None
    L = trainer.build_loss(cfg)
    losses = [L.compute_dis_loss(a, b) for a, b in zip(y_real, y_fake)]
    (losses[0] + losses[1] + losses[2]).backward()
    assert r.pos == len(log)
    grads = {(n, k): (None if p.grad is None else p.grad.detach().cpu()) for n in MODELS for k, p in models[n].named_parameters()}
    return dict(losses=[l.detach().cpu() for l in losses], grads=grads, kinks=kinks)


def to_device(models, dev):
    for m in models.values():
        m.to(dev)
        for mod in m.modules():
            if hasattr(mod, "device"):
                mod.device = dev
    return models


def hip_gen_pass(cfg, models, log, t, dev):
    from dcvgan_amd import layers, trainer
    from dcvgan_amd.rng import InjectedRng
    to_device(models, dev)
    layers.KINK_TAP = kinks = []
    r = InjectedRng(log)
    for m in models.values():
        m._rng = r
    B = cfg.batchsize
    xg = models["ggen"].sample_videos(B); xc = models["cgen"].forward_videos(xg)
    yi = models["idis"](xg[:, :, t], xc[:, :, t]); yv = models["vdis"](xg, xc); yg = models["gdis"](xg, xc)
    loss = trainer.build_loss(cfg).compute_gen_loss(yi, yv, yg)
    layers.KINK_TAP = None
    loss.backward()
    assert r.pos == len(log)
    grads = {(n, k): (None if p.grad is None else p.grad.detach().cpu()) for n in MODELS for k, p in models[n].named_parameters()}
    return dict(xg=xg.detach(), xc=xc.detach(), yi=yi.detach().cpu(), yv=yv.detach().cpu(), yg=yg.detach().cpu(), loss=loss.detach().cpu(), grads=grads,
                kinks=kinks)


def gsub(t):
    v = t.detach().reshape(-1)
    return v[::max(1, v.numel() // 256)].cpu().numpy()


# --------------------------------------------------------------------------- #
# optimiser wiring of a training iteration, checked without the kink lottery
# --------------------------------------------------------------------------- #
def recording_optimizers(cfg, models):
    """dcvgan_amd.optim.Adam for each model, recording at every .step(): the gradients it was handed and the
    parameters / moments / step counts before and after."""
    from dcvgan_amd import optim, trainer
    opts = trainer.build_optimizers(cfg, models)
    calls = []

    def wrap(name, o):
        inner = o.step

        def step():
            pre = [(p.detach().cpu().clone(), None if p.grad is None else p.grad.detach().cpu().clone(),
                    {k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in o.state.get(p, {}).items()}) for p in o.params]
            inner()
            calls.append((name, pre, [p.detach().cpu().clone() for p in o.params]))
        o.step = step
    for n, o in opts.items():
        assert isinstance(o, optim.Adam)
        wrap(n, o)
    return opts, calls


def check_optimizer_calls(cfg, calls, iteration, lrs):
    """(1) the schedule of trainer.py:318-322,355-359: idis, vdis, gdis once when the D update is due, then ggen, cgen,
    ggen; (2) every call moved its parameters EXACTLY as torch.optim.Adam(betas=(0.5, 0.999), eps 1e-8, weight_decay
    1e-5) (train.py:171-176) moves them from the same parameters, moments, step counts and gradients — to 1e-3 of the update, see below."""
    want = (["idis", "vdis", "gdis"] if iteration % cfg.num_gen_update == 0 else []) + (["ggen", "cgen", "ggen"] if iteration % cfg.num_dis_update == 0 else [])
    assert [c[0] for c in calls] == want, ([c[0] for c in calls], want)
    worst, where = 0.0, None
    for ci, (name, pre, post) in enumerate(calls):
        ps = [torch.nn.Parameter(t.clone()) for t, _, _ in pre]
        ref = torch.optim.Adam(ps, lr=lrs[name], betas=(0.5, 0.999), eps=1e-8, weight_decay=cfg.decay[name] if getattr(cfg, "decay", None) else 1e-5)
        for q, (_, g, st) in zip(ps, pre):
            q.grad = g
            if st:
                ref.state[q] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()}
        ref.step()
        for q, (t0, g, _), t1 in zip(ps, pre, post):
            if g is None:
                assert torch.equal(t0, t1)
                continue
            d_ref, d_hip = (q.detach() - t0).double(), (t1 - t0).double()
            assert float(d_ref.norm()) > 0
            e = float((d_hip - d_ref).norm() / d_ref.norm())
            if e > worst:
                worst, where = e, (ci, name, tuple(t0.shape), float(g.abs().max()), float(d_ref.norm()))
    # 1e-3 of the update, not tighter: g + weight_decay * p is one fused multiply-add in torch's vectorised CPU kernel and a multiply
    # and an add here, and where a gradient element nearly cancels weight_decay * p (|g'| ~ eps) the ulp of difference moves that
    # element's m / (sqrt(v) + eps) by up to ~1e-3 (observed 7e-5 of a 16-element tensor's update).  A missing step, a wrong learning
    # rate, beta, weight decay or bias correction is off by per cents; elementwise agreement on generic data is tests/test_adam_gpu.py's 1e-6.
    assert worst <= 1e-3, (worst, where)
    return worst
