#!/usr/bin/env python3
"""Generate golden fixtures from the REAL reference classes (CPU, fp32).

Run in the build container only (needs /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

Import recipe: stub cv2 (util.py:3 imports it, the hot path never
calls it), put /root/reference/src on sys.path, import util BEFORE generator
(import cycle util.py:13 <-> generator.py:8).  trainer.py is not importable
(evan/skvideo/colorlog/tensorboardX are absent), so the iteration of
trainer.py:279-363 is driven here directly on the reference's module, loss and
torch.optim.Adam objects (train.py:171-176).

Fixtures are DATA only: seeds, initial state_dicts, inputs, expected outputs,
gradients and checksums.  No reference source is stored.

Random draws are not stored (they are megabytes); a fixture records the seed
and the oracle re-draws from the same CPU generator in the same order.  That
the order is right is exactly what tests/test_oracle_golden.py proves.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def import_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import util  # noqa: F401  (must come first)
    import generator, discriminator, loss  # noqa: E401
    return util, generator, discriminator, loss


def sub(t: torch.Tensor, step: int = 37) -> np.ndarray:
    """Strided sample of a big activation (keeps fixtures small)."""
    return t.detach().contiguous().view(-1)[::step].numpy().copy()


def summ(t: torch.Tensor) -> np.ndarray:
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def put_state(out, prefix, sd):
    for k, v in sd.items():
        out[f"{prefix}/{k}"] = v.detach().clone().numpy()


def build_models(ref, cfg, seed):
    util, generator, discriminator, loss = ref
    torch.manual_seed(seed)
    ggen = generator.GeometricVideoGenerator(cfg["dzc"], cfg["dzm"], cfg["Cg"], cfg["geo"], cfg["ngf_g"], 16)
    cgen = generator.ColorVideoGenerator(ggen.channel, cfg["dzcol"], cfg["geo"], cfg["ngf_c"], 16)
    idis = discriminator.ImageDiscriminator(ggen.channel, cgen.channel, cfg["noise_i"][0], cfg["noise_i"][1], cfg["ndf_i"])
    vdis = discriminator.VideoDiscriminator(ggen.channel, cgen.channel, cfg["noise_v"][0], cfg["noise_v"][1], cfg["ndf_v"])
    gdis = discriminator.GradientDiscriminator(ggen.channel, cgen.channel, cfg["noise_g"][0], cfg["noise_g"][1], cfg["ndf_g"])
    models = dict(ggen=ggen, cgen=cgen, idis=idis, vdis=vdis, gdis=gdis)
    for m in models.values():
        m.apply(util.init_weights)  # train.py:164-165
    return models


def cfg_meta(out, cfg):
    for k, v in cfg.items():
        if isinstance(v, tuple):
            out[f"cfg/{k}"] = np.array([float(v[0]), float(v[1])])
        elif isinstance(v, str):
            out[f"cfg/{k}"] = np.array(v)
        else:
            out[f"cfg/{k}"] = np.array(v)


# --------------------------------------------------------------------------- #
# module-level fixture: forward outputs + all parameter gradients
# --------------------------------------------------------------------------- #
def module_fixture(ref, cfg, name, B=2, seed=1234):
    out = {}
    cfg_meta(out, cfg)
    out["meta/B"] = np.array(B)
    out["meta/seed_init"] = np.array(seed)
    models = build_models(ref, cfg, seed)
    for n, m in models.items():
        put_state(out, f"init/{n}", m.state_dict())
    ggen, cgen, idis, vdis, gdis = (models[k] for k in ("ggen", "cgen", "idis", "vdis", "gdis"))

    # ---- generators, train mode: seed -> sample -> colourise ----
    s_fwd = seed + 1
    out["meta/seed_gen_train"] = np.array(s_fwd)
    torch.manual_seed(s_fwd)
    ggen.train(); cgen.train()
    xg = ggen.sample_videos(B)
    xc = cgen.forward_videos(xg)
    out["gen_train/xg_stride"] = np.array(xg.stride())
    out["gen_train/xc_stride"] = np.array(xc.stride())
    out["gen_train/xg_sub"] = sub(xg); out["gen_train/xg_sum"] = summ(xg)
    out["gen_train/xc_sub"] = sub(xc); out["gen_train/xc_sum"] = summ(xc)
    # fixed cotangents (deterministic, seed-free)
    n_g, n_c = xg.numel(), xc.numel()
    cot_g = torch.cos(torch.arange(n_g, dtype=torch.float32) * 0.37).view(xg.shape)
    cot_c = torch.sin(torch.arange(n_c, dtype=torch.float32) * 0.11).view(xc.shape)
    ((xg * cot_g).sum() + (xc * cot_c).sum()).backward()
    for n in ("ggen", "cgen"):
        for k, p in models[n].named_parameters():
            out[f"gen_train/grad/{n}/{k}"] = p.grad.clone().numpy()
        put_state(out, f"gen_train/after/{n}", {k: v for k, v in models[n].state_dict().items() if "running" in k or "num_batches" in k})
        models[n].zero_grad()

    # ---- generators, eval mode (running stats, no dropout) ----
    s_eval = seed + 2
    out["meta/seed_gen_eval"] = np.array(s_eval)
    torch.manual_seed(s_eval)
    ggen.eval(); cgen.eval()
    with torch.no_grad():
        xg_e = ggen.sample_videos(B)
        xc_e = cgen.forward_videos(xg_e)
    out["gen_eval/xg_sub"] = sub(xg_e); out["gen_eval/xg_sum"] = summ(xg_e)
    out["gen_eval/xc_sub"] = sub(xc_e); out["gen_eval/xc_sum"] = summ(xc_e)
    ggen.train(); cgen.train()

    # ---- discriminators on fixed inputs (non-contiguous, like the trainer feeds) ----
    s_in = seed + 3
    g = torch.Generator().manual_seed(s_in)
    out["meta/seed_dis_inputs"] = np.array(s_in)
    # memory order (B,T,C,H,W) viewed as (B,C,T,H,W): the generators' output layout
    xg_in = (torch.rand(B, 16, cfg["Cg"], 64, 64, generator=g) * 2 - 1).permute(0, 2, 1, 3, 4).requires_grad_(True)
    xc_in = (torch.rand(B, 16, 3, 64, 64, generator=g) * 2 - 1).permute(0, 2, 1, 3, 4).requires_grad_(True)
    s_d = seed + 4
    out["meta/seed_dis_fwd"] = np.array(s_d)
    torch.manual_seed(s_d)
    t = 5
    out["meta/t_rand"] = np.array(t)
    for m in (idis, vdis, gdis):
        m.train()
    yi = idis(xg_in[:, :, t], xc_in[:, :, t])
    yv = vdis(xg_in, xc_in)
    yg = gdis(xg_in, xc_in)
    out["dis/yi"] = yi.detach().numpy().copy()
    out["dis/yv"] = yv.detach().numpy().copy()
    out["dis/yg"] = yg.detach().numpy().copy()
    tot = (yi * torch.linspace(-1, 1, yi.numel()).view(yi.shape)).sum() \
        + (yv * torch.linspace(1, -1, yv.numel()).view(yv.shape)).sum() \
        + (yg * torch.linspace(-0.5, 1.5, yg.numel()).view(yg.shape)).sum()
    gin = torch.autograd.grad(tot, [xg_in, xc_in], retain_graph=True)
    out["dis/grad_xg_sub"] = sub(gin[0], 11); out["dis/grad_xg_sum"] = summ(gin[0])
    out["dis/grad_xc_sub"] = sub(gin[1], 11); out["dis/grad_xc_sum"] = summ(gin[1])
    tot.backward()
    for n in ("idis", "vdis", "gdis"):
        for k, p in models[n].named_parameters():
            out[f"dis/grad/{n}/{k}"] = p.grad.clone().numpy()
        put_state(out, f"dis/after/{n}", {k: v for k, v in models[n].state_dict().items() if "running" in k or "num_batches" in k})

    # ---- losses on the three real logit shapes ----
    _, _, _, loss = ref
    gl = torch.Generator().manual_seed(seed + 5)
    ys = [torch.randn(s, generator=gl) * 2 for s in ((B, 4, 4), (B, 4, 4, 4), (B, 3, 4, 4))]
    for lname, L in (("adv", loss.AdversarialLoss()), ("hinge", loss.HingeLoss())):
        for i, y in enumerate(ys):
            yr = y.clone().requires_grad_(True)
            yf = (y.flip(0) * 0.7 + 0.1).clone().requires_grad_(True)
            v = L.compute_dis_loss(yr, yf)
            gr, gf = torch.autograd.grad(v, [yr, yf])
            out[f"loss/{lname}/dis{i}/yr"] = yr.detach().numpy().copy()
            out[f"loss/{lname}/dis{i}/yf"] = yf.detach().numpy().copy()
            out[f"loss/{lname}/dis{i}/value"] = np.array(v.item())
            out[f"loss/{lname}/dis{i}/gr"] = gr.numpy().copy()
            out[f"loss/{lname}/dis{i}/gf"] = gf.numpy().copy()
        yq = [y.clone().requires_grad_(True) for y in ys]
        v = L.compute_gen_loss(*yq)
        gq = torch.autograd.grad(v, yq, allow_unused=True)
        out[f"loss/{lname}/gen/value"] = np.array(v.item())
        for i, gg in enumerate(gq):
            out[f"loss/{lname}/gen/g{i}"] = (gg if gg is not None else torch.zeros_like(ys[i])).numpy().copy()
    np.savez(os.path.join(HERE, name), **out)
    print("wrote", name, sum(v.nbytes for v in out.values()) / 1e6, "MB")


# --------------------------------------------------------------------------- #
# step-level fixture: trainer.py:279-363 driven on the reference objects
# --------------------------------------------------------------------------- #
def step_fixture(ref, cfg, name, loss_name, num_gen_update, start_in_eval, B=2, iters=3, seed=77):
    util, generator, discriminator, loss = ref
    out = {}
    cfg_meta(out, cfg)
    out["meta/B"] = np.array(B); out["meta/iters"] = np.array(iters)
    out["meta/loss"] = np.array(loss_name); out["meta/num_gen_update"] = np.array(num_gen_update)
    out["meta/num_dis_update"] = np.array(1)
    out["meta/start_in_eval"] = np.array(int(start_in_eval))
    out["meta/seed_init"] = np.array(seed)
    models = build_models(ref, cfg, seed)
    for n, m in models.items():
        put_state(out, f"init/{n}", m.state_dict())
    ggen, cgen, idis, vdis, gdis = (models[k] for k in ("ggen", "cgen", "idis", "vdis", "gdis"))
    L = loss.AdversarialLoss() if loss_name == "adversarial-loss" else loss.HingeLoss()
    lrs = dict(ggen=2e-4, cgen=2e-4, idis=5e-4, vdis=5e-4, gdis=2e-4)
    opt = {n: torch.optim.Adam(m.parameters(), lr=lrs[n], betas=(0.5, 0.999), weight_decay=1e-5) for n, m in models.items()}
    for n, v in lrs.items():
        out[f"meta/lr/{n}"] = np.array(v)
    gdata = torch.Generator().manual_seed(seed + 1)
    lo, hi = (-0.5, 0.5) if cfg["Cg"] == 2 else (-1.0, 1.0)
    xc_real = torch.rand(B, 3, 16, 64, 64, generator=gdata) * 2 - 1
    xg_real = torch.rand(B, cfg["Cg"], 16, 64, 64, generator=gdata) * (hi - lo) + lo
    out["meta/seed_data"] = np.array(seed + 1)
    t_rands = [3, 11, 0, 15, 7][:iters]
    out["meta/t_rands"] = np.array(t_rands)
    out["meta/seed_run"] = np.array(seed + 2)
    torch.manual_seed(seed + 2)
    if start_in_eval:  # trainer.py:266-267 leaves the generators in eval()
        ggen.eval(); cgen.eval()
    losses = []
    for it in range(1, iters + 1):
        t = t_rands[it - 1]
        idis.train(); vdis.train(); gdis.train()
        idis.zero_grad(); vdis.zero_grad(); gdis.zero_grad()
        y_real_i = idis(xg_real[:, :, t], xc_real[:, :, t])
        y_real_v = vdis(xg_real, xc_real)
        y_real_g = gdis(xg_real, xc_real)
        xg_fake = ggen.sample_videos(B)
        xc_fake = cgen.forward_videos(xg_fake)
        y_fake_i = idis(xg_fake[:, :, t], xc_fake[:, :, t])
        y_fake_v = vdis(xg_fake, xc_fake)
        y_fake_g = gdis(xg_fake, xc_fake)
        loss_idis = L.compute_dis_loss(y_real_i, y_fake_i)
        loss_vdis = L.compute_dis_loss(y_real_v, y_fake_v)
        loss_gdis = L.compute_dis_loss(y_real_g, y_fake_g)
        loss_dis = loss_idis + loss_vdis + loss_gdis
        if it % num_gen_update == 0:
            loss_dis.backward()
            opt["idis"].step(); opt["vdis"].step(); opt["gdis"].step()
        ggen.train(); cgen.train()
        ggen.zero_grad(); cgen.zero_grad()
        xg_fake = ggen.sample_videos(B)
        xc_fake = cgen.forward_videos(xg_fake)
        y_fake_i = idis(xg_fake[:, :, t], xc_fake[:, :, t])
        y_fake_v = vdis(xg_fake, xc_fake)
        y_fake_g = gdis(xg_fake, xc_fake)
        loss_gen = L.compute_gen_loss(y_fake_i, y_fake_v, y_fake_g)
        loss_gen.backward()
        opt["ggen"].step(); opt["cgen"].step(); opt["ggen"].step()
        losses.append([loss_idis.item(), loss_vdis.item(), loss_gdis.item(), loss_gen.item()])
        for n, m in models.items():
            for k, v in m.state_dict().items():
                v = v.detach().float().reshape(-1)
                out[f"after{it}/{n}/{k}"] = np.concatenate([
                    np.array([v.double().abs().sum().item(), v.double().sum().item()]),
                    v[:8].double().numpy()])
    out["losses"] = np.array(losses)
    np.savez(os.path.join(HERE, name), **out)
    print("wrote", name, sum(v.nbytes for v in out.values()) / 1e6, "MB", losses)


# --------------------------------------------------------------------------- #
# full-width scalars (real channel counts, values only)
# --------------------------------------------------------------------------- #
def gsub(t: torch.Tensor) -> np.ndarray:
    """<= ~256 strided elements of a gradient tensor: direction-sensitive, unlike a norm."""
    v = t.detach().reshape(-1)
    return v[::max(1, v.numel() // 256)].numpy().copy()


def fullwidth_fixture(ref, cfg, name, loss_name="adversarial-loss", B=2, seed=99):
    util, generator, discriminator, loss = ref
    out = {}
    cfg_meta(out, cfg)
    out["meta/B"] = np.array(B); out["meta/seed_init"] = np.array(seed); out["meta/loss"] = np.array(loss_name)
    # states are re-derivable: same constructor order under the same seed
    models = build_models(ref, cfg, seed)
    for n, m in models.items():
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                out[f"init_sum/{n}/{k}"] = summ(v)
    ggen, cgen, idis, vdis, gdis = (models[k] for k in ("ggen", "cgen", "idis", "vdis", "gdis"))
    torch.manual_seed(seed + 1)
    out["meta/seed_run"] = np.array(seed + 1)
    xg = ggen.sample_videos(B); xc = cgen.forward_videos(xg)
    t = 9
    out["meta/t_rand"] = np.array(t)
    yi = idis(xg[:, :, t], xc[:, :, t]); yv = vdis(xg, xc); yg = gdis(xg, xc)
    L = loss.AdversarialLoss() if loss_name == "adversarial-loss" else loss.HingeLoss()
    v = L.compute_gen_loss(yi, yv, yg)
    v.backward()
    out["loss_gen"] = np.array(v.item())
    out["xg_sum"] = summ(xg); out["xc_sum"] = summ(xc)
    out["xg_stride"] = np.array(xg.stride()); out["xc_stride"] = np.array(xc.stride())
    out["yi"] = yi.detach().numpy().copy(); out["yv"] = yv.detach().numpy().copy(); out["yg"] = yg.detach().numpy().copy()
    for n, m in models.items():
        for k, p in m.named_parameters():
            if p.grad is None:     # hinge: y_fake_g is unused (loss.py:190-191), gdis gets no gradient
                out[f"gradnone/{n}/{k}"] = np.array(1)
                continue
            out[f"gradnorm/{n}/{k}"] = np.array(p.grad.double().norm().item())
            out[f"gradsub/{n}/{k}"] = gsub(p.grad)
    np.savez(os.path.join(HERE, name), **out)
    print("wrote", name, sum(v.nbytes for v in out.values()) / 1e6, "MB")
    return models


def step_fullwidth_fixture(ref, cfg, name, loss_name, num_gen_update, lrs, B=2, iters=2, seed=31):
    """trainer.py:279-363 at the REAL channel widths (initial states re-derivable from the seed, so only
    scalars and strided samples are stored): losses per iteration, and per tensor the Adam UPDATE
    theta_after - theta_before of every iteration as {L2 norm, strided sample}."""
    util, generator, discriminator, loss = ref
    out = {}
    cfg_meta(out, cfg)
    out["meta/B"] = np.array(B); out["meta/iters"] = np.array(iters)
    out["meta/loss"] = np.array(loss_name); out["meta/num_gen_update"] = np.array(num_gen_update)
    out["meta/num_dis_update"] = np.array(1); out["meta/start_in_eval"] = np.array(0)
    out["meta/seed_init"] = np.array(seed)
    models = build_models(ref, cfg, seed)
    for n, m in models.items():
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                out[f"init_sum/{n}/{k}"] = summ(v)
    ggen, cgen, idis, vdis, gdis = (models[k] for k in ("ggen", "cgen", "idis", "vdis", "gdis"))
    L = loss.AdversarialLoss() if loss_name == "adversarial-loss" else loss.HingeLoss()
    opt = {n: torch.optim.Adam(m.parameters(), lr=lrs[n], betas=(0.5, 0.999), weight_decay=1e-5) for n, m in models.items()}
    for n, v in lrs.items():
        out[f"meta/lr/{n}"] = np.array(v)
    gdata = torch.Generator().manual_seed(seed + 1)
    lo, hi = (-0.5, 0.5) if cfg["Cg"] == 2 else (-1.0, 1.0)
    xc_real = torch.rand(B, 3, 16, 64, 64, generator=gdata) * 2 - 1
    xg_real = torch.rand(B, cfg["Cg"], 16, 64, 64, generator=gdata) * (hi - lo) + lo
    out["meta/seed_data"] = np.array(seed + 1)
    t_rands = [3, 11, 0, 15, 7][:iters]
    out["meta/t_rands"] = np.array(t_rands)
    out["meta/seed_run"] = np.array(seed + 2)
    torch.manual_seed(seed + 2)
    losses = []
    for it in range(1, iters + 1):
        before = {n: {k: p.detach().clone() for k, p in m.named_parameters()} for n, m in models.items()}
        t = t_rands[it - 1]
        idis.train(); vdis.train(); gdis.train()
        idis.zero_grad(); vdis.zero_grad(); gdis.zero_grad()
        y_real_i = idis(xg_real[:, :, t], xc_real[:, :, t])
        y_real_v = vdis(xg_real, xc_real)
        y_real_g = gdis(xg_real, xc_real)
        xg_fake = ggen.sample_videos(B)
        xc_fake = cgen.forward_videos(xg_fake)
        y_fake_i = idis(xg_fake[:, :, t], xc_fake[:, :, t])
        y_fake_v = vdis(xg_fake, xc_fake)
        y_fake_g = gdis(xg_fake, xc_fake)
        loss_idis = L.compute_dis_loss(y_real_i, y_fake_i)
        loss_vdis = L.compute_dis_loss(y_real_v, y_fake_v)
        loss_gdis = L.compute_dis_loss(y_real_g, y_fake_g)
        loss_dis = loss_idis + loss_vdis + loss_gdis
        if it % num_gen_update == 0:
            loss_dis.backward()
            opt["idis"].step(); opt["vdis"].step(); opt["gdis"].step()
        ggen.train(); cgen.train()
        ggen.zero_grad(); cgen.zero_grad()
        xg_fake = ggen.sample_videos(B)
        xc_fake = cgen.forward_videos(xg_fake)
        y_fake_i = idis(xg_fake[:, :, t], xc_fake[:, :, t])
        y_fake_v = vdis(xg_fake, xc_fake)
        y_fake_g = gdis(xg_fake, xc_fake)
        loss_gen = L.compute_gen_loss(y_fake_i, y_fake_v, y_fake_g)
        loss_gen.backward()
        opt["ggen"].step(); opt["cgen"].step(); opt["ggen"].step()
        losses.append([loss_idis.item(), loss_vdis.item(), loss_gdis.item(), loss_gen.item()])
        for n, m in models.items():
            for k, p in m.named_parameters():
                d = p.detach() - before[n][k]
                out[f"delta{it}/{n}/{k}/norm"] = np.array(d.double().norm().item())
                out[f"delta{it}/{n}/{k}/sub"] = gsub(d)
            for k, v in m.state_dict().items():
                if "running" in k:
                    out[f"after{it}/{n}/{k}"] = summ(v)
    out["losses"] = np.array(losses)
    np.savez(os.path.join(HERE, name), **out)
    print("wrote", name, sum(v.nbytes for v in out.values()) / 1e6, "MB", losses)


def stress_d_fixture(ref, name="stress_d_32x128x128.npz", seed=555):
    """The 32 x 128 x 128 discriminator stress shape (BASELINE configs[4]): vdis / gdis are
    size-agnostic (discriminator.py:181-206,288-305), so the reference classes run it directly.  B = 1 (so
    `.squeeze()` also drops the batch dimension), flow channels, isogd-flow's noise settings."""
    util, generator, discriminator, loss = ref
    out = {}
    torch.manual_seed(seed)
    vdis = discriminator.VideoDiscriminator(2, 3, True, 0.2, 64)
    gdis = discriminator.GradientDiscriminator(2, 3, False, 0.2, 32)
    for m in (vdis, gdis):
        m.apply(util.init_weights)
        m.train()
    out["meta/seed_init"] = np.array(seed)
    for n, m in (("vdis", vdis), ("gdis", gdis)):
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                out[f"init_sum/{n}/{k}"] = summ(v)
    g = torch.Generator().manual_seed(seed + 1)
    out["meta/seed_inputs"] = np.array(seed + 1)
    xg = (torch.rand(1, 32, 2, 128, 128, generator=g) - 0.5).permute(0, 2, 1, 3, 4).requires_grad_(True)   # the generators' memory order
    xc = (torch.rand(1, 32, 3, 128, 128, generator=g) * 2 - 1).permute(0, 2, 1, 3, 4).requires_grad_(True)
    torch.manual_seed(seed + 2)
    out["meta/seed_fwd"] = np.array(seed + 2)
    yv = vdis(xg, xc); yg = gdis(xg, xc)
    out["yv"] = yv.detach().numpy().copy(); out["yg"] = yg.detach().numpy().copy()
    tot = (yv * torch.linspace(1, -1, yv.numel()).view(yv.shape)).sum() + (yg * torch.linspace(-0.5, 1.5, yg.numel()).view(yg.shape)).sum()
    tot.backward()
    out["grad_xg_sub"] = sub(xg.grad, 101); out["grad_xg_sum"] = summ(xg.grad)
    out["grad_xc_sub"] = sub(xc.grad, 101); out["grad_xc_sum"] = summ(xc.grad)
    for n, m in (("vdis", vdis), ("gdis", gdis)):
        for k, p in m.named_parameters():
            out[f"gradnorm/{n}/{k}"] = np.array(p.grad.double().norm().item())
            out[f"gradsub/{n}/{k}"] = gsub(p.grad)
        for k, v in m.state_dict().items():
            if "running" in k:
                out[f"after/{n}/{k}"] = v.detach().numpy().copy()
    np.savez(os.path.join(HERE, name), **out)
    print("wrote", name, sum(v.nbytes for v in out.values()) / 1e6, "MB", tuple(yv.shape), tuple(yg.shape))


def sampling_fixture(ref, cfg, name, seed=4321):
    """util.generate_samples on the reference generators (depth): uint8 outputs, sub-sampled."""
    util, generator, discriminator, loss = ref
    out = {}
    cfg_meta(out, cfg)
    models = build_models(ref, cfg, seed)
    for n in ("ggen", "cgen"):
        put_state(out, f"init/{n}", models[n].state_dict())
    out["meta/seed_run"] = np.array(seed + 1)
    torch.manual_seed(seed + 1)
    xg, xc = util.generate_samples(models["ggen"], models["cgen"], 3, 2)   # 2 batches of 2, truncated to 3
    assert xg.dtype == np.uint8 and xg.shape == (3, 3, 16, 64, 64) and xc.shape == (3, 3, 16, 64, 64)
    out["xg_sub"] = xg.reshape(-1)[::13].copy(); out["xc_sub"] = xc.reshape(-1)[::13].copy()
    out["xg_sum"] = np.array(int(xg.astype(np.int64).sum())); out["xc_sum"] = np.array(int(xc.astype(np.int64).sum()))
    g = torch.Generator().manual_seed(seed + 2)
    v = torch.randn(2, 3, 4, 8, 8, generator=g) * 0.8
    out["conv_in"] = v.numpy().copy(); out["conv_out"] = util.videos_to_numpy(v)
    im = torch.randn(2, 3, 8, 8, generator=g) * 0.8
    out["img_in"] = im.numpy().copy(); out["img_out"] = util.images_to_numpy(im)
    np.savez(os.path.join(HERE, name), **out)
    print("wrote", name, sum(v.nbytes for v in out.values()) / 1e6, "MB")


def interchange_fixture(ref, name="interchange.json"):
    """checkpoints written by this repo load into the reference's classes and vice versa
    (trainer.py:78-86 save_params <-> infer.py:31-36 load_state_dict), incl. the whole-module pickle of
    trainer.py:70-76 resolved through compat/ (top-level module names)."""
    import io, json, subprocess, tempfile
    util, generator, discriminator, loss = ref
    sys.path.insert(0, "/root/repo")
    from dcvgan_amd import discriminator as D2, generator as G2
    pairs = [
        ("ggen", lambda m: m.GeometricVideoGenerator(40, 10, 1, "depth", 16, 16), generator, G2),
        ("cgen", lambda m: m.ColorVideoGenerator(1, 10, "depth", 16, 16), generator, G2),
        ("idis", lambda m: m.ImageDiscriminator(1, 3, True, 0.1, 16), discriminator, D2),
        ("vdis", lambda m: m.VideoDiscriminator(1, 3, True, 0.1, 16), discriminator, D2),
        ("gdis", lambda m: m.GradientDiscriminator(1, 3, False, 0.2, 16), discriminator, D2),
    ]
    res = {}
    for nm, make, refmod, mymod in pairs:
        torch.manual_seed(1); r = make(refmod)
        torch.manual_seed(2); mine = make(mymod)
        buf = io.BytesIO(); torch.save(mine.state_dict(), buf); buf.seek(0)      # ours -> reference
        k1 = r.load_state_dict(torch.load(buf), strict=True)
        same1 = all(torch.equal(a, b) for a, b in zip(r.state_dict().values(), mine.state_dict().values()))
        torch.manual_seed(3); r2 = make(refmod)
        buf = io.BytesIO(); torch.save(r2.state_dict(), buf); buf.seek(0)        # reference -> ours
        k2 = mine.load_state_dict(torch.load(buf), strict=True)
        same2 = all(torch.equal(a, b) for a, b in zip(r2.state_dict().values(), mine.state_dict().values()))
        res[nm] = dict(ours_into_reference=bool(same1 and not k1.missing_keys and not k1.unexpected_keys),
                       reference_into_ours=bool(same2 and not k2.missing_keys and not k2.unexpected_keys),
                       keys=list(mine.state_dict().keys()) == list(r.state_dict().keys()))
    # whole-module pickle written by the REFERENCE (trainer.py:75-76) and opened in a process that only has compat/
    with tempfile.TemporaryDirectory() as td:
        torch.manual_seed(4)
        g = generator.GeometricVideoGenerator(40, 10, 1, "depth", 16, 16)
        torch.save(g, os.path.join(td, "ggen_model.pth"))
        code = ("import sys; sys.path[:0] = ['/root/repo/compat', '/root/repo']; import torch; "
                f"m = torch.load(r'{td}/ggen_model.pth', weights_only=False); "
                "print(type(m).__module__, type(m).__name__, len(m.state_dict()))")
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
        res["reference_pickle_opens_through_compat"] = out.stdout.strip() or out.stderr.strip()[-300:]
    json.dump(res, open(os.path.join(HERE, name), "w"), indent=1)
    print("wrote", name, res)


def segmentation_fixture(ref, name="segmentation_io.npz", seed=2024):
    """SURVEY §8(f).4 data paths around the segmentation branch, from the reference's own functions:
    the SURREAL part palette (util.py:325-372), argmax colouring (util.py:236-246), the dataset's
    one-hot decode (dataset.py:176-181, restated through numpy exactly as written there: np.eye(25)[labels])."""
    util = ref[0]
    out = {}
    out["palette_u8"] = np.stack([(util.segm_color(i) * 255).astype(np.uint8) for i in range(25)])
    g = np.random.default_rng(seed)
    probs = g.random((2, 25, 3, 8, 8)).astype(np.float32)
    probs[0, 3, 0, 0, 0] = probs[0, 7, 0, 0, 0] = 2.0      # a tie: the first maximum wins
    out["probs"] = probs
    out["color"] = util.geometric_info_in_color_format(probs.copy(), "segmentation")
    labels = g.integers(0, 25, size=(4, 8, 8)).astype(np.uint8)   # frames of segm.npy
    out["labels"] = labels
    out["onehot"] = np.eye(25, dtype=np.float32)[labels].transpose(3, 0, 1, 2)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name)


if __name__ == "__main__":
    ref = import_reference()
    torch.set_num_threads(8)
    small_depth = dict(geo="depth", Cg=1, dzc=5, dzm=3, dzcol=3, ngf_g=6, ngf_c=6, ndf_i=6, ndf_v=6, ndf_g=4,
                       noise_i=(True, 0.1), noise_v=(True, 0.1), noise_g=(False, 0.2))
    small_flow = dict(geo="optical-flow", Cg=2, dzc=4, dzm=2, dzcol=2, ngf_g=4, ngf_c=4, ndf_i=4, ndf_v=4, ndf_g=4,
                      noise_i=(True, 0.2), noise_v=(True, 0.2), noise_g=(True, 0.2))
    module_fixture(ref, small_depth, "modules_depth_w6.npz")
    module_fixture(ref, small_flow, "modules_flow_w4.npz")
    step_fixture(ref, small_depth, "step_depth_adv_g1.npz", "adversarial-loss", 1, False)
    step_fixture(ref, small_depth, "step_depth_adv_g1_evalstart.npz", "adversarial-loss", 1, True)
    step_fixture(ref, small_flow, "step_flow_hinge_g2.npz", "hinge-loss", 2, False)
    full = dict(geo="depth", Cg=1, dzc=40, dzm=10, dzcol=10, ngf_g=64, ngf_c=64, ndf_i=64, ndf_v=64, ndf_g=32,
                noise_i=(True, 0.1), noise_v=(True, 0.1), noise_g=(False, 0.2))
    fullwidth_fixture(ref, full, "fullwidth_isogd_depth.npz")
    # config/surreal-depth1.yml:5,27,30,47-76 (ggen ngf 96, no Noise, hinge; the yml has no gdis block: isogd's ndf 32 is injected)
    full_surreal = dict(full, ngf_g=96, noise_i=(False, 0.2), noise_v=(False, 0.2), noise_g=(False, 0.2))
    # config/isogd-flow.yml:5,9-10,16-18,27 (two flow channels, noise sigma 0.2, hinge)
    full_flow = dict(full, geo="optical-flow", Cg=2, noise_i=(True, 0.2), noise_v=(True, 0.2), noise_g=(False, 0.2))
    fullwidth_fixture(ref, full_surreal, "fullwidth_surreal_depth1.npz", "hinge-loss", seed=199)
    fullwidth_fixture(ref, full_flow, "fullwidth_isogd_flow.npz", "hinge-loss", seed=299)
    lr2 = dict(ggen=2e-4, cgen=2e-4, idis=2e-4, vdis=2e-4, gdis=2e-4)
    step_fullwidth_fixture(ref, full_surreal, "step_fullwidth_surreal_depth1.npz", "hinge-loss", 2, lr2)
    step_fullwidth_fixture(ref, full_flow, "step_fullwidth_isogd_flow.npz", "hinge-loss", 1, lr2, seed=41)
    # the headline config's own composed iteration: config/isogd-depth.yml:5,27,47-89 — BCE-with-logits, Noise sigma 0.1 on idis / vdis,
    # lr 5e-4 for idis / vdis (:70,79), 2e-4 for the others
    lr_isogd = dict(ggen=2e-4, cgen=2e-4, idis=5e-4, vdis=5e-4, gdis=2e-4)
    step_fullwidth_fixture(ref, full, "step_fullwidth_isogd_depth.npz", "adversarial-loss", 1, lr_isogd, seed=51)
    stress_d_fixture(ref)
    sampling_fixture(ref, dict(small_depth, ngf_g=4, ngf_c=4), "sampling_depth_w4.npz")
    interchange_fixture(ref)
    small_segm = dict(geo="segmentation", Cg=25, dzc=4, dzm=2, dzcol=2, ngf_g=4, ngf_c=4, ndf_i=4, ndf_v=4, ndf_g=4,
                      noise_i=(True, 0.2), noise_v=(True, 0.2), noise_g=(True, 0.2))
    module_fixture(ref, small_segm, "modules_segm_w4.npz")
    segmentation_fixture(ref)
