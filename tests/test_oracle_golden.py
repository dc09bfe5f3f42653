"""CPU: the oracle (oracle/dcvgan_oracle.py) against fixtures produced by the real
reference classes (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import dcvgan_oracle as O
from tests import goldenio as G

TOL = 2e-6  # same torch, same CPU kernels: expected bit-equal; allow thread-order noise


@pytest.mark.parametrize("fixture", ["modules_depth_w6.npz", "modules_flow_w4.npz", "modules_segm_w4.npz"])
def test_generators_train_forward_backward(fixture):
    fx = G.load(fixture); cfg = G.cfg_of(fx); st = G.states(fx)
    B = cfg.batchsize
    for m in ("ggen", "cgen"):
        O.require_grad(st[m])
    torch.manual_seed(int(fx["meta/seed_gen_train"]))
    rng = O.TorchRng()
    xg = O.ggen_sample_videos(st["ggen"], B, 16, cfg.dim_z_content, cfg.dim_z_motion, cfg.channel, rng, True, segmentation=cfg.geometric_info == 'segmentation')
    xc = O.cgen_forward_videos(st["cgen"], xg, cfg.dim_z_color, rng, True, segmentation=cfg.geometric_info == 'segmentation')
    assert tuple(xg.stride()) == tuple(fx["gen_train/xg_stride"])
    assert tuple(xc.stride()) == tuple(fx["gen_train/xc_stride"])
    assert G.relerr(G.sub(xg), fx["gen_train/xg_sub"]) < TOL
    assert G.relerr(G.sub(xc), fx["gen_train/xc_sub"]) < TOL
    cot_g = torch.cos(torch.arange(xg.numel(), dtype=torch.float32) * 0.37).view(xg.shape)
    cot_c = torch.sin(torch.arange(xc.numel(), dtype=torch.float32) * 0.11).view(xc.shape)
    ((xg * cot_g).sum() + (xc * cot_c).sum()).backward()
    for m in ("ggen", "cgen"):
        for k, p in st[m].items():
            key = f"gen_train/grad/{m}/{k}"
            if key in fx:
                assert G.relerr(p.grad.numpy(), fx[key]) < 1e-5, key
        for k in st[m]:
            key = f"gen_train/after/{m}/{k}"
            if key in fx:
                assert np.allclose(st[m][k].detach().numpy(), fx[key], rtol=1e-6, atol=1e-7), key


@pytest.mark.parametrize("fixture", ["modules_depth_w6.npz", "modules_flow_w4.npz", "modules_segm_w4.npz"])
def test_generators_eval(fixture):
    fx = G.load(fixture); cfg = G.cfg_of(fx)
    # eval pass ran AFTER one training forward: load the post-train running stats
    st = G.states(fx)
    for m in ("ggen", "cgen"):
        for k in list(st[m]):
            key = f"gen_train/after/{m}/{k}"
            if key in fx:
                st[m][k] = torch.from_numpy(np.array(fx[key]))
    torch.manual_seed(int(fx["meta/seed_gen_eval"]))
    rng = O.TorchRng()
    with torch.no_grad():
        xg = O.ggen_sample_videos(st["ggen"], cfg.batchsize, 16, cfg.dim_z_content, cfg.dim_z_motion, cfg.channel, rng, False, segmentation=cfg.geometric_info == 'segmentation')
        xc = O.cgen_forward_videos(st["cgen"], xg, cfg.dim_z_color, rng, False, segmentation=cfg.geometric_info == 'segmentation')
    assert G.relerr(G.sub(xg), fx["gen_eval/xg_sub"]) < TOL
    assert G.relerr(G.sub(xc), fx["gen_eval/xc_sub"]) < TOL


def dis_inputs(fx, cfg):
    g = torch.Generator().manual_seed(int(fx["meta/seed_dis_inputs"]))
    B = cfg.batchsize
    xg = (torch.rand(B, 16, cfg.channel, 64, 64, generator=g) * 2 - 1).permute(0, 2, 1, 3, 4).requires_grad_(True)
    xc = (torch.rand(B, 16, 3, 64, 64, generator=g) * 2 - 1).permute(0, 2, 1, 3, 4).requires_grad_(True)
    return xg, xc


@pytest.mark.parametrize("fixture", ["modules_depth_w6.npz", "modules_flow_w4.npz", "modules_segm_w4.npz"])
def test_discriminators_forward_backward(fixture):
    fx = G.load(fixture); cfg = G.cfg_of(fx); st = G.states(fx)
    for m in ("idis", "vdis", "gdis"):
        O.require_grad(st[m])
    xg, xc = dis_inputs(fx, cfg)
    torch.manual_seed(int(fx["meta/seed_dis_fwd"]))
    rng = O.TorchRng(); t = int(fx["meta/t_rand"])
    yi = O.idis_forward(st["idis"], xg[:, :, t], xc[:, :, t], cfg.use_noise["idis"], cfg.noise_sigma["idis"], rng, True)
    yv = O.vdis_forward(st["vdis"], xg, xc, cfg.use_noise["vdis"], cfg.noise_sigma["vdis"], rng, True)
    yg = O.gdis_forward(st["gdis"], xg, xc, cfg.use_noise["gdis"], cfg.noise_sigma["gdis"], rng, True)
    for y, k in ((yi, "yi"), (yv, "yv"), (yg, "yg")):
        assert y.shape == fx["dis/" + k].shape
        assert G.relerr(y.detach().numpy(), fx["dis/" + k]) < TOL, k
    tot = (yi * torch.linspace(-1, 1, yi.numel()).view(yi.shape)).sum() \
        + (yv * torch.linspace(1, -1, yv.numel()).view(yv.shape)).sum() \
        + (yg * torch.linspace(-0.5, 1.5, yg.numel()).view(yg.shape)).sum()
    tot.backward()
    assert G.relerr(G.sub(xg.grad, 11), fx["dis/grad_xg_sub"]) < 1e-5
    assert G.relerr(G.sub(xc.grad, 11), fx["dis/grad_xc_sub"]) < 1e-5
    for m in ("idis", "vdis", "gdis"):
        for k, p in st[m].items():
            key = f"dis/grad/{m}/{k}"
            if key in fx:
                assert G.relerr(p.grad.numpy(), fx[key]) < 1e-5, key


@pytest.mark.parametrize("lname,kind", [("adv", "adversarial-loss"), ("hinge", "hinge-loss")])
def test_losses(lname, kind):
    fx = G.load("modules_depth_w6.npz")
    ys = []
    for i in range(3):
        yr = torch.from_numpy(fx[f"loss/{lname}/dis{i}/yr"]).requires_grad_(True)
        yf = torch.from_numpy(fx[f"loss/{lname}/dis{i}/yf"]).requires_grad_(True)
        v = O.dis_loss(kind, yr, yf)
        gr, gf = torch.autograd.grad(v, [yr, yf])
        assert abs(v.item() - float(fx[f"loss/{lname}/dis{i}/value"])) < 1e-6
        assert np.allclose(gr.numpy(), fx[f"loss/{lname}/dis{i}/gr"], atol=1e-7)
        assert np.allclose(gf.numpy(), fx[f"loss/{lname}/dis{i}/gf"], atol=1e-7)
        ys.append(yr.detach().clone().requires_grad_(True))
    v = O.gen_loss(kind, *ys)
    assert abs(v.item() - float(fx[f"loss/{lname}/gen/value"])) < 1e-6
    gq = torch.autograd.grad(v, ys, allow_unused=True)
    for i, gg in enumerate(gq):
        ref = fx[f"loss/{lname}/gen/g{i}"]
        got = gg.numpy() if gg is not None else np.zeros_like(ref)
        assert np.allclose(got, ref, atol=1e-7)


@pytest.mark.parametrize("fixture", ["step_depth_adv_g1.npz", "step_depth_adv_g1_evalstart.npz", "step_flow_hinge_g2.npz"])
def test_step(fixture):
    fx = G.load(fixture)
    cfg = G.cfg_of(fx, loss=str(fx["meta/loss"]), num_gen_update=int(fx["meta/num_gen_update"]),
                   num_dis_update=int(fx["meta/num_dis_update"]), start_in_eval=bool(fx["meta/start_in_eval"]))
    st = G.states(fx)
    B = cfg.batchsize
    gd = torch.Generator().manual_seed(int(fx["meta/seed_data"]))
    lo, hi = (-0.5, 0.5) if cfg.channel == 2 else (-1.0, 1.0)
    xc_real = torch.rand(B, 3, 16, 64, 64, generator=gd) * 2 - 1
    xg_real = torch.rand(B, cfg.channel, 16, 64, 64, generator=gd) * (hi - lo) + lo
    torch.manual_seed(int(fx["meta/seed_run"]))
    so = O.StepOracle(cfg, st)
    for it in range(1, int(fx["meta/iters"]) + 1):
        r = so.step(xc_real, xg_real, int(fx["meta/t_rands"][it - 1]))
        got = [r["loss_idis"], r["loss_vdis"], r["loss_gdis"], r["loss_gen"]]
        assert np.allclose(got, fx["losses"][it - 1], rtol=2e-5, atol=1e-6), (it, got, fx["losses"][it - 1])
        for m in G.MODELS:
            for k, v in so.st[m].items():
                ref = fx[f"after{it}/{m}/{k}"]
                v = v.detach().float().reshape(-1).double()
                got = np.concatenate([[v.abs().sum().item(), v.sum().item()], v[:8].numpy()])
                n = len(got)
                assert np.allclose(got, ref[:n], rtol=5e-4, atol=1e-5), (it, m, k, got, ref)


def test_fullwidth_scalars_shapes():
    fx = G.load("fullwidth_isogd_depth.npz")
    assert fx["yi"].shape == (2, 4, 4) and fx["yv"].shape == (2, 4, 4, 4) and fx["yg"].shape == (2, 3, 4, 4)


def test_sampling_path():
    """util.generate_samples / videos_to_numpy / images_to_numpy — uint8, bit-exact."""
    fx = G.load("sampling_depth_w4.npz"); cfg = G.cfg_of(fx, B=2); st = G.states(fx)
    torch.manual_seed(int(fx["meta/seed_run"]))
    xg, xc = O.generate_samples_depth(st["ggen"], st["cgen"], 3, 2, 16, cfg.dim_z_content, cfg.dim_z_motion, cfg.dim_z_color, O.TorchRng())
    assert xg.dtype == np.uint8 and xg.shape == (3, 3, 16, 64, 64) and xc.shape == (3, 3, 16, 64, 64)
    assert np.array_equal(xg.reshape(-1)[::13], fx["xg_sub"]) and int(xg.astype(np.int64).sum()) == int(fx["xg_sum"])
    assert np.array_equal(xc.reshape(-1)[::13], fx["xc_sub"]) and int(xc.astype(np.int64).sum()) == int(fx["xc_sum"])
    assert np.array_equal(O.videos_to_uint8(torch.from_numpy(fx["conv_in"])), fx["conv_out"])


def test_segmentation_data_paths():
    """palette, argmax colouring (first maximum on ties) and the dataset's one-hot decode,
    against fixtures produced by the reference's own functions."""
    fx = np.load(G.path("segmentation_io.npz")) if hasattr(G, "path") else np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "segmentation_io.npz"))
    from dcvgan_amd.sampling import SEGM_PALETTE_U8
    assert np.array_equal(np.array(SEGM_PALETTE_U8, dtype=np.uint8), fx["palette_u8"])
    assert np.array_equal(O.segmentation_to_color(fx["probs"], fx["palette_u8"]), fx["color"])
    assert np.array_equal(O.segmentation_one_hot(fx["labels"]), fx["onehot"])
