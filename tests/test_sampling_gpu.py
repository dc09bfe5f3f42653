"""GPU: the forward-only sampling path against the reference's util.generate_samples /
videos_to_numpy / images_to_numpy outputs (fixture) — uint8 results."""
import numpy as np
import pytest
import torch

from oracle import dcvgan_oracle as O
from tests import goldenio as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from dcvgan_amd import native
    native.lib()
    return torch.device("cuda:0")


def test_conversions_are_byte_exact(dev):
    from dcvgan_amd import sampling
    fx = G.load("sampling_depth_w4.npz")
    v = torch.from_numpy(fx["conv_in"]).to(dev)
    assert np.array_equal(sampling.videos_to_numpy(v), fx["conv_out"])
    vs = v.permute(0, 2, 1, 3, 4).contiguous().permute(0, 2, 1, 3, 4)      # the generators' strided layout
    assert np.array_equal(sampling.videos_to_numpy(vs), fx["conv_out"])
    im = torch.from_numpy(fx["img_in"]).to(dev)
    out = sampling.images_to_numpy(im)
    assert out.shape == fx["img_out"].shape and np.array_equal(out, fx["img_out"])


def test_generate_samples_depth(dev):
    """3 videos in batches of 2 (truncation), eval-mode generators, same draws as the reference run."""
    from dcvgan_amd import sampling, trainer
    from dcvgan_amd.rng import InjectedRng
    fx = G.load("sampling_depth_w4.npz"); cfg = G.cfg_of(fx, B=2); st = G.states(fx)
    torch.manual_seed(int(fx["meta/seed_run"]))
    rng = O.TorchRng()
    O.generate_samples_depth(st["ggen"], st["cgen"], 3, 2, 16, cfg.dim_z_content, cfg.dim_z_motion, cfg.dim_z_color, rng)
    models = trainer.build_models(cfg, dev)
    for n in ("ggen", "cgen"):
        models[n].load_state_dict({k: v.clone() for k, v in G.states(fx)[n].items()}); models[n].to(dev)
        models[n]._rng = None
    r = InjectedRng(rng.log)
    models["ggen"]._rng = r; models["cgen"]._rng = r
    xg, xc = sampling.generate_samples(models["ggen"], models["cgen"], 3, 2)
    assert not models["ggen"].training and not models["cgen"].training
    for got, key in ((xg, "xg"), (xc, "xc")):
        assert got.dtype == np.uint8 and got.shape == (3, 3, 16, 64, 64)
        d = np.abs(got.reshape(-1)[::13].astype(np.int16) - fx[key + "_sub"].astype(np.int16))
        # float activations agree to ~1e-6, so a byte can differ by 1 only where a value sits on a rounding edge
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (key, d.max(), (d > 0).mean())
    assert np.array_equal(xg[:, 0], xg[:, 1]) and np.array_equal(xg[:, 0], xg[:, 2])   # depth tiled to RGB


def test_generate_samples_shapes_like_reference_test(dev):
    """test_util.py:22-58: depth and optical-flow, (num, batchsize) in (3,1), (3,2), (3,4)."""
    from dcvgan_amd import generator, sampling
    for info, ch in (("depth", 1), ("optical-flow", 2)):
        ggen = generator.GeometricVideoGenerator(dim_z_content=30, dim_z_motion=10, channel=ch, geometric_info=info, video_length=16, ngf=8).to(dev)
        cgen = generator.ColorVideoGenerator(in_ch=ch, dim_z=10, geometric_info=info, ngf=8).to(dev)
        for num, bs in ((3, 1), (3, 2), (3, 4)):
            xg, xc = sampling.generate_samples(ggen, cgen, num, bs)
            for a in (xg, xc):
                assert isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.shape == (num, 3, 16, 64, 64)


def test_flow_visualisation_properties(dev):
    """Hue encodes direction (OpenCV 8-bit H = degrees/2), value the per-frame normalised magnitude."""
    from dcvgan_amd import sampling
    f = torch.zeros(1, 2, 2, 8, 8, device=dev)
    f[0, 0, 0, :, :4] = 0.5        # frame 0, left half: flow along +x  -> hue 0   -> red
    f[0, 1, 0, :4, 4:] = 0.25      # frame 0, upper right: flow along +y -> hue 45 (90 deg) -> green-ish, half magnitude
    #                                (lower right stays 0, so the frame's min-max range is [0, 4])
    rgb = sampling.geometry_to_color(f, "optical-flow")
    assert rgb.shape == (1, 3, 2, 8, 8) and rgb.dtype == np.uint8
    assert tuple(rgb[0, :, 0, 0, 0]) == (255, 0, 0)
    r, g, b = rgb[0, :, 0, 0, 7]
    assert g > r and g > b and 100 <= g <= 140            # value = (0.25*8 - 0)/(0.5*8 - 0) * 255 ~ 127
    assert rgb[0, :, 1].max() == 0                          # frame 1: no motion -> black


def test_segmentation_branch(dev):
    """SURVEY §8(f).4 on the device: part colouring and one-hot decode byte-exact against the reference fixture,
    argmax -> {-1,+1} remap exact, channel softmax forward/backward against torch."""
    import os
    from dcvgan_amd import dataprep, ops, sampling
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "segmentation_io.npz"))
    probs = torch.from_numpy(fx["probs"]).to(dev)
    assert np.array_equal(sampling.geometry_to_color(probs, "segmentation"), fx["color"])
    # strided input: the generators' (B,T,C,H,W) memory order viewed as (B,C,T,H,W)
    strided = probs.permute(0, 2, 1, 3, 4).contiguous().permute(0, 2, 1, 3, 4)
    assert np.array_equal(sampling.geometry_to_color(strided, "segmentation"), fx["color"])
    labels = torch.from_numpy(fx["labels"]).to(dev)[None]          # one clip of 4 frames
    assert np.array_equal(dataprep.decode_segmentation(labels).cpu().numpy()[0], fx["onehot"])
    frames = probs.permute(0, 2, 1, 3, 4).reshape(-1, 25, 8, 8)      # cgen sees frames
    idx = torch.argmax(frames.cpu(), 1, keepdim=True)
    want = torch.full_like(frames.cpu(), -1.0).scatter_(1, idx, 1.0)  # generator.py:381-385
    assert torch.equal(ops.segm_onehot(frames).cpu(), want)
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(6, 25, 16, 16, generator=g) * 3).requires_grad_(True)
    cot = torch.randn(6, 25, 16, 16, generator=g)
    y_ref = torch.softmax(x, 1)
    (gx_ref,) = torch.autograd.grad((y_ref * cot).sum(), x)
    xd = x.detach().to(dev).requires_grad_(True)
    y = ops.softmax_channels(xd)
    (gx,) = torch.autograd.grad((y * cot.to(dev)).sum(), xd)
    assert (y.cpu() - y_ref).abs().max() < 1e-6       # fp32 exp/divide rounding
    assert (gx.cpu() - gx_ref).abs().max() < 1e-6
