"""CPU: the drop-in nn.Module surface — constructor signatures, attributes,
state_dict keys/shapes/order against the reference's own state_dicts (golden fixtures), init
semantics, and loud failure without a GPU."""
import inspect
import json

import numpy as np
import pytest
import torch

from dcvgan_amd import discriminator as D
from dcvgan_amd import generator as Gm
from dcvgan_amd import loss as Lm
from dcvgan_amd import trainer, util
from dcvgan_amd.configs import CONFIGS, flops_per_video_iteration
from tests import goldenio as G


def test_constructor_signatures():
    sig = lambda c: list(inspect.signature(c.__init__).parameters)[1:]
    assert sig(Gm.GeometricVideoGenerator) == ["dim_z_content", "dim_z_motion", "channel", "geometric_info", "ngf", "video_length"]
    assert sig(Gm.ColorVideoGenerator) == ["in_ch", "dim_z", "geometric_info", "ngf", "video_length"]
    for c in (D.ImageDiscriminator, D.VideoDiscriminator, D.GradientDiscriminator):
        assert sig(c) == ["ch1", "ch2", "use_noise", "noise_sigma", "ndf"]
        assert inspect.signature(c.__init__).parameters["ndf"].default == 64
    assert inspect.signature(Gm.GeometricVideoGenerator.__init__).parameters["ngf"].default == 64


@pytest.mark.parametrize("fixture", ["modules_depth_w6.npz", "modules_flow_w4.npz", "modules_segm_w4.npz"])
def test_state_dict_matches_reference(fixture):
    fx = G.load(fixture); cfg = G.cfg_of(fx)
    models = trainer.build_models(cfg, torch.device("cpu"))
    ref = G.states(fx)
    for n, m in models.items():
        sd = m.state_dict()
        assert list(sd.keys()) == list(ref[n].keys()), n           # names AND order
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(ref[n][k].shape), (n, k)
        m.load_state_dict({k: v.clone() for k, v in ref[n].items()}, strict=True)


def test_state_dict_sizes_at_real_width():
    m = trainer.build_models(CONFIGS["isogd-depth"], torch.device("cpu"))
    assert [len(m[k].state_dict()) for k in trainer.MODEL_NAMES] == [29, 74, 15, 15, 19]
    assert [sum(p.numel() for p in m[k].parameters()) for k in trainer.MODEL_NAMES] == [3165716, 10600768, 662272, 2646784, 666048]


def test_same_seed_gives_the_reference_init():
    """Genuine torch.nn containers built in the reference's order draw the same init stream."""
    fx = G.load("fullwidth_isogd_depth.npz")
    torch.manual_seed(int(fx["meta/seed_init"]))
    models = trainer.build_models(CONFIGS["isogd-depth"].scaled(batchsize=2), torch.device("cpu"))
    for n, m in models.items():
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                assert np.allclose(G.summ(v), fx[f"init_sum/{n}/{k}"], rtol=1e-6, atol=1e-6), (n, k)


def test_init_weights_touches_2d_layers_only():
    torch.manual_seed(0)
    v = D.VideoDiscriminator(1, 3, True, 0.1, 16); v.apply(util.init_weights)
    assert float(v.main[2].weight.std()) == 0.0            # BatchNorm3d keeps gamma = 1
    i = D.ImageDiscriminator(1, 3, True, 0.1, 64); i.apply(util.init_weights)
    assert abs(float(i.main[1].weight.std()) - 0.02) < 2e-3 and abs(float(i.main[2].weight.mean()) - 1.0) < 2e-2


def test_attributes_and_str():
    g = Gm.GeometricVideoGenerator(40, 10, 2, "optical-flow", 32, 16)
    assert (g.dim_z, g.channel, g.video_length, g.ngf, g.geometric_info) == (50, 2, 16, 32, "optical-flow")
    assert json.loads(str(g))["ggen"]["dim_zc"] == 40
    c = Gm.ColorVideoGenerator(2, 10, "optical-flow", 32, 16)
    assert (c.channel, c.in_ch, c.out_ch, c.dim_z, c.n_down_blocks, c.n_up_blocks) == (3, 2, 3, 10, 6, 6)
    assert json.loads(str(c))["cgen"]["n_up_blocks"] == 6
    d = D.GradientDiscriminator(2, 3, False, 0.2, 32)
    assert json.loads(str(d))["vdis"]["ndf"] == 32        # the reference labels gdis "vdis" too
    assert isinstance(Lm.AdversarialLoss(), Lm.Loss) and isinstance(Lm.HingeLoss(), Lm.Loss)
    assert isinstance(g.main[-1], torch.nn.Tanh)
    assert isinstance(Gm.GeometricVideoGenerator(4, 2, 25, "segmentation", 8).main[-1], torch.nn.Softmax)


def test_generic_module_protocol():
    import copy, io
    c = Gm.ColorVideoGenerator(1, 3, "depth", 4, 16)
    c2 = copy.deepcopy(c); c2.train(); c2.eval(); c2.zero_grad(); c2.cpu()
    buf = io.BytesIO(); torch.save(c, buf); buf.seek(0)
    c3 = torch.load(buf, weights_only=False)
    assert list(c3.state_dict()) == list(c.state_dict())
    opt = torch.optim.Adam(c.parameters(), lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5)   # train.py:174
    assert len(opt.param_groups[0]["params"]) == len(list(c.parameters()))


def test_forward_fails_loudly_without_gpu():
    from dcvgan_amd.native import NativeError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = Gm.GeometricVideoGenerator(4, 2, 1, "depth", 4, 16)
    with pytest.raises(NativeError):
        g.sample_videos(1)


def test_configs_restated():
    c = CONFIGS["isogd-depth"]
    assert (c.batchsize, c.seed, c.loss, c.width["vdis"], c.width["gdis"], c.lr["idis"], c.noise_sigma["vdis"]) == (70, 15, "adversarial-loss", 64, 32, 5e-4, 0.1)
    s = CONFIGS["surreal-depth1"]
    assert (s.batchsize, s.loss, s.num_gen_update, s.width["ggen"], s.width["gdis"]) == (100, "hinge-loss", 2, 96, 32)
    f = CONFIGS["isogd-flow"]
    assert (f.channel, f.geometric_info, f.noise_sigma["idis"]) == (2, "optical-flow", 0.2)
    # BASELINE.md §4 / SURVEY §8(d): as-written and "minimal" GFLOP per video and iteration
    for name, want, want_min in (("debug-isogd-depth", 134.44, 96.75), ("isogd-depth", 166.12, 128.33), ("isogd-flow", 165.41, 127.29)):
        assert abs(flops_per_video_iteration(CONFIGS[name]) / 1e9 - want) < 0.05 and abs(flops_per_video_iteration(CONFIGS[name], True) / 1e9 - want_min) < 0.05
    # surreal-depth1 updates the discriminators every 2nd iteration (surreal-depth1.yml:30): 188.31 with every backward, 153.67 on average
    assert abs(flops_per_video_iteration(s) / 1e9 - 153.67) < 0.05


def test_checkpoint_interchange_fixture():
    """SURVEY §8(f).2 — recorded by tests/golden/make_golden.py where the reference is importable:
    our state_dicts load (strict) into the reference classes and vice versa, same keys in the same order,
    and a whole-module pickle written by the reference (trainer.py:75-76) opens through compat/."""
    import json, os
    res = json.load(open(os.path.join(G.GOLDEN, "interchange.json")))
    for n in ("ggen", "cgen", "idis", "vdis", "gdis"):
        assert res[n] == {"ours_into_reference": True, "reference_into_ours": True, "keys": True}, (n, res[n])
    assert res["reference_pickle_opens_through_compat"] == "dcvgan_amd.generator GeometricVideoGenerator 29"


def test_load_model_like_infer(tmp_path):
    """infer.py:14-38 flow on our own artefacts: whole-module pickle + params file -> model on the device."""
    from dcvgan_amd import sampling
    torch.manual_seed(0)
    g = Gm.GeometricVideoGenerator(4, 2, 1, "depth", 4, 16)
    torch.save(g, tmp_path / "ggen_model.pth")
    torch.manual_seed(1)
    g2 = Gm.GeometricVideoGenerator(4, 2, 1, "depth", 4, 16)
    torch.save(g2.state_dict(), tmp_path / "ggen_params_00001.pth")
    m = sampling.load_model(tmp_path / "ggen_model.pth", tmp_path / "ggen_params_00001.pth", torch.device("cpu"))
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), g2.state_dict().values()))
    assert m.device == torch.device("cpu")
