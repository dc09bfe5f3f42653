"""GPU: `Trainer.log_samples`' call sequence (/root/reference/src/trainer.py:126-171) replayed on the HIP modules with
`util` resolved through compat/ — the documented drop-in route must survive the trainer's first log."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from tests import goldenio as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def util():
    """`import util` with compat/ ahead on sys.path, as an unchanged trainer.py would do it."""
    compat = os.path.join(ROOT, "compat")
    saved = sys.modules.pop("util", None)
    sys.path.insert(0, compat)
    try:
        mod = importlib.import_module("util")
        assert mod.__file__.startswith(compat)
        yield mod
    finally:
        sys.path.remove(compat)
        sys.modules.pop("util", None)
        if saved is not None:
            sys.modules["util"] = saved


def test_log_samples_sequence(util):
    from dcvgan_amd import generator, native
    native.lib()
    dev = util.current_device()                                    # trainer.py:39
    assert dev.type == "cuda"
    rows, cols = 2, 2
    num_log = rows * cols
    ggen = generator.GeometricVideoGenerator(dim_z_content=30, dim_z_motion=10, channel=1, geometric_info="depth", video_length=16, ngf=8).to(dev)
    cgen = generator.ColorVideoGenerator(in_ch=1, dim_z=10, geometric_info="depth", ngf=8).to(dev)
    for m in (ggen, cgen):
        m.apply(util.init_weights)                                 # train.py:165
    ggen.eval(); cgen.eval()                                       # trainer.py:126-127
    xg_fake, xc_fake = util.generate_samples(ggen, cgen, num_log, num_log)           # :131
    assert xg_fake.dtype == np.uint8 and xg_fake.shape == xc_fake.shape == (num_log, 3, 16, 64, 64)
    _ = xg_fake[:, 0], xc_fake[:, 0]                               # :134-135 (histograms)
    xg_grid = util.make_video_grid(xg_fake, rows, cols)            # :138-139
    xc_grid = util.make_video_grid(xc_fake, rows, cols)
    x_fake = np.concatenate([xg_grid, xc_grid], axis=-1).transpose(0, 2, 1, 3, 4)    # :140-143
    assert x_fake.shape == (1, 16, 3, rows * 64, 2 * cols * 64)
    np.testing.assert_array_equal(x_fake[0, :, :, 64:128, 64:128], xg_fake[3].transpose(1, 0, 2, 3))
    np.testing.assert_array_equal(x_fake[0, :, :, :64, 128:192], xc_fake[0].transpose(1, 0, 2, 3))

    # the real half: a DataLoader batch is HOST memory (trainer.py:147-156)
    fx = G.load("sampling_depth_w4.npz")
    xc_real = torch.from_numpy(np.concatenate([fx["conv_in"], fx["conv_in"]]))       # (4,3,4,8,8) CPU float
    want_c = np.concatenate([fx["conv_out"], fx["conv_out"]])
    got_c = util.videos_to_numpy(xc_real)                          # :152, a CPU tensor
    assert got_c.dtype == np.uint8 and np.array_equal(got_c, want_c)   # byte-exact against the reference's own output
    xg_real = xc_real[:, :1].data.cpu().numpy()                    # :155, a numpy array in [-1,1]
    got_g = util.geometric_info_in_color_format(xg_real, ggen.geometric_info)        # :156
    assert got_g.shape == (4, 3, 4, 8, 8) and got_g.dtype == np.uint8
    for ch in range(3):                                            # util.py:219-222: tiled to RGB, (x+1)/2*255 truncated
        assert np.array_equal(got_g[:, ch], want_c[:, 0])
    x_real = np.concatenate([util.make_video_grid(got_g, rows, cols), util.make_video_grid(got_c, rows, cols)], axis=-1)
    assert x_real.transpose(0, 2, 1, 3, 4).shape == (1, 4, 3, rows * 8, 2 * cols * 8)

    # evaluate()'s call shape (trainer.py:187-195): keyword arguments, no geometry
    none, xc = util.generate_samples(ggen, cgen, 3, 2, with_geo=False, desc="sampling 3 videos", verbose=True)
    assert none is None and xc.shape == (3, 3, 16, 64, 64)
    img = util.images_to_numpy(torch.from_numpy(fx["img_in"]))
    assert np.array_equal(img, fx["img_out"])
