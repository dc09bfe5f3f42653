"""GPU: device-side input normalisation against the numpy formulas of dataset.py:125-186
and the assertions of the reference's test_dataset.py:32-95 (solid-colour frames, value ranges)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from dcvgan_amd import native
    native.lib()
    return torch.device("cuda:0")


def test_color_and_depth_frames(dev):
    from dcvgan_amd import dataprep
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, size=(3, 16, 64, 64, 3), dtype=np.uint8)
    want = frames.transpose(0, 4, 1, 2, 3).astype(np.float32) / 127.5 - 1.0          # dataset.py:130-131
    got = dataprep.decode_color(torch.from_numpy(frames).to(dev)).cpu().numpy()
    assert got.shape == (3, 3, 16, 64, 64) and np.array_equal(got, want)
    assert got.min() >= -1.0 and got.max() <= 1.0
    # test_dataset.py:63-79: solid colours survive the round trip (x+1)/2*255 -> uint8
    colors = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255]], dtype=np.uint8)
    solid = np.broadcast_to(colors[np.arange(16) % 3][None, :, None, None, :], (1, 16, 64, 64, 3)).copy()
    back = dataprep.decode_color(torch.from_numpy(solid).to(dev)).cpu().numpy()
    back = ((back.transpose(0, 2, 3, 4, 1) + 1) / 2 * 255).astype(np.uint8)
    assert np.array_equal(back, solid)
    gray = rng.integers(0, 256, size=(2, 16, 64, 64, 1), dtype=np.uint8)
    d = dataprep.decode_depth(torch.from_numpy(gray).to(dev)).cpu().numpy()
    assert d.shape == (2, 1, 16, 64, 64) and np.array_equal(d, gray.transpose(0, 4, 1, 2, 3).astype(np.float32) / 127.5 - 1.0)


def test_flow(dev):
    from dcvgan_amd import dataprep
    rng = np.random.default_rng(1)
    flow = (rng.standard_normal((2, 16, 64, 64, 2)) * 9).astype(np.float32)
    got = dataprep.decode_flow(torch.from_numpy(flow).to(dev), 64).cpu().numpy()
    assert np.array_equal(got, flow.transpose(0, 4, 1, 2, 3) / float(64))                 # dataset.py:173-174


def test_surreal_depth(dev):
    from dcvgan_amd import dataprep
    rng = np.random.default_rng(2)
    depth = np.full((3, 16, 32, 32), 1e10, dtype=np.float32)
    depth[0, :, 8:24, 10:20] = rng.uniform(2.0, 6.0, size=(16, 16, 10)).astype(np.float32)   # a person
    depth[1, :, 4:8, 4:8] = 3.5                                                               # flat foreground: ma == mi
    # clip 2: no foreground at all
    want = np.ones((3, 1, 16, 32, 32), dtype=np.float32)
    for b in range(3):                                                                        # dataset.py:141-156
        mask = depth[b] < 1e10
        if mask.any():
            h = depth[b][mask]
            ma, mi = h.max(), h.min()
            if ma - mi > 0:
                h = (h - mi) / (ma - mi)
            want[b, 0][mask] = h * 1.8 - 1.0
    got = dataprep.decode_surreal_depth(torch.from_numpy(depth).to(dev)).cpu().numpy()
    assert np.array_equal(got, want)
    assert got[0].min() == -1.0 and np.isclose(got[0][got[0] < 1.0].max(), 0.8)


# --------------------------------------------------------------------------- #
# pinned: outputs of the reference's own VideoDataset.__getitem__ (tests/golden/make_dataset_golden.py)
# --------------------------------------------------------------------------- #
def _fx():
    from tests import goldenio as G
    return G.load("dataset_norm.npz")


def test_against_the_reference_dataset_class(dev):
    """dataset.py:125-186 executed on the reference's mock dataset (+ a SURREAL-format one): same bytes."""
    from dcvgan_amd import dataprep
    fx = _fx()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n = 0
    for i in range(3):   # mock: solid-colour PNG frames (test_dataset.py:63-95) — clips batched as a DataLoader would
        got = dataprep.decode_color(up(fx[f"mock/{i}/color_in"][None])).cpu().numpy()[0]
        assert got.dtype == np.float32 and np.array_equal(got, fx[f"mock/{i}/color_out"]); n += 1
        got = dataprep.decode_depth(up(fx[f"mock/{i}/depth_in"][None])).cpu().numpy()[0]
        assert np.array_equal(got, fx[f"mock/{i}/depth_out"]); n += 1
    batch = np.stack([fx[f"mock/{i}/color_in"] for i in range(3)])
    assert np.array_equal(dataprep.decode_color(up(batch)).cpu().numpy(), np.stack([fx[f"mock/{i}/color_out"] for i in range(3)]))
    got = dataprep.decode_flow(up(fx["mock/0/flow_in"][None]), 64).cpu().numpy()[0]
    assert np.array_equal(got, fx["mock/0/flow_out"]); n += 1
    got = dataprep.decode_color(up(fx["surreal/0/color_in"][None])).cpu().numpy()[0]
    assert np.array_equal(got, fx["surreal/0/color_out"]); n += 1
    # SURREAL depth: a person, a flat foreground (max == min), no background, no foreground — one batch
    depth = np.stack([fx[f"surreal/{i}/depth_in"] for i in range(4)])
    got = dataprep.decode_surreal_depth(up(depth)).cpu().numpy()
    for i in range(4):
        assert np.array_equal(got[i], fx[f"surreal/{i}/depth_out"]), i
        n += 1
    segm = np.stack([fx[f"surreal/{i}/segm_in"] for i in range(4)])
    got = dataprep.decode_segmentation(up(segm)).cpu().numpy()
    for i in range(4):
        assert np.array_equal(got[i], fx[f"surreal/{i}/segm_out"]), i
        n += 1
    assert n == 16
