"""CPU: the numpy restatement of dataset.py:125-186 that tests/test_dataprep_gpu.py uses as its formula oracle,
against the arrays the reference's own VideoDataset.__getitem__ returned (tests/golden/dataset_norm.npz)."""
import numpy as np

from tests import goldenio as G


def test_formulas_match_the_reference_dataset():
    fx = G.load("dataset_norm.npz")
    for i in range(3):
        assert np.array_equal(fx[f"mock/{i}/color_in"].transpose(3, 0, 1, 2).astype(np.float32) / 127.5 - 1.0, fx[f"mock/{i}/color_out"])
        assert np.array_equal(fx[f"mock/{i}/depth_in"].transpose(3, 0, 1, 2).astype(np.float32) / 127.5 - 1.0, fx[f"mock/{i}/depth_out"])
        # test_dataset.py:63-95: solid frames, colours / grey levels cycle with (clip + frame) % 3
        back = ((fx[f"mock/{i}/color_out"].transpose(1, 2, 3, 0) + 1) / 2 * 255).astype(np.uint8)
        for j, frame in enumerate(back):
            assert (frame == np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255]][(i + j) % 3])).all()
    assert np.array_equal(fx["mock/0/flow_in"].transpose(3, 0, 1, 2) / float(64), fx["mock/0/flow_out"])
    for i in range(4):
        d = fx[f"surreal/{i}/depth_in"]
        want = np.ones(d.shape, dtype=np.float32)
        mask = d < 1e10
        if mask.any():
            h = d[mask]
            ma, mi = h.max(), h.min()
            if ma - mi > 0:
                h = (h - mi) / (ma - mi)
            want[mask] = h * 1.8 - 1.0
        assert np.array_equal(want[None], fx[f"surreal/{i}/depth_out"])
        assert np.array_equal(np.eye(25, dtype=np.float32)[fx[f"surreal/{i}/segm_in"]].transpose(3, 0, 1, 2), fx[f"surreal/{i}/segm_out"])
