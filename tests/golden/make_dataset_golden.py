#!/usr/bin/env python3
"""Golden vectors for the device input pipeline from the REAL reference dataset class.

Run in the build container only (needs /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_dataset_golden.py

`/root/reference/src/dataset.py` is imported with cv2 / skvideo stubbed (neither is installed; dataset.py only
reaches them through dataio) and `dataio.read_img` replaced by a PIL reader with the same contract (uint8, HWC,
RGB; grayscale -> (H, W, 1)).  `VideoDataset.__getitem__` (dataset.py:110-186) then runs unmodified on
  * the reference's own mock dataset `data/processed/mock/train/` (solid-colour PNG frames + optical-flow.npy, the
    data behind src/test/test_dataset.py:32-95) for colour / depth / optical-flow, and
  * a synthetic `surreal` dataset written here in the reference's on-disk format (colour PNGs, depth.npy with the
    1e10 background, segm.npy labels) for the SURREAL depth and segmentation branches (dataset.py:136-156,176-181).
The fixture stores the DISK-ORDER inputs (what a DataLoader would hand to dcvgan_amd.dataprep) and the arrays the
reference returned.  Data only."""
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_dataset():
    for n in ("cv2", "skvideo", "skvideo.io"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["skvideo"].io = sys.modules["skvideo.io"]
    sys.path.insert(0, REF + "/src")
    import util  # noqa: F401  (before generator: import cycle)
    import dataio
    import dataset

    def read_img(path, grayscale=False):   # dataio.read_img's contract (dataio.py:10-35) without OpenCV
        im = Image.open(str(path))
        if grayscale:
            return np.expand_dims(np.asarray(im.convert("L"), dtype=np.uint8), -1)
        return np.asarray(im.convert("RGB"), dtype=np.uint8)

    dataio.read_img = read_img
    return dataset, read_img


def main():
    dataset, read_img = import_dataset()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)   # dataset.PROCESSED_PATH is relative: data/processed/<name>/<mode>
        os.makedirs("data/processed")
        os.symlink(REF + "/data/processed/mock", "data/processed/mock")
        T, S = 16, 64
        # ---- the reference's mock dataset: colour, depth (PNG), optical flow ----
        for geo in ("depth", "optical-flow"):
            ds = dataset.VideoDataset("mock", Path("data/raw/mock"), None, T, S, -1, geo, "train", "png")
            assert len(ds) == 3
            for i in range(len(ds)):
                path, n_frames = ds.video_list[i]
                assert n_frames == T + 1            # -> np.random.randint(1) == 0: frames 0..15
                item = ds[i]
                if geo == "depth":
                    out[f"mock/{i}/color_in"] = np.stack([read_img(path / "color" / f"{t:03d}.png") for t in range(T)])
                    out[f"mock/{i}/color_out"] = item["color"]
                    out[f"mock/{i}/depth_in"] = np.stack([read_img(path / "depth" / f"{t:03d}.png", grayscale=True) for t in range(T)])
                    out[f"mock/{i}/depth_out"] = item["depth"]
                elif i == 0:   # one clip of real flow in full (the arrays are MBs); all three are checked for range below
                    flow = np.load(str(path / "optical-flow.npy"))
                    out[f"mock/{i}/flow_in"] = np.ascontiguousarray(flow[:T])
                    out[f"mock/{i}/flow_out"] = item["optical-flow"]
                if geo != "depth":
                    assert item["optical-flow"].shape == (2, T, S, S) and np.abs(item["optical-flow"]).max() <= 1.0   # test_dataset.py:61-62
        # ---- a synthetic SURREAL-format dataset: depth.npy (1e10 background) and segm.npy ----
        rng = np.random.default_rng(20240)
        root = Path("data/processed/surreal/train")
        H = W = 32
        clips = []
        for i in range(4):
            d = root / str(i)
            (d / "color").mkdir(parents=True)
            for t in range(T + 1):
                Image.fromarray(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)).save(d / "color" / f"{t:03d}.png")
            depth = np.full((T + 1, H, W), 1e10, dtype=np.float32)
            if i == 0:
                depth[:, 8:24, 10:20] = rng.uniform(2.0, 6.0, size=(T + 1, 16, 10)).astype(np.float32)   # a person
            elif i == 1:
                depth[:, 4:8, 4:8] = 3.5                                                                 # flat foreground: max == min
            elif i == 2:
                depth[:, :, :] = rng.uniform(0.5, 9.0, size=(T + 1, H, W)).astype(np.float32)             # no background at all
            # clip 3: no foreground at all
            np.save(d / "depth.npy", depth)
            np.save(d / "segm.npy", rng.integers(0, 25, size=(T + 1, H, W)).astype(np.uint8))
            clips.append(f"{i} {T + 1}\n")
        (root / "list.txt").write_text("".join(clips))
        for geo in ("depth", "segmentation"):
            ds = dataset.VideoDataset("surreal", Path("data/raw/surreal"), None, T, H, -1, geo, "train", "png")
            for i in range(len(ds)):
                path, _ = ds.video_list[i]
                item = ds[i]
                if geo == "depth":
                    out[f"surreal/{i}/depth_in"] = np.load(path / "depth.npy")[:T]
                    out[f"surreal/{i}/depth_out"] = item["depth"]
                    if i == 0:   # random (non-solid) colour frames through the PNG reader, one clip
                        out[f"surreal/{i}/color_in"] = np.stack([read_img(path / "color" / f"{t:03d}.png") for t in range(T)])
                        out[f"surreal/{i}/color_out"] = item["color"]
                else:
                    out[f"surreal/{i}/segm_in"] = np.load(path / "segm.npy")[:T]
                    out[f"surreal/{i}/segm_out"] = item["segmentation"]
        os.chdir("/tmp")
    for k, v in out.items():
        assert isinstance(v, np.ndarray), k
    np.savez_compressed(os.path.join(HERE, "dataset_norm.npz"), **out)
    print("wrote dataset_norm.npz", {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.endswith("_out") and k.split("/")[1] == "0"})


if __name__ == "__main__":
    main()
