"""Drop-in alias: put this directory on sys.path and `import util` resolves to the HIP-backed
implementation with the reference's names (src/util.py).  See INTEGRATION.md.

Every `util.*` name the unchanged reference touches on the path resolves here
(/root/reference/src/trainer.py:39,131,138-139,152,156,163-164,187; src/train.py:165;
src/infer.py:34-35,72): `current_device`, `init_weights`, `generate_samples`, `videos_to_numpy`,
`images_to_numpy`, `geometric_info_in_color_format`, `make_video_grid`.  The float -> uint8
conversions run on the device (HIP kernels of `dcvgan_amd.sampling`); inputs the reference's trainer
hands over from the host (a DataLoader batch, `trainer.py:148-156`: a CPU tensor for `videos_to_numpy`,
a numpy array for `geometric_info_in_color_format`) are uploaded first — there is no host arithmetic
path, and without the HIP library these raise `NativeError`.
"""
import numpy as _np
import torch as _torch

from dcvgan_amd.util import *  # noqa: F401,F403
from dcvgan_amd import util as _impl
from dcvgan_amd import sampling as _sampling

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})


def _on_device(x):
    """numpy array or tensor -> float32 tensor on the process's current HIP device (`current_device`)."""
    t = _torch.from_numpy(_np.ascontiguousarray(x)) if isinstance(x, _np.ndarray) else x
    return t.detach().to(device=_impl.current_device(), dtype=_torch.float32)


def videos_to_numpy(tensor):
    """util.py:58-79: (B,C,T,H,W) float in [-1,1] -> uint8 numpy, same axis order."""
    return _sampling.videos_to_numpy(_on_device(tensor))


def images_to_numpy(tensor):
    """util.py:31-55: (B,C,H,W) float -> (B,H,W,C) uint8 numpy."""
    return _sampling.images_to_numpy(_on_device(tensor))


def geometric_info_in_color_format(xg, geometric_info):
    """util.py:198-248: geometry clips (numpy or tensor, (B,Cg,T,H,W) float) -> uint8 (B,3,T,H,W).
    Values are clipped to [-1,1] first, as `generate_samples` does before it calls this
    (util.py:306-307); the reference's own conversion of out-of-range floats wraps."""
    return _sampling.geometry_to_color(_on_device(xg), geometric_info)


generate_samples = _sampling.generate_samples
load_model = _sampling.load_model


def make_video_grid(videos, rows, cols):
    """util.py:82-123: (rows*cols, C, T, H, W) -> one (1, C, T, rows*H, cols*W) mosaic video; clip
    r*cols + c lands in grid cell (r, c).  Pure index shuffling on the host array the logger takes."""
    n, ch, t, h, w = videos.shape
    if n != rows * cols:
        raise AssertionError(f"make_video_grid: {n} videos do not fill a {rows} x {cols} grid")
    cells = videos.reshape(rows, cols, ch, t, h, w)
    mosaic = _np.moveaxis(cells, (0, 1), (2, 4))       # (C, T, rows, H, cols, W)
    return mosaic.reshape(1, ch, t, rows * h, cols * w)
