"""Forward-only sampling path: `util.generate_samples`, `util.videos_to_numpy`,
`util.images_to_numpy` and the model loader of `infer.py:14-38`, with the float -> uint8 conversion
done on the device by HIP kernels (dcv_videos_to_uint8, dcv_flow_to_rgb).

Reference: /root/reference/src/util.py:31-79,198-322; /root/reference/src/infer.py:14-84.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np
import torch

from .native import NativeError, _require, check, dims5, lib, ptr, stream_ptr


def _to_uint8(x: torch.Tensor, channel_repeat: int = 1) -> torch.Tensor:
    """uint8((clip(x,-1,1)+1)/2*255) for a (B,C,T,H,W) or (B,C,H,W) float video/image batch."""
    _require(x, "to_uint8 input")
    xd = dims5(x)
    shape = list(x.shape)
    shape[1] *= channel_repeat
    out = torch.empty(shape, dtype=torch.uint8, device=x.device)
    check(lib().dcv_videos_to_uint8(ptr(x), C.byref(xd), C.c_void_p(out.data_ptr()), channel_repeat, stream_ptr()), "dcv_videos_to_uint8")
    return out


def videos_to_numpy(tensor: torch.Tensor) -> np.ndarray:
    """util.py:58-79 — (B,C,T,H,W) float in [-1,1] -> uint8 numpy, same axis order."""
    return _to_uint8(tensor).cpu().numpy()


def images_to_numpy(tensor: torch.Tensor) -> np.ndarray:
    """util.py:31-55 — (B,C,H,W) float -> (B,H,W,C) uint8 numpy."""
    return _to_uint8(tensor).cpu().numpy().transpose(0, 2, 3, 1)


def geometry_to_color(xg: torch.Tensor, geometric_info: str) -> np.ndarray:
    """util.geometric_info_in_color_format (util.py:198-248) on the device: (B,Cg,T,H,W) float ->
    uint8 (B,3,T,H,W).  depth: clipped, tiled to 3 channels; optical-flow: direction/magnitude in HSV."""
    _require(xg, "geometry video")
    if geometric_info == "depth":
        return _to_uint8(xg, 3).cpu().numpy()
    if geometric_info == "optical-flow":
        B, _, T, H, W = xg.shape
        out = torch.empty((B, 3, T, H, W), dtype=torch.uint8, device=xg.device)
        mm = torch.empty(2 * B * T, dtype=torch.float32, device=xg.device)
        fd = dims5(xg)   # the kernel clips to [-1, 1] first, like util.py:306-307, then scales by H (util.py:227)
        check(lib().dcv_flow_to_rgb(ptr(xg), C.byref(fd), float(H), C.c_void_p(out.data_ptr()), ptr(mm), stream_ptr()), "dcv_flow_to_rgb")
        return out.cpu().numpy()
    raise NotImplementedError(f"geometry visualisation for {geometric_info!r} (segmentation is SURVEY §8(f).4)")


def generate_samples(ggen, cgen, num: int, batchsize: int = 20, with_geo: bool = True, verbose: bool = False,
                     desc: str = "generating samples") -> Tuple[np.ndarray, np.ndarray]:
    """util.generate_samples (util.py:251-322): eval-mode generators under no_grad, `num` videos in
    batches of `batchsize` (the last batch is truncated), returns (xg uint8 (num,3,T,H,W) or the
    empty list's stand-in None when with_geo is False, xc uint8 (num,3,T,H,W))."""
    ggen.eval()
    cgen.eval()
    xg_batches: List[np.ndarray] = []
    xc_batches: List[np.ndarray] = []
    for _ in range(0, num, batchsize):
        with torch.no_grad():
            xg = ggen.sample_videos(batchsize)
            xc = cgen.forward_videos(xg)
        if with_geo:
            xg_batches.append(geometry_to_color(xg, ggen.geometric_info))
        xc_batches.append(videos_to_numpy(xc))
    xg_out = np.concatenate(xg_batches)[:num] if with_geo else None
    return xg_out, np.concatenate(xc_batches)[:num]


def load_model(model_path, params_path, device=None):
    """infer.py:14-38: unpickle the whole module object, load the state_dict, re-point `.device`."""
    from . import util
    device = device if device is not None else util.current_device()
    model = torch.load(model_path, map_location="cpu", weights_only=False)
    model.load_state_dict(torch.load(params_path, map_location="cpu"))
    model = model.to(device)
    for m in model.modules():
        if hasattr(m, "device"):
            m.device = device
    return model
