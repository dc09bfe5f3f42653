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


# SURREAL body-part colours as uint8 RGB: util.segm_color(i) * 255 truncated (util.py:325-372, after
# gulvarol/surreal demo/segmColorMap.m); data, checked against tests/golden/segmentation_io.npz
SEGM_PALETTE_U8 = (
    (114, 139, 163), (216, 82, 24), (236, 176, 31), (125, 46, 90), (118, 171, 47), (76, 189, 237), (131, 196, 185),
    (237, 220, 103), (176, 172, 202), (156, 195, 106), (119, 164, 196), (235, 167, 91), (166, 206, 97), (174, 119, 175),
    (201, 201, 201), (189, 218, 183), (234, 190, 212), (237, 237, 166), (93, 78, 162), (157, 0, 65), (243, 211, 167),
    (253, 209, 123), (50, 130, 188), (152, 214, 164), (226, 156, 137),
)


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
    if geometric_info == "segmentation":   # util.py:236-246: argmax over the 25 part maps, SURREAL part palette
        B, Cg, T, H, W = xg.shape
        if Cg != len(SEGM_PALETTE_U8):
            raise NativeError(f"segmentation visualisation expects {len(SEGM_PALETTE_U8)} part channels, got {Cg}")
        pal = torch.tensor(SEGM_PALETTE_U8, dtype=torch.uint8, device=xg.device)
        out = torch.empty((B, 3, T, H, W), dtype=torch.uint8, device=xg.device)
        xd = dims5(xg)
        check(lib().dcv_segm_to_rgb(ptr(xg), C.byref(xd), C.c_void_p(pal.data_ptr()), C.c_void_p(out.data_ptr()), stream_ptr()), "dcv_segm_to_rgb")
        return out.cpu().numpy()
    raise NotImplementedError(f"geometry visualisation for {geometric_info!r}")


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
