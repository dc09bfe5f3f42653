"""Device-side input normalisation: what `VideoDataset.__getitem__`
(/root/reference/src/dataset.py:125-186) does per clip on the CPU, done per *batch* on the GPU so a
data-parallel trainer (560 clips/step at 8 GPUs) is not host-bound.  The DataLoader then only has to
hand over disk-order frames — uint8 (B,T,H,W,C) for images, fp32 (B,T,H,W,2) flow, fp32 (B,T,H,W)
SURREAL depth — ideally from pinned memory with `non_blocking=True`.

Pinned by execution: tests/golden/make_dataset_golden.py imports the reference's dataset module (cv2 / skvideo
stubbed, the image reader shimmed over PIL) and stores what `VideoDataset.__getitem__` returns for the reference's
own mock dataset and for a SURREAL-format one; tests/test_dataprep_gpu.py compares these kernels byte for byte.
"""
from __future__ import annotations

import ctypes as C

import torch

from .native import NativeError, check, lib, ptr, stream_ptr


def _decode(frames: torch.Tensor, div: float, sub: float) -> torch.Tensor:
    if not frames.is_cuda or frames.dim() != 5 or not frames.is_contiguous():
        raise NativeError("decode: expected a contiguous (B,T,H,W,C) tensor on the HIP device")
    if frames.dtype not in (torch.uint8, torch.float32):
        raise NativeError(f"decode: uint8 or float32 frames, got {frames.dtype}")
    B, T, H, W, Cc = frames.shape
    out = torch.empty((B, Cc, T, H, W), dtype=torch.float32, device=frames.device)
    check(lib().dcv_decode_video(C.c_void_p(frames.data_ptr()), int(frames.dtype == torch.uint8), B, T, H, W, Cc, div, sub, ptr(out), stream_ptr()),
          "dcv_decode_video")
    return out


def decode_color(frames_u8: torch.Tensor) -> torch.Tensor:
    """uint8 (B,T,H,W,3) -> fp32 (B,3,T,H,W) in [-1,1]: x/127.5 - 1 (dataset.py:127-131)."""
    return _decode(frames_u8, 127.5, 1.0)


def decode_depth(frames_u8: torch.Tensor) -> torch.Tensor:
    """uint8 grayscale (B,T,H,W,1) -> fp32 (B,1,T,H,W) in [-1,1] (dataset.py:158-168)."""
    return _decode(frames_u8, 127.5, 1.0)


def decode_flow(flow: torch.Tensor, image_size: int) -> torch.Tensor:
    """fp32 (B,T,H,W,2) pixels/frame -> (B,2,T,H,W) / image_size (dataset.py:170-174)."""
    return _decode(flow, float(image_size), 0.0)


def decode_surreal_depth(depth: torch.Tensor) -> torch.Tensor:
    """fp32 (B,T,H,W) metres with background >= 1e10 -> (B,1,T,H,W): foreground min-max normalised per
    clip to [-1, 0.8], background 1.0 (dataset.py:137-156)."""
    if not depth.is_cuda or depth.dim() != 4 or depth.dtype != torch.float32 or not depth.is_contiguous():
        raise NativeError("decode_surreal_depth: expected a contiguous float32 (B,T,H,W) tensor on the HIP device")
    B, T, H, W = depth.shape
    out = torch.empty((B, 1, T, H, W), dtype=torch.float32, device=depth.device)
    mm = torch.empty(2 * B, dtype=torch.float32, device=depth.device)
    check(lib().dcv_surreal_depth(ptr(depth), B, T, H, W, ptr(out), ptr(mm), stream_ptr()), "dcv_surreal_depth")
    return out


def decode_segmentation(labels: torch.Tensor, num_parts: int = 25) -> torch.Tensor:
    """uint8 label frames (B,T,H,W) -> one-hot fp32 (B,num_parts,T,H,W): np.eye(25)[labels] channel-first
    (dataset.py:176-181)."""
    if not labels.is_cuda or labels.dim() != 4 or labels.dtype != torch.uint8 or not labels.is_contiguous():
        raise NativeError("decode_segmentation: expected a contiguous uint8 (B,T,H,W) tensor on the HIP device")
    B, T, H, W = labels.shape
    out = torch.empty((B, num_parts, T, H, W), dtype=torch.float32, device=labels.device)
    check(lib().dcv_decode_segmentation(C.c_void_p(labels.data_ptr()), B, T, H, W, num_parts, ptr(out), stream_ptr()), "dcv_decode_segmentation")
    return out


class DevicePrefetcher:
    """Keeps the next training batch on the device one iteration ahead (trainer.py:293-297 does ``batch[...].to(self.device)`` at the top of
    every iteration: a synchronous, stream-ordered copy that waits for the previous iteration's backward + Adam and serialises 73 MB of
    PCIe traffic per isogd-depth batch in front of the step).  `source` yields dicts (or tuples) of pinned host tensors — what the
    reference's DataLoader produces with pin_memory=True (train.py:101-109); the copies run on a stream of their own while the current
    iteration computes, and ``next()`` hands out DEVICE tensors after making the current stream wait for them, so the trainer's ``.to(device)``
    is a no-op.  Double-buffered: a batch's memory is reused only after the iteration that consumed it has been enqueued behind it."""

    def __init__(self, source, device):
        self.it = iter(source)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self._next = None
        self._preload()

    def _to(self, v):
        return v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v

    def _preload(self):
        try:
            b = next(self.it)
        except StopIteration:
            self._next = None
            return
        with torch.cuda.stream(self.stream):
            if isinstance(b, dict):
                self._next = {k: self._to(v) for k, v in b.items()}
            else:
                self._next = type(b)(self._to(v) for v in b)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        b = self._next
        for v in (b.values() if isinstance(b, dict) else b):
            if isinstance(v, torch.Tensor):
                v.record_stream(cur)
        self._preload()
        return b
