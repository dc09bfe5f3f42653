"""ctypes binding of libdcvgan_hip.so (include/dcvgan_hip.h).

There is NO fallback: if the library is missing, or a call fails, a
``NativeError`` is raised.  PyTorch is used for device memory and streams only;
every pointer handed over is ``tensor.data_ptr()`` and the stream is the current
HIP stream of the calling thread (autograd's backward thread included).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCV_LIB_PATH") or os.path.join(_HERE, "libdcvgan_hip.so")      # DCV_LIB_PATH: another build of the same library (A/B runs of tools/)

ACT_NONE, ACT_LEAKY, ACT_TANH = 0, 1, 2
DCV_OK, DCV_EINVAL, DCV_EWORKSPACE, DCV_EHIP, DCV_EUNSUPPORTED = 0, -1, -2, -3, -4      # include/dcvgan_hip.h


class NativeError(RuntimeError):
    pass


class Dims5(C.Structure):
    _fields_ = [("n", C.c_int32), ("c", C.c_int32), ("d", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("sn", C.c_int64), ("sc", C.c_int64), ("sd", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64)]


class WPack(C.Structure):
    """dcv_wpack: caller-owned K-major packed weights of one (layer, pass, input geometry)."""
    _fields_ = [("buf", C.c_void_p), ("bytes", C.c_size_t), ("ready", C.c_int32),
                ("precision", C.c_int32)]   # dcv_conv_effective_precision(g) the buffer was / is to be packed for: 1 fp32, 2 bf16 products, 3 fp32-on-bf16


class ConvGeom(C.Structure):
    _fields_ = [("kd", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("sd", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
                ("pd", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32),
                ("transposed", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("mfma", C.c_int32)]     # 0 = process default, 1 = fp32, 2 = bf16 products, 3 = fp32 emulated on the bf16 pipe (per-module switch)

    def key(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


_P = C.c_void_p
_D = C.POINTER(Dims5)
_G = C.POINTER(ConvGeom)
_SIGS = {
    "dcv_last_error": (C.c_char_p, []),
    "dcv_version": (C.c_int, []),
    "dcv_abi_struct_sizes": (None, [C.POINTER(C.c_size_t)]),
    "dcv_conv_effective_precision": (C.c_int, [_G]),
    "dcv_launch_count": (C.c_uint64, []),
    "dcv_debug_kernel_info": (C.c_int, [C.c_char_p, C.c_size_t]),
    "dcv_debug_last_kernel": (C.c_char_p, []),
    "dcv_set_precision": (C.c_int, [C.c_int]),
    "dcv_get_precision": (C.c_int, []),
    "dcv_conv_workspace_bytes": (C.c_size_t, [_G, _D, _D, C.c_int]),
    "dcv_conv_packed_bytes": (C.c_size_t, [_G, _D, _D, C.c_int]),
    "dcv_conv_forward": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, C.c_float, _P, _P, C.c_size_t, _P]),
    "dcv_conv_stats_bytes": (C.c_size_t, [_G, _D, _D]),
    "dcv_conv_forward_stats": (C.c_int, [_G, _P, _D, _P, _P, _D, _P, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), _P, _P, C.c_size_t, _P]),
    "dcv_conv_backward_data": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, _P, _P, C.c_size_t, _P]),
    "dcv_conv_backward_data_gated": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, _P, _D, C.c_int, C.c_float, _P, _P, C.c_size_t, _P]),
    "dcv_bn_forward_stats_only": (C.c_int, [_P, _D, _P, _P, _P, _P, _P, C.c_float, C.c_float, _P, C.c_int, C.c_int, _P]),
    "dcv_bn_apply": (C.c_int, [_P, _D, _P, _D, _P, _P, _P, _P, _P, C.c_int, C.c_float, _P]),
    "dcv_conv_forward_bn": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, C.c_float, _P, _P, C.c_size_t, C.c_int, _P, _D, _P, _P, _P, _P, C.c_int, C.c_float, _P]),
    "dcv_conv_backward_weight_bn": (C.c_int, [_G, _P, _D, _P, _D, _P, C.c_int, _P, C.c_size_t, C.c_int, _P, _D, _P, _P, _P, _P, C.c_int, C.c_float, _P]),
    "dcv_conv_backward_data_bn_workspace_bytes": (C.c_size_t, [_D, C.c_int]),
    "dcv_conv_backward_data_bn": (C.c_int, [_G, _P, _D, _P, _P, _D, _P, _P, C.c_size_t, C.c_int, _P, _D, _P, _P, _P, _P, C.c_int, C.c_float, _P, _D, _P, _P,
                                            _P, C.c_size_t, C.POINTER(C.c_int), _P]),
    "dcv_conv_backward_weight": (C.c_int, [_G, _P, _D, _P, _D, _P, _P, C.c_size_t, _P]),
    "dcv_conv_backward_weight_acc": (C.c_int, [_G, _P, _D, _P, _D, _P, C.c_int, _P, C.c_size_t, _P]),
    "dcv_bn_workspace_bytes": (C.c_size_t, [C.c_int]),
    "dcv_bn_act_forward": (C.c_int, [_P, _D, _P, _D, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "dcv_bn_act_forward_stats": (C.c_int, [_P, _D, _P, _D, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "dcv_bn_act_backward": (C.c_int, [_P, _D, _P, _D, _P, _D, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P, _P, _P, C.c_size_t, _P]),
    "dcv_act_forward": (C.c_int, [_P, _D, _P, _D, C.c_int, C.c_float, _P]),
    "dcv_act_backward": (C.c_int, [_P, _D, _P, _D, _P, _D, C.c_int, C.c_float, _P]),
    "dcv_axpby": (C.c_int, [_P, _D, C.c_float, _P, _D, C.c_float, _P, _D, _P]),
    "dcv_noise_add": (C.c_int, [_P, _D, _P, _D, C.c_float, C.c_uint64, C.c_uint64, _P]),
    "dcv_normal_fill": (C.c_int, [_P, C.c_int64, C.c_uint64, C.c_uint64, _P]),
    "dcv_normal_fill_many": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, _P]),
    "dcv_dropout_mask": (C.c_int, [_P, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, _P]),
    "dcv_decode_video": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P]),
    "dcv_surreal_depth": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "dcv_softmax_channels_forward": (C.c_int, [_P, _D, _P, _D, _P]),
    "dcv_softmax_channels_backward": (C.c_int, [_P, _D, _P, _D, _P, _D, _P]),
    "dcv_segm_onehot": (C.c_int, [_P, _D, _P, _D, _P]),
    "dcv_segm_to_rgb": (C.c_int, [_P, _D, _P, _P, _P]),
    "dcv_decode_segmentation": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "dcv_videos_to_uint8": (C.c_int, [_P, _D, _P, C.c_int, _P]),
    "dcv_flow_to_rgb": (C.c_int, [_P, _D, C.c_float, _P, _P, _P]),
    "dcv_gan_loss": (C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int, _P, _P]),
    "dcv_scale_dev": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "dcv_gru_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dcv_gru_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "dcv_gru_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "dcv_cl_packed_bytes": (C.c_size_t, [_G, _D, _D, C.c_int]),
    "dcv_cl_pack_weights": (C.c_int, [_G, _D, _D, C.c_int, _P, _P, C.c_size_t, _P]),
    "dcv_cl_conv_workspace_bytes": (C.c_size_t, [_G, _D, _D, C.c_int]),
    "dcv_cl_conv_forward": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "dcv_cl_conv_stats_bytes": (C.c_size_t, [_G, _D, _D]),
    "dcv_cl_conv_forward_stats": (C.c_int, [_G, _P, _D, _P, _P, _D, _P, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), _P, C.c_size_t, _P]),
    "dcv_cl_conv_backward_data": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, _P, C.c_size_t, _P]),
    "dcv_cl_wgrad_workspace_bytes": (C.c_size_t, [_G, _D, _D]),
    "dcv_cl_conv_backward_data_gated": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, _P, _D, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "dcv_cl_conv_backward_weight": (C.c_int, [_G, _P, _D, _P, _D, _P, _P, C.c_size_t, _P]),
    "dcv_cl_conv_backward_weight_acc": (C.c_int, [_G, _P, _D, _P, _D, _P, C.c_int, _P, C.c_size_t, _P]),
    "dcv_cl_from_f32": (C.c_int, [_P, _D, _P, _D, _P]),
    "dcv_cl_to_f32": (C.c_int, [_P, _D, _P, _D, C.c_int, _P]),
    "dcv_cl_elementwise": (C.c_int, [C.c_int, _P, _D, _P, _D, _P, _D, C.c_float, C.c_float, C.c_uint64, C.c_uint64, _P]),
    "dcv_cl_bn_workspace_bytes": (C.c_size_t, [C.c_int]),
    "dcv_cl_bn_act_forward": (C.c_int, [_P, _D, _P, _D, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "dcv_cl_bn_act_forward_stats": (C.c_int, [_P, _D, _P, _D, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "dcv_cl_bn_act_backward": (C.c_int, [_P, _D, _P, _D, _P, _D, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P, _P, _P, C.c_size_t, _P]),
    "dcv_adam_step_multi": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, _P]),
    "dcv_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, _P]),
    # the fp16 build of the channels-last path (same signatures; ABI 3)
    "dcv_clf16_packed_bytes": (C.c_size_t, [_G, _D, _D, C.c_int]),
    "dcv_clf16_conv_workspace_bytes": (C.c_size_t, [_G, _D, _D, C.c_int]),
    "dcv_clf16_pack_weights": (C.c_int, [_G, _D, _D, C.c_int, _P, _P, C.c_size_t, _P]),
    "dcv_clf16_conv_forward": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "dcv_clf16_conv_stats_bytes": (C.c_size_t, [_G, _D, _D]),
    "dcv_clf16_conv_forward_stats": (C.c_int, [_G, _P, _D, _P, _P, _D, _P, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), _P, C.c_size_t, _P]),
    "dcv_clf16_conv_backward_data": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, _P, C.c_size_t, _P]),
    "dcv_clf16_conv_backward_data_gated": (C.c_int, [_G, _P, _D, _P, _P, _D, C.c_int, _P, _D, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "dcv_clf16_wgrad_workspace_bytes": (C.c_size_t, [_G, _D, _D]),
    "dcv_clf16_conv_backward_weight": (C.c_int, [_G, _P, _D, _P, _D, _P, _P, C.c_size_t, _P]),
    "dcv_clf16_conv_backward_weight_acc": (C.c_int, [_G, _P, _D, _P, _D, _P, C.c_int, _P, C.c_size_t, _P]),
    "dcv_clf16_from_f32": (C.c_int, [_P, _D, _P, _D, _P]),
    "dcv_clf16_to_f32": (C.c_int, [_P, _D, _P, _D, C.c_int, _P]),
    "dcv_clf16_elementwise": (C.c_int, [C.c_int, _P, _D, _P, _D, _P, _D, C.c_float, C.c_float, C.c_uint64, C.c_uint64, _P]),
    "dcv_clf16_bn_workspace_bytes": (C.c_size_t, [C.c_int]),
    "dcv_clf16_bn_act_forward": (C.c_int, [_P, _D, _P, _D, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "dcv_clf16_bn_act_forward_stats": (C.c_int, [_P, _D, _P, _D, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "dcv_clf16_bn_act_backward": (C.c_int, [_P, _D, _P, _D, _P, _D, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P, _P, _P, C.c_size_t, _P]),
}
EXPORTS = tuple(_SIGS)

ABI_VERSION = 4
_lib = None
_lock = threading.Lock()


def lib():
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise NativeError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or dcvgan_amd/csrc/build.sh). There is no CPU/PyTorch fallback for the HIP path.")
                l = C.CDLL(LIB_PATH)
                for name, (res, args) in _SIGS.items():
                    fn = getattr(l, name)  # AttributeError here = header/library mismatch
                    fn.restype = res
                    fn.argtypes = args
                # the library cannot see how large the structs behind our pointers are: compare declarations before the first real call
                if l.dcv_version() != ABI_VERSION:
                    raise NativeError(f"{LIB_PATH} speaks ABI version {l.dcv_version()}, this binding was written for {ABI_VERSION}: rebuild the library")
                sizes = (C.c_size_t * 3)()
                l.dcv_abi_struct_sizes(sizes)
                mine = (C.sizeof(Dims5), C.sizeof(ConvGeom), C.sizeof(WPack))
                if tuple(sizes) != mine:
                    raise NativeError(f"struct sizes differ: library {tuple(sizes)} vs binding {mine} (dcv_dims5, dcv_conv_geom, dcv_wpack)")
                _lib = l
                # DCV_PRECISION=fp32|bf16|f32x6: the process default of the MFMA precision (same as set_precision(); lets a whole test run or an unchanged
                # trainer be put on another precision from outside)
                env = os.environ.get("DCV_PRECISION")
                if env:
                    if env not in ("fp32", "bf16", "f32x6"):
                        raise NativeError(f"DCV_PRECISION={env!r}: expected fp32, bf16 or f32x6")
                    l.dcv_set_precision({"fp32": 0, "bf16": 1, "f32x6": 2}[env])
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise NativeError(f"{what} failed (code {rc}): {lib().dcv_last_error().decode(errors='replace')}")


PRECISION_CODE = {None: 0, "default": 0, "fp32": 1, "bf16": 2, "f32x6": 3}


def set_precision(mode: str):
    """PROCESS DEFAULT — 'fp32', 'bf16' (bf16 MFMA products with fp32 accumulation in the large GEMM kernels: throughput mode) or 'f32x6'
    (experimental: fp32 operands split exactly into three bf16 pieces, six bf16 MFMA products, fp32 accumulation — fp32-grade results on the
    16x faster matrix pipe).  Modules can override it one by one: dcvgan_amd.util.set_precision(module, "bf16" | "fp32" | "f32x6" | None)."""
    check(lib().dcv_set_precision({"fp32": 0, "bf16": 1, "f32x6": 2}[mode]), "dcv_set_precision")


def csrc_digest(cl: bool = False) -> str:
    """sha256 over the sources of the fp32 kernels the headline benchmark runs (csrc/conv_mfma.hip, elementwise.hip, dcv_common.h, build.sh).
    Profiles committed under profiles/ carry it, so a number measured on other kernels is never attached to this build (bench.py).  (The bf16
    channels-last path's sources — conv_cl16.hip, cl_elementwise.hip — are separate translation units and not part of it.)"""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    files = ("conv_mfma.hip", "elementwise.hip", "dcv_common.h", "build.sh")
    if cl:      # the bf16 channels-last path: its own translation units on top (profiles of that path carry this digest)
        files += ("conv_cl16.hip", "cl_elementwise.hip")
    for f in files:
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()


def launch_count() -> int:
    return int(lib().dcv_launch_count())


# --------------------------------------------------------------------------- #
# tensor plumbing
# --------------------------------------------------------------------------- #
def _require(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise NativeError(f"{what}: expected a HIP device tensor, got {t.device} — the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise NativeError(f"{what}: expected float32, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # launches go to the CURRENT device's stream and the library's index tables live on it
        raise NativeError(f"{what}: tensor is on {t.device} but the current device is cuda:{torch.cuda.current_device()} "
                          "(one process per GPU: call torch.cuda.set_device first)")


def dims5(t: torch.Tensor) -> Dims5:
    """Describe a 2-D (N,C), 4-D (N,C,H,W) or 5-D (N,C,D,H,W) tensor view."""
    sz, st = list(t.shape), list(t.stride())
    if t.dim() == 2:
        sz, st = sz + [1, 1, 1], st + [0, 0, 0]
    elif t.dim() == 4:
        sz, st = sz[:2] + [1] + sz[2:], st[:2] + [0] + st[2:]
    elif t.dim() != 5:
        raise NativeError(f"unsupported tensor rank {t.dim()}")
    # size-1 dims may carry arbitrary strides; normalise them for the kernels' contiguity tests
    for i in range(5):
        if sz[i] == 1:
            st[i] = 0
    return Dims5(*sz, *st)


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    if not torch.cuda.is_available():
        raise NativeError("no HIP device: the DCVGAN kernels run on the GPU only (there is no CPU fallback)")
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Scratch(threading.local):
    """Per-thread, per-device grow-only scratch buffers (workspace the C ABI asks
    the caller to own).  Stream-ordered reuse is safe: all launches that touch a
    buffer are on the calling thread's current stream."""

    def __init__(self):
        self.bufs = {}

    def get(self, tag: str, nbytes: int, device) -> torch.Tensor:
        key = (tag, str(device), torch.cuda.current_stream(device).cuda_stream)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


scratch = _Scratch()
