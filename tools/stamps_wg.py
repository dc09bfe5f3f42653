import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from dcvgan_amd import native as N, ops
from dcvgan_amd.native import dims5, ptr, stream_ptr
dev = torch.device("cuda:0")
import os
L = C.CDLL(os.environ.get("DCV_STAMP_LIB", "/tmp/libdcvgan_hip_stamp.so"))   # tools/build_stamp.sh
Fr = 1120
if len(sys.argv) > 1 and sys.argv[1] == "down0":   # cgen.down0: Conv2d(64, 64, 4, 2, 1) at 64x64 -> the 64 x 128 tile (TD = 1)
    x = torch.randn(Fr, 64, 64, 64, device=dev); w = torch.randn(64, 64, 4, 4, device=dev) * 0.05
    g = ops.conv_geom(w, (2, 2), (1, 1), False)
    y = torch.randn(Fr, 64, 32, 32, device=dev); dw = torch.empty_like(w)
else:                                               # cgen.up5: ConvTranspose2d(128, 64, 4, 2, 1) at 32x32 -> the 128 x 128 tile
    x = torch.randn(Fr, 128, 32, 32, device=dev); w = torch.randn(128, 64, 4, 4, device=dev) * 0.05
    g = ops.conv_geom(w, (2, 2), (1, 1), True)
    y = torch.randn(Fr, 64, 64, 64, device=dev); dw = torch.empty_like(w)
xd, yd = dims5(x), dims5(y)
L.dcv_conv_workspace_bytes.restype = C.c_size_t
need = L.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 2)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
def run(lib_, n=5):
    lib_.dcv_conv_workspace_bytes.restype = C.c_size_t
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        assert lib_.dcv_conv_backward_weight(C.byref(g), ptr(x), C.byref(xd), ptr(y), C.byref(yd), ptr(dw), C.c_void_p(ws.data_ptr()), C.c_size_t(need), stream_ptr()) == 0
    e0.record()
    for _ in range(n):
        lib_.dcv_conv_backward_weight(C.byref(g), ptr(x), C.byref(xd), ptr(y), C.byref(yd), ptr(dw), C.c_void_p(ws.data_ptr()), C.c_size_t(need), stream_ptr())
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
Lp = C.CDLL("dcvgan_amd/libdcvgan_hip.so")
for rep in range(2):
    print("wall ms/op: stamp build %.3f   shipped build %.3f" % (run(L), run(Lp)))
buf = np.zeros((4096, 4, 6), dtype=np.uint64)
L.dcv_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 4096)
b = buf.astype(np.float64)
nb = int((b[:, 0, 5] > 0).sum())
s = b[:nb]
per = s[..., :3].sum(axis=(0, 1)) / s[..., 5].sum()
print("blocks", nb, "tiles/block", s[..., 5].mean(), "cycles per tile: mfma-loop %.0f  addr+fold %.0f  wait+barrier %.0f  total %.0f ; wave lifetime %.0f" % (per[0], per[1], per[2], per.sum(), s[..., 4].mean()))
print("shader clock while the kernel runs: %.3f GHz (s_memtime cycles per 100 MHz s_memrealtime tick, mean over waves)" % (s[..., 4] / s[..., 3] * 0.1).mean())
print("per-wave mfma-loop cycles (block 0):", (s[0, :, 0] / s[0, :, 5]).round(), " wait:", (s[0, :, 2] / s[0, :, 5]).round())
