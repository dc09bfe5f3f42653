import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from dcvgan_amd import native as N, ops
from dcvgan_amd.native import dims5, ptr, stream_ptr
dev = torch.device("cuda:0")
import os
L = C.CDLL(os.environ.get("DCV_STAMP_LIB", "/tmp/libdcvgan_hip_stamp.so"))   # tools/build_stamp.sh
Fr = 1120
x = torch.randn(Fr, 128, 32, 32, device=dev); w = torch.randn(128, 64, 4, 4, device=dev) * 0.05
g = ops.conv_geom(w, (2, 2), (1, 1), True)
y = torch.empty(Fr, 64, 64, 64, device=dev)
xd, yd = dims5(x), dims5(y)
L.dcv_conv_workspace_bytes.restype = C.c_size_t
need = L.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 0)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
for _ in range(2):
    rc = L.dcv_conv_forward(C.byref(g), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), 0, C.c_float(0.0), None, C.c_void_p(ws.data_ptr()), C.c_size_t(need), stream_ptr())
    assert rc == 0
torch.cuda.synchronize()
def run(lib_):
    lib_.dcv_conv_workspace_bytes.restype = C.c_size_t
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        lib_.dcv_conv_forward(C.byref(g), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), 0, C.c_float(0.0), None, C.c_void_p(ws.data_ptr()), C.c_size_t(need), stream_ptr())
    e0.record()
    for _ in range(5):
        lib_.dcv_conv_forward(C.byref(g), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), 0, C.c_float(0.0), None, C.c_void_p(ws.data_ptr()), C.c_size_t(need), stream_ptr())
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 5
Lp = C.CDLL("dcvgan_amd/libdcvgan_hip.so")
for rep in range(2):
    print("wall ms/op: stamp build %.3f   shipped build %.3f" % (run(L), run(Lp)))
buf = np.zeros((4096, 4, 6), dtype=np.uint64)
L.dcv_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 4096)
b = buf.astype(np.float64)
its = b[..., 5]
for name, sl in (("first 768 blocks", slice(0, 768)), ("blocks 1536-2304", slice(1536, 2304)), ("last 768", slice(3328, 4096))):
    s = b[sl]
    per = s[..., :4].sum(axis=(0, 1)) / s[..., 5].sum()
    print(name, "cycles/iteration: load-issue %.0f  mfma %.0f  store %.0f  barrier %.0f  total %.0f ; wave lifetime %.0f cycles (%d its)" % (per[0], per[1], per[2], per[3], per.sum(), s[..., 4].mean(), its[sl].mean()))
