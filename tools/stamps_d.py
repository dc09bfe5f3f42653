"""Cycle stamps of the depth-step (DSTEP) data gradients of the 3-D discriminators: where a workgroup's cycles
go.  Needs the stamped build: gpurun -- 'bash tools/build_stamp.sh && python3 tools/stamps_d.py vdis.5 gdis.9 vdis.1 gdis.5'"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from dcvgan_amd import native as N, ops
from dcvgan_amd.native import dims5, ptr, stream_ptr
dev = torch.device("cuda:0")
L = C.CDLL(os.environ.get("DCV_STAMP_LIB", "/tmp/libdcvgan_hip_stamp.so"))
Lp = C.CDLL("dcvgan_amd/libdcvgan_hip.so")
B = 70
S3, P3 = (1, 2, 2), (0, 1, 1)
CASES = {"vdis.5": (128, 256, (B, 128, 10, 16, 16)), "gdis.9": (64, 128, (B, 64, 9, 16, 16)), "vdis.1": (64, 128, (B, 64, 13, 32, 32)), "gdis.5": (32, 64, (B, 32, 12, 32, 32))}
for name in sys.argv[1:] or ["vdis.5"]:
    cin, cout, xs = CASES[name]
    w = torch.randn(cout, cin, 4, 4, 4, device=dev) * 0.05
    g = ops.conv_geom(w, S3, P3, False)
    dx = torch.empty(xs, device=dev)
    dy = torch.randn(ops._out_shape(g, dx), device=dev)
    dxd, dyd = dims5(dx), dims5(dy)
    for l in (L, Lp):
        l.dcv_conv_workspace_bytes.restype = C.c_size_t
        l.dcv_debug_last_kernel.restype = C.c_char_p
    need = L.dcv_conv_workspace_bytes(C.byref(g), C.byref(dxd), C.byref(dyd), 1)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    def run(lib_, n=5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            assert lib_.dcv_conv_backward_data(C.byref(g), ptr(dy), C.byref(dyd), ptr(w), ptr(dx), C.byref(dxd), 0, None, C.c_void_p(ws.data_ptr()), C.c_size_t(need), stream_ptr()) == 0
        e0.record()
        for _ in range(n):
            lib_.dcv_conv_backward_data(C.byref(g), ptr(dy), C.byref(dyd), ptr(w), ptr(dx), C.byref(dxd), 0, None, C.c_void_p(ws.data_ptr()), C.c_size_t(need), stream_ptr())
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n
    flop = 2.0 * dy.numel() / cout * cout * cin * 64     # as priced in the layer table: every tap of every output position of the FORWARD conv
    for rep in range(2):
        a, b = run(L), run(Lp)
        print(name, "wall ms/op: stamp build %.3f   shipped build %.3f (%.1f TFLOP/s as priced)  kernel %s" % (a, b, flop / b / 1e9, Lp.dcv_debug_last_kernel().decode()))
    buf = np.zeros((4096, 4, 6), dtype=np.uint64)
    L.dcv_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), 4096)
    fw = (buf[..., 3] >> np.uint64(32)).astype(np.float64); buf[..., 3] &= np.uint64(0xffffffff)
    pa = ((buf[..., 5] >> np.uint64(16)) & np.uint64(0xffffff)).astype(np.float64); pb = (buf[..., 5] >> np.uint64(40)).astype(np.float64); buf[..., 5] &= np.uint64(0xffff)
    b = buf.astype(np.float64)
    nb = int((b[:, 0, 5] > 0).sum())
    s = b[:nb]
    n = s[..., 5].mean()
    late = slice(1024, nb) if nb > 1200 else slice(0, nb)
    print(name, "prologue split, workgroups that start beside running ones: index arithmetic %.0f | depth-mask OR + barriers %.0f | accumulators, first tile issue %.0f cycles" % (
        pa[:nb][late].mean(), pb[:nb][late].mean(), (s[..., 0] - pa[:nb] - pb[:nb])[late].mean()))
    pro = s[..., 0].mean(axis=1)
    print(name, "prologue cycles by block id: <1024: %.0f   1024-2047: %.0f   >=2048: %.0f | percentiles 10/50/90: %s" % (
        pro[:1024].mean(), pro[1024:2048].mean() if nb > 1024 else 0, pro[2048:].mean() if nb > 2048 else 0, np.percentile(pro, [10, 50, 90]).round()))
    print(name, "blocks(stamped)", nb, "steps/block %.0f (min %.0f max %.0f) | per wave: prologue %.0f  epilogue %.0f  per step: wait+barrier %.0f  mfma-loop %.0f | final-wait %.0f lifetime %.0f (min %.0f max %.0f)" % (
        n, s[..., 5].min(), s[..., 5].max(), s[..., 0].mean(), s[..., 3].mean(), s[..., 1].sum() / s[..., 5].sum(), s[..., 2].sum() / s[..., 5].sum(), fw[:nb].mean(), s[..., 4].mean(),
        s[..., 4].min(), s[..., 4].max()))
