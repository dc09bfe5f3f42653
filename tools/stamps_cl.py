#!/usr/bin/env python3
"""Cycle stamps of cl_gather_kernel (bf16 channels-last): where a wave's cycles go, per layer.
Needs the stamped build: EXTRA_HIPCC_FLAGS=-DDCV_CL_STAMP DCV_OUT=dcvgan_amd/alt_libdcvgan_hip.so bash dcvgan_amd/csrc/build.sh
    DCV_LIB_PATH=dcvgan_amd/alt_libdcvgan_hip.so python3 tools/stamps_cl.py [batch]"""
import ctypes as C
import sys

sys.path.insert(0, '.')
import numpy as np
import torch
from dcvgan_amd import native as N, ops, ops_cl
from dcvgan_amd.native import dims5, ptr, stream_ptr, lib

dev = torch.device("cuda:0")
L = lib()
rd = L.dcv_cl_debug_read_stamps
rd.restype = C.c_int
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
F = B * 16
# name, transposed, cin, cout, k, s, p, x shape, which (0 fwd, 1 dgrad)
CASES = [
    ("vdis.1 fwd", False, 64, 128, (4, 4, 4), (1, 2, 2), (0, 1, 1), (B, 64, 13, 32, 32), 0),
    ("vdis.1 dgrad", False, 64, 128, (4, 4, 4), (1, 2, 2), (0, 1, 1), (B, 64, 13, 32, 32), 1),
    ("cgen.up5 fwd", True, 128, 64, (4, 4), (2, 2), (1, 1), (F, 128, 32, 32), 0),
    ("cgen.up5 dgrad", True, 128, 64, (4, 4), (2, 2), (1, 1), (F, 128, 32, 32), 1),
    ("cgen.down0 fwd", False, 64, 64, (4, 4), (2, 2), (1, 1), (F, 64, 64, 64), 0),
    ("cgen.up3 fwd", True, 512, 128, (4, 4), (2, 2), (1, 1), (F, 512, 8, 8), 0),
    ("ggen.3 fwd", True, 768, 384, (4, 4), (2, 2), (1, 1), (F, 768, 4, 4), 0),
]
buf = np.zeros((4096, 8, 8), dtype=np.uint64)
for name, tr, cin, cout, k, s, p, xs, which in CASES:
    x = torch.randn(xs, device=dev)
    w = torch.randn(((cin, cout) if tr else (cout, cin)) + k, device=dev) * 0.05
    g = ops.conv_geom(w, s, p, tr)
    with torch.no_grad():
        x = ops_cl.from_f32(x)
        y = ops_cl.conv(x, w, g)
        dy = ops_cl.from_f32(torch.randn(y.shape, device=dev))
        dx = ops_cl.cl_empty(x.shape, dev)
    xd, yd, dxd = dims5(x), dims5(y), dims5(dx)
    pk0 = ops_cl._packed(w, 0, g, xd, yd, tuple(x.shape)); pk1 = ops_cl._packed(w, 1, g, dxd, yd, tuple(x.shape))

    def run():
        if which == 0:
            wsp, wsn = ops._ws("clconv", L.dcv_cl_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 0), dev)
            N.check(L.dcv_cl_conv_forward(C.byref(g), ptr(x), C.byref(xd), ptr(pk0), ptr(y), C.byref(yd), 0, 0.0, wsp, wsn, stream_ptr()), "f")
        else:
            wsp, wsn = ops._ws("clconv", L.dcv_cl_conv_workspace_bytes(C.byref(g), C.byref(dxd), C.byref(yd), 1), dev)
            N.check(L.dcv_cl_conv_backward_data(C.byref(g), ptr(dy), C.byref(yd), ptr(pk1), ptr(dx), C.byref(dxd), 0, wsp, wsn, stream_ptr()), "d")
    run(); run(); torch.cuda.synchronize()
    rd(buf.ctypes.data_as(C.c_void_p), 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1)
    rd(buf.ctypes.data_as(C.c_void_p), 1)
    b = buf.astype(np.float64)
    ok = b[:, :, 5] > 0
    nb = int(ok[:, 0].sum())
    nw = int(ok[0].sum())
    s_ = b[:nb, :nw]
    steps = s_[..., 6].mean()
    late = s_[1024:] if nb > 1500 else s_
    print("%-16s %s  %.3f ms | workgroups stamped %d x %d waves, %d K steps | per wave, cycles: prologue %.0f  per step: wait+barrier %.0f  DMA issue %.0f  frag reads + MFMA issue %.0f (8 MFMAs = 256)  | epilogue %.0f  lifetime %.0f | late workgroups: prologue %.0f wait/step %.0f" % (
        name, L.dcv_debug_last_kernel().decode()[17:36], ms, nb, nw, steps, s_[..., 0].mean(), s_[..., 1].sum() / s_[..., 6].sum(), s_[..., 2].sum() / s_[..., 6].sum(),
        s_[..., 3].sum() / s_[..., 6].sum(), s_[..., 4].mean(), s_[..., 5].mean(), late[..., 0].mean(), late[..., 1].sum() / late[..., 6].sum()))
