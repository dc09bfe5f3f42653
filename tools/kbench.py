"""Per-layer conv microbenchmark at the isogd-depth B=70 shapes: fwd / bwd-data / bwd-weight
through the C ABI, HIP-event timed, TFLOP/s each.  Usage: python tools/kbench.py [filter] [B]"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native as N, ops
import os as _os
N.LIB_PATH = _os.environ.get('DCV_LIB', N.LIB_PATH)
from dcvgan_amd.native import dims5, ptr, stream_ptr, lib

dev = torch.device("cuda:0")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 70
F = B * 16
flt = sys.argv[1] if len(sys.argv) > 1 else ""
# name, transposed, cin, cout, k, s, p, input shape, count per step (fwd, dgrad, wgrad)
L = []
def add(name, tr, cin, cout, k, s, p, xshape, nf, nd, nw): L.append((name, tr, cin, cout, k, s, p, xshape, nf, nd, nw))
ngf = 64
add("ggen.0 convT 50->512 1x1->4", True, 50, 512, (4, 4), (1, 1), (0, 0), (F, 50, 1, 1), 2, 2, 2)
add("ggen.3 convT 512->256 @4", True, 512, 256, (4, 4), (2, 2), (1, 1), (F, 512, 4, 4), 2, 2, 2)
add("ggen.6 convT 256->128 @8", True, 256, 128, (4, 4), (2, 2), (1, 1), (F, 256, 8, 8), 2, 2, 2)
add("ggen.9 convT 128->64 @16", True, 128, 64, (4, 4), (2, 2), (1, 1), (F, 128, 16, 16), 2, 2, 2)
add("ggen.12 convT 64->1 @32", True, 64, 1, (4, 4), (2, 2), (1, 1), (F, 64, 32, 32), 2, 2, 2)
add("cgen.in conv3 1->64 @64", False, 1, 64, (3, 3), (1, 1), (1, 1), (F, 1, 64, 64), 2, 2, 2)
for i, (a, b, sp) in enumerate([(64, 64, 64), (64, 128, 32), (128, 256, 16), (256, 256, 8), (256, 256, 4), (256, 256, 2)]):
    add(f"cgen.down{i} conv {a}->{b} @{sp}", False, a, b, (4, 4), (2, 2), (1, 1), (F, a, sp, sp), 2, 2, 2)
for i, (a, b, sp) in enumerate([(266, 256, 1), (512, 256, 2), (512, 256, 4), (512, 128, 8), (256, 64, 16), (128, 64, 32)]):
    add(f"cgen.up{i} convT {a}->{b} @{sp}", True, a, b, (4, 4), (2, 2), (1, 1), (F, a, sp, sp), 2, 2, 2)
add("cgen.out convT3 128->3 @64", True, 128, 3, (3, 3), (1, 1), (1, 1), (F, 128, 64, 64), 2, 2, 2)
add("idis.g conv 1->32 @64", False, 1, 32, (4, 4), (2, 2), (1, 1), (B, 1, 64, 64), 3, 2, 3)
add("idis.c conv 3->32 @64", False, 3, 32, (4, 4), (2, 2), (1, 1), (B, 3, 64, 64), 3, 2, 3)
add("idis.1 conv 64->128 @32", False, 64, 128, (4, 4), (2, 2), (1, 1), (B, 64, 32, 32), 3, 3, 3)
add("idis.5 conv 128->256 @16", False, 128, 256, (4, 4), (2, 2), (1, 1), (B, 128, 16, 16), 3, 3, 3)
add("idis.9 conv 256->1 @8", False, 256, 1, (4, 4), (2, 2), (1, 1), (B, 256, 8, 8), 3, 3, 3)
S3, P3, K3 = (1, 2, 2), (0, 1, 1), (4, 4, 4)
add("vdis.g conv3d 1->32", False, 1, 32, K3, S3, P3, (B, 1, 16, 64, 64), 3, 2, 3)
add("vdis.c conv3d 3->32", False, 3, 32, K3, S3, P3, (B, 3, 16, 64, 64), 3, 2, 3)
add("vdis.1 conv3d 64->128", False, 64, 128, K3, S3, P3, (B, 64, 13, 32, 32), 3, 3, 3)
add("vdis.5 conv3d 128->256", False, 128, 256, K3, S3, P3, (B, 128, 10, 16, 16), 3, 3, 3)
add("vdis.9 conv3d 256->1", False, 256, 1, K3, S3, P3, (B, 256, 7, 8, 8), 3, 3, 3)
add("gdis.1 conv3d 1->32", False, 1, 32, K3, S3, P3, (B, 1, 15, 64, 64), 3, 2, 3)
add("gdis.5 conv3d 32->64", False, 32, 64, K3, S3, P3, (B, 32, 12, 32, 32), 3, 3, 3)
add("gdis.9 conv3d 64->128", False, 64, 128, K3, S3, P3, (B, 64, 9, 16, 16), 3, 3, 3)
add("gdis.13 conv3d 128->1", False, 128, 1, K3, S3, P3, (B, 128, 6, 8, 8), 3, 3, 3)

import os
if os.environ.get("KSWEEP"):
    L = []
    for cin in (32, 64, 128, 256, 512, 1024):
        add(f"sweep conv {cin}->128 s2 @32", False, cin, 128, (4, 4), (2, 2), (1, 1), (F, cin, 32, 32), 1, 0, 0)
    for cin in (32, 64, 128, 256, 512, 1024):
        add(f"sweep conv {cin}->64 s2 @32", False, cin, 64, (4, 4), (2, 2), (1, 1), (F, cin, 32, 32), 1, 0, 0)

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps

Lb = lib()
tot = [0.0, 0.0, 0.0]; totf = 0.0
print("%-34s %9s | %8s %6s | %8s %6s | %8s %6s" % ("layer", "GF", "fwd ms", "TF/s", "dgrad", "TF/s", "wgrad", "TF/s"))
for (name, tr, cin, cout, k, s, p, xs, nf, nd, nw) in L:
    if flt and flt not in name: continue
    x = torch.randn(xs, device=dev)
    w = torch.randn(((cin, cout) if tr else (cout, cin)) + k, device=dev) * 0.05
    g = ops.conv_geom(w, s, p, tr)
    with torch.no_grad():
        y = ops.conv(x, w, g)
    dy = torch.randn_like(y)
    xd, yd = dims5(x), dims5(y)
    # flops: 2 * (#output positions of the non-transposed-equivalent) * cin * cout * taps-per-output
    if tr:
        macs = x.numel() // cin * cin * cout * (k[0] * k[1])          # every input pixel touches all taps
    else:
        macs = y.numel() // cout * cin * cout * (k[0] * k[1] * (k[2] if len(k) == 3 else 1))
    gf = 2 * macs / 1e9
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    def fwd():
        need = Lb.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 0); wsp, wsn = ops._ws("conv", need, dev)
        N.check(Lb.dcv_conv_forward(C.byref(g), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), 0, 0.0, None, wsp, wsn, stream_ptr()), "f")
    def dgrad():
        need = Lb.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 1); wsp, wsn = ops._ws("conv", need, dev)
        N.check(Lb.dcv_conv_backward_data(C.byref(g), ptr(dy), C.byref(yd), ptr(w), ptr(dx), C.byref(xd), 0, None, wsp, wsn, stream_ptr()), "d")
    def wgrad():
        need = Lb.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 2); wsp, wsn = ops._ws("conv", need, dev)
        N.check(Lb.dcv_conv_backward_weight(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(yd), ptr(dw), wsp, wsn, stream_ptr()), "w")
    tf, td, tw = timeit(fwd), timeit(dgrad), timeit(wgrad)
    tot[0] += tf * nf; tot[1] += td * nd; tot[2] += tw * nw; totf += gf * (nf + nd + nw)
    print("%-34s %9.1f | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f" % (name, gf, tf, gf / tf, td, gf / td, tw, gf / tw))
print("per-step conv time (ms): fwd %.1f dgrad %.1f wgrad %.1f total %.1f ; %.1f TF/s avg" % (tot[0], tot[1], tot[2], sum(tot), totf / sum(tot)))
