"""fp32 emulated on the bf16 matrix pipe ("f32x6": three bf16 pieces per operand, six v_mfma_f32_32x32x16_bf16 products, fp32 accumulation) beside
the native fp32 MFMA kernels and the bf16-product mode: speed of the dominant kernel (cgen.up_blocks.5 forward at the benchmark batch) and of its data /
weight gradients, and the relative L2 error of each mode against an fp64 evaluation — on the dominant kernel's geometry and on the K sweep of
tools/acc_probe.py (K = cin x taps from 512 to 8192), with torch's fp32 CPU convolution (the reference's arithmetic) in the same table.
Usage: python3 tools/f32x6_probe.py [B]"""
import sys
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from dcvgan_amd import native, ops

native.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
torch.set_num_threads(16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
MODES = ("fp32", "bf16", "f32x6")


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"== dominant kernel geometry: ConvTranspose2d(128, 64, 4, 2, 1) on ({B * 16}, 128, 32, 32); errors against fp64 on the first 8 frames")
Fr = B * 16
x = torch.randn(Fr, 128, 32, 32); w = torch.randn(128, 64, 4, 4) * 0.02
cot = torch.randn(Fr, 64, 64, 64)
n8 = 8
x8, c8 = x[:n8].double().requires_grad_(True), cot[:n8].double()
w64 = w.double().requires_grad_(True)
y64 = F.conv_transpose2d(x8, w64, None, 2, 1)
gx64, gw64 = torch.autograd.grad((y64 * c8).sum(), [x8, w64])
x8f = x[:n8].clone().requires_grad_(True); wf = w.clone().requires_grad_(True)
y32 = F.conv_transpose2d(x8f, wf, None, 2, 1)
gx32, gw32 = torch.autograd.grad((y32 * cot[:n8]).sum(), [x8f, wf])
print("%-8s | %-28s | %-28s | %-28s" % ("mode", "forward: ms TFLOP/s err", "data gradient", "weight gradient (8 frames)"))
print("%-8s | %28s | %28s | %28s" % ("cpu fp32", "%.2e" % rel(y32.detach(), y64.detach()), "%.2e" % rel(gx32, gx64), "%.2e" % rel(gw32, gw64)))
fl = 2.0 * Fr * 64 * 64 * 64 * 128 * 4
xd, wd, cd = x.to(dev), w.to(dev), cot.to(dev)
res = {}
for mode in MODES:
    native.set_precision(mode)
    g = ops.conv_geom(wd, (2, 2), (1, 1), True)
    with torch.no_grad():
        t_f = timed(lambda: ops.conv(xd, wd, g))
        kf = native.lib().dcv_debug_last_kernel().decode()
    xr = xd.clone().requires_grad_(True); wr = wd.clone().requires_grad_(True)
    y = ops.conv(xr, wr, g)
    t_dx = timed(lambda: torch.autograd.grad(y, xr, cd, retain_graph=True))
    t_dw = timed(lambda: torch.autograd.grad(y, wr, cd, retain_graph=True))
    # errors on the 8-frame subset (the weight gradient sums over frames: evaluate it on the subset itself)
    xs = xd[:n8].clone().requires_grad_(True); ws = wd.clone().requires_grad_(True)
    ys = ops.conv(xs, ws, g)
    gxs, gws = torch.autograd.grad(ys, [xs, ws], cd[:n8])
    e = (rel(ys.detach().cpu(), y64.detach()), rel(gxs.cpu(), gx64), rel(gws.cpu(), gw64))
    res[mode] = (t_f, t_dx, t_dw, e)
    print("%-8s | %7.3f %7.1f %.2e      | %7.3f %7.1f %.2e      | %7.3f %7.1f %.2e      | %s" %
          (mode, t_f, fl / t_f / 1e9, e[0], t_dx, fl / t_dx / 1e9, e[1], t_dw, fl / t_dw / 1e9, e[2], kf), flush=True)
native.set_precision("fp32")
print("speed-up f32x6 / fp32: forward %.2fx, data gradient %.2fx, weight gradient %.2fx; error f32x6 / fp32: %.2f %.2f %.2f" %
      (res["fp32"][0] / res["f32x6"][0], res["fp32"][1] / res["f32x6"][1], res["fp32"][2] / res["f32x6"][2],
       res["f32x6"][3][0] / res["fp32"][3][0], res["f32x6"][3][1] / res["fp32"][3][1], res["f32x6"][3][2] / res["fp32"][3][2]))

print("\n== K sweep (forward, relative L2 against fp64)")
print("%-28s %6s | %9s %9s %9s %9s" % ("layer", "K", "cpu fp32", "hip fp32", "hip f32x6", "hip bf16"))
for name, cin, cout, k3, sp in (("conv2d 32->64 @32", 32, 64, False, 32), ("conv2d 128->128 @16", 128, 128, False, 16), ("conv2d 256->256 @8", 256, 256, False, 8),
                                ("conv2d 512->256 @8", 512, 256, False, 8), ("conv3d 64->128 (vdis.1)", 64, 128, True, 16), ("conv3d 128->256 (vdis.5)", 128, 256, True, 16)):
    if k3:
        xx = torch.randn(4, cin, 7, sp, sp); ww = torch.randn(cout, cin, 4, 4, 4) * 0.05
        conv = lambda a, b: F.conv3d(a, b, None, (1, 2, 2), (0, 1, 1)); s, p = (1, 2, 2), (0, 1, 1)
    else:
        xx = torch.randn(16, cin, sp, sp); ww = torch.randn(cout, cin, 4, 4) * 0.05
        conv = lambda a, b: F.conv2d(a, b, None, 2, 1); s, p = (2, 2), (1, 1)
    K = cin * (64 if k3 else 16)
    y64 = conv(xx.double(), ww.double()); y32 = conv(xx, ww)
    xd2, wd2 = xx.to(dev), ww.to(dev)
    errs = []
    for mode in ("fp32", "f32x6", "bf16"):
        native.set_precision(mode)
        with torch.no_grad():
            errs.append(rel(ops.conv(xd2, wd2, ops.conv_geom(wd2, s, p, False)).cpu(), y64))
    native.set_precision("fp32")
    print("%-28s %6d | %9.2e %9.2e %9.2e %9.2e" % (name, K, rel(y32, y64), errs[0], errs[1], errs[2]), flush=True)
