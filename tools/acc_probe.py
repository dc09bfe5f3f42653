"""Accuracy of the fp32 MFMA K chain: relative L2 error of the HIP conv forward / data gradient against an fp64
evaluation, beside torch's fp32 CPU conv, for K = cin * taps from 512 to 8192 — and of the same HIP conv evaluated
in channel chunks summed in fp32 (= what a two-level accumulation inside the kernel would give).
Usage: python tools/acc_probe.py"""
import sys
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from dcvgan_amd import native, ops

native.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
torch.set_num_threads(16)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


print("%-28s %6s | %9s %9s | %s" % ("layer", "K", "cpu fp32", "hip", "hip in channel chunks (chunk K: err)"))
for name, cin, cout, k3, sp in (("conv2d 32->64 @32", 32, 64, False, 32), ("conv2d 128->128 @16", 128, 128, False, 16), ("conv2d 256->256 @8", 256, 256, False, 8),
                                ("conv2d 512->256 @8", 512, 256, False, 8), ("conv3d 64->128 (vdis.1)", 64, 128, True, 16), ("conv3d 128->256 (vdis.5)", 128, 256, True, 16)):
    if k3:
        x = torch.randn(4, cin, 7, sp, sp); w = torch.randn(cout, cin, 4, 4, 4) * 0.05
        conv = lambda xx, ww: F.conv3d(xx, ww, None, (1, 2, 2), (0, 1, 1))
        s, p = (1, 2, 2), (0, 1, 1)
    else:
        x = torch.randn(16, cin, sp, sp); w = torch.randn(cout, cin, 4, 4) * 0.05
        conv = lambda xx, ww: F.conv2d(xx, ww, None, 2, 1)
        s, p = (2, 2), (1, 1)
    K = cin * (64 if k3 else 16)
    y64 = conv(x.double(), w.double())
    y32 = conv(x, w)
    xd, wd = x.to(dev), w.to(dev)
    with torch.no_grad():
        yh = ops.conv(xd, wd, ops.conv_geom(wd, s, p, False)).cpu()
        chunks = []
        for cc in (cin // 2, cin // 4, cin // 8, cin // 16):
            if cc < 4:
                continue
            acc = None
            for c0 in range(0, cin, cc):
                wc = wd[:, c0:c0 + cc].contiguous()
                part = ops.conv(xd[:, c0:c0 + cc], wc, ops.conv_geom(wc, s, p, False))
                acc = part if acc is None else acc + part
            chunks.append((cc * (64 if k3 else 16), rel(acc.cpu(), y64)))
    print("%-28s %6d | %9.2e %9.2e | %s" % (name, K, rel(y32, y64), rel(yh, y64), "  ".join("%d: %.2e" % c for c in chunks)))
