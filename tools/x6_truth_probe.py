"""One layer (default vdis.5: conv3d 128 -> 256, 4x4x4, stride (1,2,2)), forward and data gradient: native fp32 kernels and the f32x6 mode against torch's fp64 on the host.
Usage: python tools/x6_truth_probe.py [B]"""
import sys
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from dcvgan_amd import native as N, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N.lib()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
cases = [("vdis.5 conv3d 128->256", (B, 128, 10, 16, 16), (256, 128, 4, 4, 4), (1, 2, 2), (0, 1, 1)),
         ("gdis.5 conv3d 32->64", (B, 32, 12, 32, 32), (64, 32, 4, 4, 4), (1, 2, 2), (0, 1, 1)),
         ("idis.1 conv2d 64->128", (B * 4, 64, 32, 32), (128, 64, 4, 4), (2, 2), (1, 1))]


def rel(a, b):
    return float((a.double().cpu() - b).norm() / b.norm())


for name, xs, ws, s, p in cases:
    x = torch.randn(xs, generator=g); w = torch.randn(ws, generator=g) * 0.05
    conv = F.conv3d if len(xs) == 5 else F.conv2d
    big = "gpu64" in sys.argv          # fp64 truth from torch's own (non-MIOpen) convolution on the device: large batches in seconds
    xr, wr = (t.double().to(dev if big else "cpu").requires_grad_(True) for t in (x, w))
    with torch.backends.cudnn.flags(enabled=not big):
        y64 = conv(xr, wr, None, s, p)
        dy = torch.randn(y64.shape, generator=g)
        dx64, dw64 = torch.autograd.grad(y64, [xr, wr], dy.double().to(y64.device))
    y64, dx64, dw64 = y64.detach().cpu(), dx64.cpu(), dw64.cpu()
    row = []
    for m in ("fp32", "f32x6", "bf16"):
        N.set_precision(m)
        xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        y = ops.conv(xd, wd, ops.conv_geom(wd, s, p, False))
        dx, dw = torch.autograd.grad(y, [xd, wd], dy.to(dev))
        row.append((m, rel(y.detach(), y64), rel(dx, dx64), rel(dw, dw64)))
    print(name, " K =", ws[1] * ws[2] * ws[3] * (ws[4] if len(ws) == 5 else 1))
    for m, a, b_, c in row:
        print("    %-6s forward %.3e   data gradient %.3e   weight gradient %.3e   (relative L2 against fp64)" % (m, a, b_, c))
N.set_precision("fp32")
