"""Three layers, forward / data gradient / weight gradient: native fp32 kernels, f32x6 and bf16 products against torch's fp64 — relative L2 error AND its bias
(mean signed error in units of std / sqrt(N): ~0 for a rounding that is unbiased) and the error of per-channel SUMS of the output, which a bias hits coherently
(BatchNorm statistics and BatchNorm-parameter gradients are such sums).  Usage: python tools/x6_truth_probe.py [B] [gpu64]"""
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
        def bias(a, b):
            e = a.double().cpu() - b
            return float(e.mean() / e.std() * e.numel() ** 0.5)

        def chsum(a, b):
            dims = tuple(i for i in range(b.dim()) if i != 1)
            return float((a.double().cpu().sum(dims) - b.sum(dims)).norm() / b.sum(dims).norm())
        row.append((m, rel(y.detach(), y64), rel(dx, dx64), rel(dw, dw64), bias(y.detach(), y64), bias(dx, dx64), chsum(y.detach(), y64), chsum(dx, dx64)))
    print(name, " K =", ws[1] * ws[2] * ws[3] * (ws[4] if len(ws) == 5 else 1))
    for m, a, b_, c, ba, bb, ca, cb in row:
        print("    %-6s forward %.3e   data gradient %.3e   weight gradient %.3e   (relative L2 against fp64);  bias of the error, forward %+7.1f  data gradient %+7.1f (x std / sqrt N);"
              "  per-channel sums, forward %.3e  data gradient %.3e" % (m, a, b_, c, ba, bb, ca, cb))
N.set_precision("fp32")
