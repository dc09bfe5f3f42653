"""Every convolution of a config at its real shapes (tools/layer_table.py's list): forward and data gradient in the f32x6 mode against the native fp32 kernels on the same
operands — relative L2 difference per layer.  Both are fp32 arithmetic, so anything above ~1e-6 points at a kernel variant, not at rounding.
Usage: python tools/x6_vs_fp32_layers.py [config] [batch] [mode]"""
import sys
sys.path.insert(0, '.')
cfgname = sys.argv[1] if len(sys.argv) > 1 else "surreal-depth1"
batch = sys.argv[2] if len(sys.argv) > 2 else "100"
mode = sys.argv[3] if len(sys.argv) > 3 else "f32x6"
src = open("tools/layer_table.py").read()
sys.argv = ["layer_table.py", cfgname, "--batch", batch]
exec(src[:src.index("def timeit")])          # argument parsing + the layer list L (name, transposed, cin, cout, k, s, p, xshape, ...)
import torch
from dcvgan_amd import native as N, ops
lib()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


print("%-34s %12s %12s %12s   kernel (%s)" % ("layer", "fwd", "dgrad", "wgrad", mode))
worst = 0.0
for (name, tr, cin, cout, k, s, p, xs, nf, nd, nw) in L:
    g0 = torch.Generator().manual_seed(hash(name) & 0xffff)
    x = torch.randn(xs, generator=g0).to(dev)
    w = (torch.randn(((cin, cout) if tr else (cout, cin)) + k, generator=g0) * 0.05).to(dev)
    out = {}
    for m in ("fp32", mode):
        N.set_precision(m)
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = ops.conv(xr, wr, ops.conv_geom(wr, s, p, tr))
        if m == "fp32":
            dy = torch.randn(y.shape, generator=g0).to(dev)
        kern = N.lib().dcv_debug_last_kernel().decode() if hasattr(N.lib(), "dcv_debug_last_kernel") else ""
        dx, dw = torch.autograd.grad(y, [xr, wr], dy)
        out[m] = (y.detach(), dx, dw, kern)
    a, b = out[mode], out["fp32"]
    r = [rel(a[i], b[i]) for i in range(3)]
    worst = max(worst, *r)
    print("%-34s %12.3e %12.3e %12.3e   %s" % (name, r[0], r[1], r[2], a[3][:70]))
N.set_precision("fp32")
print("largest difference:", "%.3e" % worst)
