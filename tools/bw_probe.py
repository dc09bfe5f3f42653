"""Streaming rate of the elementwise / BatchNorm passes on the largest activation of the step (1120 x 64 x 64 x 64 fp32 = 1.17 GB)
beside torch's plain elementwise kernels on the same tensor (what the memory system delivers to a trivial kernel)."""
import sys; sys.path.insert(0, '.')
import torch
from dcvgan_amd import ops, native
native.lib()
dev = torch.device("cuda:0")
x = torch.randn(1120, 64, 64, 64, device=dev); y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
gb = x.numel() * 4 / 1e9
ms = t(lambda: y.copy_(x)); print(f"torch copy (r+w): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"torch add scalar (r+w): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")
g = torch.ones(64, device=dev); b = torch.zeros(64, device=dev); rm = torch.zeros(64, device=dev); rv = torch.ones(64, device=dev)
with torch.no_grad():
    ms = t(lambda: ops.bn_act(x, g, b, rm, rv, False, ops.ACT_LEAKY, 0.2, None, 0.1, 1e-5)); print(f"dcv bn_act eval (r+w): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")
    ms = t(lambda: ops.act(x, ops.ACT_LEAKY, 0.2)); print(f"dcv act (r+w): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")
xr = x.clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
z = ops.bn_act(xr, gr, br, rm, rv, True, ops.ACT_LEAKY, 0.2, None, 0.1, 1e-5)
dz = torch.randn_like(z)
ms = t(lambda: torch.autograd.grad(z, [xr, gr, br], dz, retain_graph=True)); print(f"dcv bn_act backward (reduce 2r + apply 2r+w): {ms:.3f} ms  {5 * gb / ms:.2f} TB/s")
