"""The video discriminator's conv3d 128 -> 256 data gradient on the gradient that REALLY arrives there in a backward pass (eval-mode module, cosine cotangent, as
tests/test_b100_gpu.py::test_batch_split_identity_b100), native fp32 kernels and f32x6 against torch's fp64: is the mode's error input-dependent?
Usage: python tools/x6_real_dy_probe.py [B]"""
import sys
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from dcvgan_amd import native as N, ops, trainer
from dcvgan_amd.configs import CONFIGS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N.lib(); N.set_precision("fp32")
dev = torch.device("cuda:0")
cfg = CONFIGS["surreal-depth1"].scaled(batchsize=B)
torch.manual_seed(78)
models = trainer.build_models(cfg, dev)
g = torch.Generator(device=dev).manual_seed(4)
vdis = models["vdis"]
for mod in vdis.modules():
    if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
        mod.running_mean.copy_(torch.randn(mod.num_features, device=dev, generator=g) * 0.1)
        mod.running_var.copy_(torch.rand(mod.num_features, device=dev, generator=g) + 0.5)
vdis.eval()
cap = {}
orig = ops.conv


def wrapped(x, w, geom, *a, **kw):
    y = orig(x, w, geom, *a, **kw)
    if tuple(w.shape) == (256, 128, 4, 4, 4):
        cap["x"], cap["w"], cap["g"] = x.detach().clone(), w.detach().clone(), geom
        y.register_hook(lambda gr: cap.__setitem__("dy", gr.detach().clone()))
    return y


ops.conv = wrapped
gg = torch.Generator().manual_seed(1)
xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=gg) * 2 - 1).to(dev).requires_grad_(True)
xc = (torch.rand(B, 3, 16, 64, 64, generator=gg) * 2 - 1).to(dev).requires_grad_(True)
y = vdis(xg, xc)
cot = torch.cos(torch.arange(y.numel(), dtype=torch.float32) * 0.3).view(y.shape).to(dev)
(y * cot).sum().backward()
ops.conv = orig
x, w, geom, dy = cap["x"], cap["w"], cap["g"], cap["dy"]
print("captured: x", tuple(x.shape), " dy", tuple(dy.shape), " |dy| rms %.3e max %.3e  zeros %.1f %%" % (float(dy.pow(2).mean().sqrt()), float(dy.abs().max()), 100 * float((dy == 0).float().mean())))
with torch.backends.cudnn.flags(enabled=False):
    xr = x.double().requires_grad_(True)
    y64 = F.conv3d(xr, w.double(), None, (1, 2, 2), (0, 1, 1))
    (dx64,) = torch.autograd.grad(y64, [xr], dy.double())
    (dx64r,) = torch.autograd.grad(F.conv3d(xr, w.double(), None, (1, 2, 2), (0, 1, 1)), [xr], torch.randn_like(dy).double() * float(dy.pow(2).mean().sqrt()))


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


for m in ("fp32", "f32x6", "bf16"):
    N.set_precision(m)
    xd = x.clone().requires_grad_(True)
    yy = ops.conv(xd, w, geom)
    (dx,) = torch.autograd.grad(yy, [xd], dy)
    e = (dx.double() - dx64).abs()
    print("%-6s data gradient on the real dy: relative L2 %.3e   max|err| / rms(dx) %.3e   kernel %s" % (m, rel(dx, dx64), float(e.max() / dx64.pow(2).mean().sqrt()), N.lib().dcv_debug_last_kernel().decode()[:80]))
N.set_precision("fp32")
