"""The video discriminator (eval mode, no Noise, cosine cotangent), forward + backward in one precision mode, against an fp64 torch evaluation of the same module that uses
THAT run's own (Leaky)ReLU sign patterns: per-tensor gradient error without the branch lottery.  Usage: python tools/x6_replay_probe.py [B] [mode ...]"""
import dataclasses
import sys
sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from dcvgan_amd import native as N, ops, trainer
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.rng import PhiloxRng

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
modes = sys.argv[2:] or ["fp32", "f32x6"]
N.lib()
dev = torch.device("cuda:0")
cfg = CONFIGS["surreal-depth1"].scaled(batchsize=B)
cfg = dataclasses.replace(cfg, use_noise={k: False for k in cfg.use_noise})
torch.manual_seed(78)
models = trainer.build_models(cfg, dev)
g = torch.Generator(device=dev).manual_seed(4)
d = models["vdis"]
for mod in d.modules():
    if isinstance(mod, torch.nn.BatchNorm3d):
        mod.running_mean.copy_(torch.randn(mod.num_features, device=dev, generator=g) * 0.1)
        mod.running_var.copy_(torch.rand(mod.num_features, device=dev, generator=g) + 0.5)
d.eval()
gg = torch.Generator().manual_seed(1)
xg0 = (torch.rand(B, cfg.channel, 16, 64, 64, generator=gg) * 2 - 1).to(dev)
xc0 = (torch.rand(B, 3, 16, 64, 64, generator=gg) * 2 - 1).to(dev)


class Forced(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, mask):
        ctx.save_for_backward(mask)
        return torch.where(mask, z, 0.2 * z)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return dy * torch.where(mask, 1.0, 0.2).to(dy.dtype), None


def run_hip(mode):
    N.set_precision(mode)
    d._rng = PhiloxRng(5)
    rec, count, saved = {}, {}, {}

    def put(tag, t):
        n = count[tag] = count.get(tag, 0) + 1
        rec[f"{tag}#{n}"] = t.detach().clone()

    for opname in ("conv", "bn_act"):
        orig = saved[opname] = getattr(ops, opname)

        def wrapped(*a, _o=orig, _n=opname, **kw):
            out = _o(*a, **kw)
            tag = _n + " " + "x".join(map(str, out.shape))
            put("fwd " + tag, out)
            out.register_hook(lambda gr, tag=tag: put("bwd " + tag, gr))
            return out
        setattr(ops, opname, wrapped)
    xg, xc = xg0.clone().requires_grad_(True), xc0.clone().requires_grad_(True)
    y = d(xg, xc)
    cot = torch.cos(torch.arange(y.numel(), dtype=torch.float32) * 0.3).view(y.shape).to(dev)
    (y * cot).sum().backward()
    for k, v in saved.items():
        setattr(ops, k, v)
    for n_, p in d.named_parameters():
        rec["param " + n_] = p.grad.detach().clone(); p.grad = None
    rec["input xg"], rec["input xc"] = xg.grad.clone(), xc.grad.clone()
    rec["y"] = y.detach().clone()
    return rec, cot


def replay64(rec, cot):
    P = {n_: p.detach().double().requires_grad_(True) for n_, p in d.named_parameters()}
    bn = {n_: m for n_, m in d.named_modules() if isinstance(m, torch.nn.BatchNorm3d)}
    xg, xc = xg0.double().requires_grad_(True), xc0.double().requires_grad_(True)
    S, Pd = (1, 2, 2), (0, 1, 1)
    with torch.backends.cudnn.flags(enabled=False):
        hg = Forced.apply(F.conv3d(xg, P["conv_g.0.weight"], None, S, Pd), rec["fwd conv 16x32x13x32x32#1".replace("16x", f"{B}x")] > 0)
        hc = Forced.apply(F.conv3d(xc, P["conv_c.0.weight"], None, S, Pd), rec["fwd conv 16x32x13x32x32#2".replace("16x", f"{B}x")] > 0)
        h = torch.cat([hc, hg], 1)
        inter = {}
        for conv_name, bn_name, shape in (("main.1", "main.2", f"{B}x128x10x16x16"), ("main.5", "main.6", f"{B}x256x7x8x8")):
            h = F.conv3d(h, P[conv_name + ".weight"], None, S, Pd)
            h.retain_grad(); inter["bwd conv " + shape + "#1"] = h
            m = bn[bn_name]
            h = F.batch_norm(h, m.running_mean.double(), m.running_var.double(), P[bn_name + ".weight"], P[bn_name + ".bias"], False, 0.0, m.eps)
            h = Forced.apply(h, rec["fwd bn_act " + shape + "#1"] > 0)
            h.retain_grad(); inter["bwd bn_act " + shape + "#1"] = h
        y = F.conv3d(h, P["main.9.weight"], None, S, Pd).squeeze()
        (y * cot.double()).sum().backward()
    out = {"param " + n_: p.grad for n_, p in P.items()}
    out["input xg"], out["input xc"] = xg.grad, xc.grad
    out.update({k: v.grad for k, v in inter.items()})
    out["y"] = y.detach()
    return out


def rel(a, b):
    return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))


res = {}
for m in modes:
    rec, cot = run_hip(m)
    ref = replay64(rec, cot)
    res[m] = {k: rel(rec[k], ref[k]) for k in ref}
N.set_precision("fp32")
keys = ["y"] + [k for k in res[modes[0]] if k.startswith("bwd")] + [k for k in res[modes[0]] if k.startswith("param")] + ["input xg", "input xc"]
print("%-44s " % "tensor (relative L2 against the fp64 replay)" + " ".join("%12s" % m for m in modes))
for k in keys:
    print("%-44s " % k + " ".join("%12.3e" % res[m][k] for m in modes))
