"""One discriminator (eval mode, cosine cotangent, as tests/test_b100_gpu.py's batch-split identity), forward + backward in native fp32 and in f32x6: the gradient arriving at
every op's output and every parameter gradient, f32x6 against fp32, in backward order — where along the chain do the two separate?
Usage: python tools/x6_chain_probe.py [module] [B]"""
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native as N, ops, trainer
from dcvgan_amd.configs import CONFIGS

which = sys.argv[1] if len(sys.argv) > 1 else "vdis"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N.lib()
dev = torch.device("cuda:0")
cfg = CONFIGS["surreal-depth1"].scaled(batchsize=B)
torch.manual_seed(78)
models = trainer.build_models(cfg, dev)
g = torch.Generator(device=dev).manual_seed(4)
d = models[which]
for mod in d.modules():
    if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
        mod.running_mean.copy_(torch.randn(mod.num_features, device=dev, generator=g) * 0.1)
        mod.running_var.copy_(torch.rand(mod.num_features, device=dev, generator=g) + 0.5)
d.eval()
from dcvgan_amd.rng import PhiloxRng
gg = torch.Generator().manual_seed(1)
xg0 = (torch.rand(B, cfg.channel, 16, 64, 64, generator=gg) * 2 - 1).to(dev)
xc0 = (torch.rand(B, 3, 16, 64, 64, generator=gg) * 2 - 1).to(dev)


def run(mode):
    N.set_precision(mode)
    d._rng = PhiloxRng(5)
    rec, order, count = {}, [], {}
    saved = {}

    def put(tag, t):
        n = count[tag] = count.get(tag, 0) + 1
        rec[f"{tag}#{n}"] = t.detach().clone(); order.append(f"{tag}#{n}")

    for opname in ("conv", "bn_act", "act", "noise_add", "temporal_diff", "cat_channels", "copy_into"):
        orig = saved[opname] = getattr(ops, opname)

        def wrapped(*a, _o=orig, _n=opname, **kw):
            out = _o(*a, **kw)
            if torch.is_tensor(out):
                tag = _n + " " + "x".join(map(str, out.shape))
                put("fwd " + tag, out)
                if out.requires_grad:
                    out.register_hook(lambda gr, tag=tag: put("bwd " + tag, gr))
            return out
        setattr(ops, opname, wrapped)
    xg, xc = xg0.clone().requires_grad_(True), xc0.clone().requires_grad_(True)
    y = d(xg[:, :, 2], xc[:, :, 2]) if which == "idis" else d(xg, xc)
    cot = torch.cos(torch.arange(y.numel(), dtype=torch.float32) * 0.3).view(y.shape).to(dev)
    (y * cot).sum().backward()
    for k, v in saved.items():
        setattr(ops, k, v)
    for n_, p in d.named_parameters():
        put("param grad " + n_, p.grad); p.grad = None
    put("input grad xg", xg.grad); put("input grad xc", xc.grad)
    return rec, order


a, order = run("fp32")
b, _ = run(sys.argv[3] if len(sys.argv) > 3 else "f32x6")
N.set_precision("fp32")
for k in order:
    x, y = a[k].double(), b[k].double()
    r = float((x - y).norm() / x.norm().clamp_min(1e-30))
    print("%-52s %.3e%s" % (k, r, "   <<<" if r > 2e-5 else ""))

# the first elementwise step after which the two modes separate: is it the gate of a handful of elements (a pre-activation within rounding of zero) or arithmetic?
for shape in ("16x256x7x8x8", "16x128x10x16x16"):
    fa, fb = a.get(f"fwd bn_act {shape}#1"), b.get(f"fwd bn_act {shape}#1")
    ga, gb = a.get(f"bwd conv {shape}#1"), b.get(f"bwd conv {shape}#1")
    if fa is None or ga is None:
        continue
    flips = (fa > 0) != (fb > 0)
    same = ~flips
    r_same = float(((ga - gb)[same]).double().norm() / ga[same].double().norm())
    print(f"BatchNorm + LeakyReLU at {shape}: {int(flips.sum())} of {flips.numel()} outputs have another sign in the other mode; gradient behind it, elements with the SAME sign: relative L2 {r_same:.3e}; all elements: {float((ga - gb).double().norm() / ga.double().norm()):.3e}")

# per mode: does the gate the BACKWARD kernel applied agree with the sign of the output the FORWARD kernel stored?  (eval mode: dx = gate * dy * gamma * invstd)
bns = [m for m in d.modules() if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d))]
for tag, recs in (("fp32", a), (sys.argv[3] if len(sys.argv) > 3 else "f32x6", b)):
    for bn, shape in zip(bns, [k.split()[2].split("#")[0] for k in order if k.startswith("fwd bn_act")]):
        y, dy, dx = recs[f"fwd bn_act {shape}#1"], recs[f"bwd bn_act {shape}#1"], recs[f"bwd conv {shape}#1"]
        sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).view(1, -1, *([1] * (y.dim() - 2)))
        ratio = dx.double() / (dy.double() * sc.double())
        ok = dy.abs() > 1e-20
        gate_pos = (ratio - 1.0).abs() < 0.3
        bad = ok & (gate_pos != (y > 0))
        print(f"{tag}: BatchNorm + LeakyReLU at {shape}: {int(bad.sum())} elements whose backward gate disagrees with the stored output's sign" +
              (f"; e.g. y = {float(y[bad][0]):+.3e}, ratio {float(ratio[bad][0]):.3f}" if int(bad.sum()) else ""))
