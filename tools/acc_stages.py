"""Where does the HIP path's distance from fp64 come from?  For one config at full width: forward errors of every
stage (HIP and CPU fp32, each against the fp64 oracle), then the discriminators alone on IDENTICAL inputs (the fp32
oracle's fakes), so generator forward error and discriminator kernel error are separated.
Usage: python tools/acc_stages.py [config] [B]"""
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, trainer
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.rng import InjectedRng
from oracle import dcvgan_oracle as O
from tests import fullwidth as FW

native.lib()
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "surreal-depth1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = CONFIGS[name].scaled(batchsize=B)
torch.manual_seed(123)
models = trainer.build_models(cfg, torch.device("cpu"))
t = 7
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
r32 = FW.oracle_gen_pass(cfg, models, 321, t)
r64 = FW.oracle_gen_pass(cfg, models, 321, t, torch.float64)
hip = FW.hip_gen_pass(cfg, models, r32["log"], t, dev)
print("stage        cpu32        hip")
for k in ("xg", "xc", "yi", "yv", "yg", "loss"):
    print("%-8s %.3e   %.3e" % (k, rel(r32[k], r64[k]), rel(hip[k], r64[k])))
print("per-frame error of xc (hip):", ["%.1e" % rel(hip["xc"][:, :, f], r64["xc"][:, :, f]) for f in range(16)])

# discriminators alone, identical inputs (the fp32 oracle's fakes, detached), hinge generator loss on them
xg_in, xc_in = r32["xg"].detach(), r32["xc"].detach()


def d_only(dtype):
    st = FW.states_of({n: models[n].cpu() for n in ("idis", "vdis", "gdis")}, dtype)
    rng = O.ReplayRng([(k, v.to(dtype)) for k, v in dlog]) if dlog is not None else O.TorchRng()
    xg = xg_in.detach().clone().to(dtype).requires_grad_(True); xc = xc_in.detach().clone().to(dtype).requires_grad_(True)
    yi = O.idis_forward(st["idis"], xg[:, :, t], xc[:, :, t], cfg.use_noise["idis"], cfg.noise_sigma["idis"], rng, True)
    yv = O.vdis_forward(st["vdis"], xg, xc, cfg.use_noise["vdis"], cfg.noise_sigma["vdis"], rng, True)
    yg = O.gdis_forward(st["gdis"], xg, xc, cfg.use_noise["gdis"], cfg.noise_sigma["gdis"], rng, True)
    O.gen_loss(cfg.loss, yi, yv, yg).backward()
    return st, xg.grad, xc.grad, (yi, yv, yg), rng


dlog = None
torch.manual_seed(5)
st32, gxg32, gxc32, y32, rng = d_only(torch.float32)
dlog = rng.log
st64, gxg64, gxc64, y64, _ = d_only(torch.float64)
FW.to_device(models, dev)
r = InjectedRng(dlog)
for m in models.values():
    m._rng = r
    m.zero_grad()
xg = xg_in.detach().clone().to(dev).requires_grad_(True); xc = xc_in.detach().clone().to(dev).requires_grad_(True)
yi = models["idis"](xg[:, :, t], xc[:, :, t]); yv = models["vdis"](xg, xc); yg = models["gdis"](xg, xc)
trainer.build_loss(cfg).compute_gen_loss(yi, yv, yg).backward()
print("\ndiscriminators alone on identical inputs")
for k, a, b, c in (("yi", y32[0], yi, y64[0]), ("yv", y32[1], yv, y64[1]), ("yg", y32[2], yg, y64[2]), ("d/dxg", gxg32, xg.grad, gxg64), ("d/dxc", gxc32, xc.grad, gxc64)):
    print("%-30s %.3e   %.3e" % (k, rel(a.detach(), c.detach()), rel(b.detach(), c.detach())))
for n in ("idis", "vdis", "gdis"):
    for k, p in models[n].named_parameters():
        g64 = st64[n][k].grad
        if g64 is None:
            continue
        print("%-30s %.3e   %.3e" % (n + "/" + k, rel(st32[n][k].grad, g64), rel(p.grad, g64)))
