# full-width G-phase: HIP vs CPU fp32 vs CPU fp64 — is the deep-gradient error conditioning or a bug?
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import dcvgan_oracle as O
from tests import goldenio as G
from dcvgan_amd import trainer
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.rng import InjectedRng
dev = torch.device("cuda:0")
cfg = CONFIGS["isogd-depth"].scaled(batchsize=2)
torch.manual_seed(99)
models = trainer.build_models(cfg, torch.device("cpu"))
st0 = {n: {k: v.detach().clone() for k, v in m.state_dict().items()} for n, m in models.items()}
t = 9
def cpu(dtype):
    st = {n: {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in d.items()} for n, d in st0.items()}
    for m in st: O.require_grad(st[m])
    class R(O.TorchRng):
        def normal(self, shape): return super().normal(shape).to(dtype)
        def dropout2d_mask(self, n,c,p): return super().dropout2d_mask(n,c,p).to(dtype)
    torch.manual_seed(100); rng = R()
    xg = O.ggen_sample_videos(st["ggen"], 2, 16, 40, 10, 1, rng, True)
    xc = O.cgen_forward_videos(st["cgen"], xg, 10, rng, True)
    yi = O.idis_forward(st["idis"], xg[:,:,t], xc[:,:,t], True, 0.1, rng, True)
    yv = O.vdis_forward(st["vdis"], xg, xc, True, 0.1, rng, True)
    yg = O.gdis_forward(st["gdis"], xg, xc, False, 0.2, rng, True)
    O.gen_loss("adversarial-loss", yi, yv, yg).backward()
    return st, rng
s32, rng = cpu(torch.float32); s64, _ = cpu(torch.float64)
for m in models.values():
    m.to(dev)
    for mod in m.modules():
        if hasattr(mod, "device"): mod.device = dev
r = InjectedRng(rng.log)
for m in models.values(): m._rng = r
xg = models["ggen"].sample_videos(2); xc = models["cgen"].forward_videos(xg)
trainer.build_loss(cfg).compute_gen_loss(models["idis"](xg[:,:,t], xc[:,:,t]), models["vdis"](xg, xc), models["gdis"](xg, xc)).backward()
rel = lambda x,y: float((x.detach().double().cpu()-y.detach().double().cpu()).norm()/y.detach().double().cpu().norm())
print("%-40s %10s %10s %10s" % ("param", "hip-vs-64", "cpu32-vs-64", "hip-vs-32"))
for n in ("ggen","cgen"):
    pd = dict(models[n].named_parameters())
    for k in pd:
        a, b, c = pd[k].grad, s32[n][k].grad, s64[n][k].grad
        e1, e2, e3 = rel(a,c), rel(b,c), rel(a,b)
        if max(e1,e2) > 2e-5 or "recurrent" in k: print("%-40s %10.2e %10.2e %10.2e" % (n+"."+k, e1, e2, e3))
