import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import dcvgan_oracle as O
from tests import goldenio as G
from dcvgan_amd import trainer
from dcvgan_amd.rng import InjectedRng
dev = torch.device("cuda:0")
fx = G.load("step_depth_adv_g1.npz")
cfg = G.cfg_of(fx, loss="adversarial-loss"); B = cfg.batchsize
rel = lambda x,y: float((x.detach().double().cpu()-y.detach().double().cpu()).norm()/y.detach().double().cpu().norm())
st = G.states(fx)
for m in st: O.require_grad(st[m])
torch.manual_seed(123); rng = O.TorchRng(); t=3
xg = O.ggen_sample_videos(st["ggen"], B, 16, cfg.dim_z_content, cfg.dim_z_motion, 1, rng, True); xg.retain_grad()
xc = O.cgen_forward_videos(st["cgen"], xg, cfg.dim_z_color, rng, True); xc.retain_grad()
yi = O.idis_forward(st["idis"], xg[:,:,t], xc[:,:,t], True, 0.1, rng, True)
yv = O.vdis_forward(st["vdis"], xg, xc, True, 0.1, rng, True)
yg = O.gdis_forward(st["gdis"], xg, xc, False, 0.2, rng, True)
l = O.gen_loss("adversarial-loss", yi, yv, yg); l.backward()
models = trainer.build_models(cfg, dev)
for n,m in models.items(): m.load_state_dict({k:v.detach().clone() for k,v in G.states(fx)[n].items()}); m.to(dev)
r = InjectedRng(rng.log)
for m in models.values(): m._rng = r
xgd = models["ggen"].sample_videos(B); xgd.retain_grad()
xcd = models["cgen"].forward_videos(xgd); xcd.retain_grad()
yid = models["idis"](xgd[:,:,t], xcd[:,:,t]); yvd = models["vdis"](xgd, xcd); ygd = models["gdis"](xgd, xcd)
for y in (yid,yvd,ygd): y.retain_grad()
ld = trainer.build_loss(cfg).compute_gen_loss(yid, yvd, ygd); ld.backward()
print("loss", l.item(), ld.item())
print("fwd xg", rel(xgd,xg), "xc", rel(xcd,xc), "yi", rel(yid,yi), "yv", rel(yvd,yv), "yg", rel(ygd,yg))
print("dL/dxc", rel(xcd.grad, xc.grad), "dL/dxg", rel(xgd.grad, xg.grad))
for n in G.MODELS:
    pd = dict(models[n].named_parameters())
    print(n, " ".join("%s=%.1e" % (k.replace("main.","").replace("_blocks",""), rel(pd[k].grad, p.grad)) for k,p in st[n].items() if k in pd and p.grad is not None))
