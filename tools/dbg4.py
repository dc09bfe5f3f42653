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
gd = torch.Generator().manual_seed(int(fx["meta/seed_data"]))
xc_r = torch.rand(B,3,16,64,64,generator=gd)*2-1; xg_r = torch.rand(B,1,16,64,64,generator=gd)*2-1
t = 3
def oracle(mode):
    st = G.states(fx)
    for m in st: O.require_grad(st[m])
    torch.manual_seed(123); rng = O.TorchRng()
    fw = lambda xg, xc: (O.idis_forward(st["idis"], xg[:,:,t], xc[:,:,t], True, 0.1, rng, True), O.vdis_forward(st["vdis"], xg, xc, True, 0.1, rng, True), O.gdis_forward(st["gdis"], xg, xc, False, 0.2, rng, True))
    gen = lambda: (lambda xg: (xg, O.cgen_forward_videos(st["cgen"], xg, cfg.dim_z_color, rng, True)))(O.ggen_sample_videos(st["ggen"], B, 16, cfg.dim_z_content, cfg.dim_z_motion, 1, rng, True))
    if mode >= 1:
        yr = fw(xg_r, xc_r); xg, xc = gen(); yf = fw(xg, xc)
        ld = sum(O.dis_loss("adversarial-loss", a, b) for a, b in zip(yr, yf))
        if mode >= 2: ld.backward()
        for m in ("ggen","cgen"):
            for p in O.trainable(st[m]): p.grad = None
    xg, xc = gen(); xg.retain_grad(); xc.retain_grad()
    l = O.gen_loss("adversarial-loss", *fw(xg, xc)); l.backward()
    return st, rng, xg, xc, l
def hip(mode, log):
    models = trainer.build_models(cfg, dev)
    for n,m in models.items(): m.load_state_dict({k:v.detach().clone() for k,v in G.states(fx)[n].items()}); m.to(dev)
    r = InjectedRng(log)
    for m in models.values(): m._rng = r
    L = trainer.build_loss(cfg)
    fw = lambda xg, xc: (models["idis"](xg[:,:,t], xc[:,:,t]), models["vdis"](xg, xc), models["gdis"](xg, xc))
    gen = lambda: (lambda xg: (xg, models["cgen"].forward_videos(xg)))(models["ggen"].sample_videos(B))
    if mode >= 1:
        yr = fw(xg_r.to(dev), xc_r.to(dev)); xg, xc = gen(); yf = fw(xg, xc)
        ld = sum(L.compute_dis_loss(a, b) for a, b in zip(yr, yf))
        if mode >= 2: ld.backward()
        models["ggen"].zero_grad(); models["cgen"].zero_grad()
    xg, xc = gen(); xg.retain_grad(); xc.retain_grad()
    l = L.compute_gen_loss(*fw(xg, xc)); l.backward()
    return models, xg, xc, l
for mode in (0, 1, 2):
    st, rng, xg, xc, l = oracle(mode)
    models, xgd, xcd, ld = hip(mode, rng.log)
    pd = dict(models["cgen"].named_parameters())
    errs = [rel(pd[k].grad, p.grad) for k,p in st["cgen"].items() if k in pd]
    pg = dict(models["ggen"].named_parameters())
    errg = [rel(pg[k].grad, p.grad) for k,p in st["ggen"].items() if k in pg]
    print("mode", mode, "loss", l.item(), ld.item(), "dxc %.1e dxg %.1e" % (rel(xcd.grad, xc.grad), rel(xgd.grad, xg.grad)), "cgen max %.1e ggen max %.1e" % (max(errs), max(errg)))
