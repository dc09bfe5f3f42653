import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import dcvgan_oracle as O
from tests import goldenio as G
from dcvgan_amd import trainer
from dcvgan_amd.rng import InjectedRng
dev = torch.device("cuda:0")
fx = G.load("step_depth_adv_g1.npz")
cfg = G.cfg_of(fx, loss=str(fx["meta/loss"]), num_gen_update=1, num_dis_update=1)
B = cfg.batchsize
gd = torch.Generator().manual_seed(int(fx["meta/seed_data"]))
xc = torch.rand(B,3,16,64,64,generator=gd)*2-1; xg = torch.rand(B,1,16,64,64,generator=gd)*2-1
torch.manual_seed(int(fx["meta/seed_run"]))
so = O.StepOracle(cfg, G.states(fx))
for o in so.opt.values(): o.step = lambda: None
so.step(xc, xg, 3)
models = trainer.build_models(cfg, dev)
for n,m in models.items(): m.load_state_dict({k:v.detach().clone() for k,v in G.states(fx)[n].items()}); m.to(dev)
r = InjectedRng(so.rng.log)
for m in models.values(): m._rng = r
opts = trainer.build_optimizers(cfg, models)
for o in opts.values(): o.step = lambda: None
runner = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=True)
print(runner.step(xc.to(dev), xg.to(dev), 3))
rows=[]
for n in G.MODELS:
    pd = dict(models[n].named_parameters())
    for k,p in so.st[n].items():
        if k not in pd or p.grad is None: continue
        a = pd[k].grad.detach().cpu().double(); b = p.grad.double()
        relv = float((a-b).norm()/b.norm().clamp_min(1e-30))
        # elementwise: error relative to |b| for the Adam-relevant quantity
        ew = ((a-b).abs()/(b.abs()+1e-8)).max().item()
        rows.append((relv, ew, n, k, b.abs().min().item(), b.abs().max().item()))
rows.sort(reverse=True)
for r_ in rows[:25]: print("%.2e  ew=%.2e  %s %s  |g| in [%.2e, %.2e]" % r_)
