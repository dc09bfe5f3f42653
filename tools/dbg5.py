import sys; sys.path.insert(0,'.')
import numpy as np, torch
import torch.nn.functional as F
from oracle import dcvgan_oracle as O
from tests import goldenio as G
from dcvgan_amd import trainer, ops, layers
from dcvgan_amd.rng import InjectedRng
dev = torch.device("cuda:0")
fx = G.load("step_depth_adv_g1.npz")
cfg = G.cfg_of(fx, loss=str(fx["meta/loss"]), num_gen_update=1, num_dis_update=1); B = cfg.batchsize
gd = torch.Generator().manual_seed(int(fx["meta/seed_data"]))
xc = torch.rand(B,3,16,64,64,generator=gd)*2-1; xg = torch.rand(B,1,16,64,64,generator=gd)*2-1
cpu_log = []
_relu, _lrelu = F.relu, F.leaky_relu
def relu(x, *a, **k): cpu_log.append(x.detach().clone()); return _relu(x, *a, **k)
def lrelu(x, *a, **k): cpu_log.append(x.detach().clone()); return _lrelu(x, *a, **k)
O.F.relu, O.F.leaky_relu = relu, lrelu
torch.manual_seed(int(fx["meta/seed_run"]))
so = O.StepOracle(cfg, G.states(fx))
for o in so.opt.values(): o.step = lambda: None
so.step(xc, xg, 3)
hip_log = []
_bn, _conv = ops.bn_act, ops.conv
def bn_act(*a, **k):
    y = _bn(*a, **k); hip_log.append(y.detach()); return y
def conv(x, w, g, act=0, slope=0.0):
    y = _conv(x, w, g, act, slope)
    if act == ops.ACT_LEAKY: hip_log.append(y.detach())
    return y
ops.bn_act = bn_act; ops.conv = conv
models = trainer.build_models(cfg, dev)
for n,m in models.items(): m.load_state_dict({k:v.detach().clone() for k,v in G.states(fx)[n].items()}); m.to(dev)
r = InjectedRng(so.rng.log)
for m in models.values(): m._rng = r
opts = trainer.build_optimizers(cfg, models)
for o in opts.values(): o.step = lambda: None
runner = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=True)
runner.step(xc.to(dev), xg.to(dev), 3)
print(len(cpu_log), len(hip_log))
for i, (a, b) in enumerate(zip(cpu_log, hip_log)):
    b = b.cpu()
    if a.shape != b.shape: print(i, "shape mismatch", a.shape, b.shape); continue
    flips = ((a > 0) != (b > 0))
    if flips.any():
        idx = flips.nonzero()
        print("call", i, tuple(a.shape), "flips", int(flips.sum()), "cpu pre-act values", a[flips][:5].tolist(), "hip post-act", b[flips][:5].tolist())
