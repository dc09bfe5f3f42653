"""Phase durations of the headline iteration (HIP events on the main stream at StepRunner's phase boundaries; the lanes join the main stream at every
boundary, so the durations add up to the iteration).   python tools/phases.py [config] [--steps 8]"""
import argparse
import collections
import sys

sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, trainer
from dcvgan_amd.configs import CONFIGS

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="isogd-depth")
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
native.lib()
cfg = CONFIGS[a.config]
dev = torch.device("cuda:0")
torch.manual_seed(cfg.seed)
models = trainer.build_models(cfg, dev)
runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg))
g = torch.Generator().manual_seed(1)
B = cfg.batchsize
xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)
for i in range(3):
    runner.step(xc, xg, i)
torch.cuda.synchronize()
runner.phase_marks = []
for i in range(a.steps):
    runner.step(xc, xg, 3 + i)
torch.cuda.synchronize()
tot = collections.OrderedDict()
marks = runner.phase_marks
for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
    key = n1 if n1 != "start" else "(between iterations)"
    tot[key] = tot.get(key, 0.0) + e0.elapsed_time(e1)
s = 0.0
for k, v in tot.items():
    print(f"{v / a.steps:8.3f} ms  {k}")
    s += v / a.steps
print(f"{s:8.3f} ms  sum")
