"""Which calls of one G+D iteration synchronise the host with the GPU?  torch.cuda.set_sync_debug_mode("warn") around two warmed-up iterations; prints each distinct
warning site with its count, plus the host time of step() with and without a full queue ahead of it.
usage: python3 tools/sync_probe.py [config [precision]]"""
import os, sys, time, warnings, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcvgan_amd import trainer
from dcvgan_amd.configs import CONFIGS

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "surreal-depth1"]
if len(sys.argv) > 2 and sys.argv[2] == "bf16cl":
    from dcvgan_amd import ops_cl
    ops_cl.enable(True)
dev = torch.device("cuda:0")
torch.manual_seed(0)
models = trainer.build_models(cfg, dev)
opts = trainer.build_optimizers(cfg, models)
run = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg))
B = cfg.batchsize
xc = torch.rand(B, 3, cfg.video_length, 64, 64, device=dev) * 2 - 1
xg = torch.rand(B, cfg.channel, cfg.video_length, 64, 64, device=dev) * 2 - 1
for i in range(4):
    run.step(xc, xg, i)
torch.cuda.synchronize()
sites = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "dcvgan_amd" in f.filename or "tools/" in f.filename]
    sites[(str(message)[:80], " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-4:]))] += 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for i in range(2):
    run.step(xc, xg, i)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("synchronising calls in 2 iterations:", sum(sites.values()))
for (m, w), n in sites.most_common(): print("%3d x %s\n      %s" % (n, m, w))
# host time of step() per iteration, queue kept full (no synchronize between iterations)
t = []
t0 = time.perf_counter()
for i in range(8):
    a = time.perf_counter(); run.step(xc, xg, i); t.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("8 iterations back to back: %.1f ms each; host time of step(): %s ms" % (tot / 8 * 1e3, [round(x * 1e3, 1) for x in t]))
