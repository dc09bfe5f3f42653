"""Soak: N iterations of the isogd-depth step at B = 16 (or argv[2]; 70 = the bench batch, where the ragged split-K and every big-problem variant run); device and host memory must be flat after the first few iterations
(scratch buffers, packed-weight caches and the library's index-table cache are all bounded by the set of layer geometries).
Usage: python tools/soak.py [N] [B] [config] [fp32|bf16|f32x6|bf16cl]"""
import os
import resource
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, trainer
from dcvgan_amd.configs import CONFIGS

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
CFG = sys.argv[3] if len(sys.argv) > 3 else "isogd-depth"
MODE = sys.argv[4] if len(sys.argv) > 4 else "fp32"
native.lib()
if MODE == "bf16cl":
    from dcvgan_amd import ops_cl
    ops_cl.enable(True)
else:
    native.set_precision(MODE)
dev = torch.device("cuda:0")
cfg = CONFIGS[CFG].scaled(batchsize=B)
torch.manual_seed(1)
models = trainer.build_models(cfg, dev)
runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg))
g = torch.Generator().manual_seed(2)
xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)
marks = []
for i in range(N):
    out = runner.step(xc, xg, i % 16)
    if i in (5, 20, N // 2, N - 1) or (os.environ.get('SOAK_VERBOSE') and i % 20 in (0, 1)):
        torch.cuda.synchronize()
        torch.empty(1, device=dev)      # an allocation lets the allocator take back blocks that were freed on other streams
        marks.append((i, torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3,
                      {k: round(float(v), 4) for k, v in out.items()}))
for m in marks:
    print("iter %4d  allocated %.1f MB  reserved %.1f MB  host maxrss %.1f MB  %s" % m)
# (tensors that crossed streams are released an event later: +-1 MB at the sample point; iterations whose D update is gated off hold ~4 MB less at B = 16 than the others,
#  and the marks fall on both kinds)
assert abs(marks[-1][1] - marks[1][1]) < 8.0 * max(1, B // 16), "device memory grows"
assert marks[-1][3] - marks[1][3] < 64, "host memory grows"
assert all(v == v and abs(v) < 1e3 for v in marks[-1][4].values())
print(f"soak ok ({CFG}, {MODE}, B = {B}, {N} iterations)")
