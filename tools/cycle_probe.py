"""Does an iteration leave reference cycles that hold device tensors?  (A cycle is freed only when Python's cyclic collector gets to it: until then its tensors stay
allocated — tools/soak.py's "device memory grows".)  Runs N iterations with the collector off, then collects with DEBUG_SAVEALL and lists what was only reachable
from cycles: object types, and every tensor among them with its size.  Usage: python3 tools/cycle_probe.py [B] [iterations] [fp32|bf16cl]"""
import collections
import gc
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, trainer
from dcvgan_amd.configs import CONFIGS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
MODE = sys.argv[3] if len(sys.argv) > 3 else "fp32"
native.lib()
if MODE == "bf16cl":
    from dcvgan_amd import ops_cl
    ops_cl.enable(True)
dev = torch.device("cuda:0")
cfg = CONFIGS["isogd-depth"].scaled(batchsize=B)
torch.manual_seed(1)
models = trainer.build_models(cfg, dev)
runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg))
g = torch.Generator().manual_seed(2)
xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)
runner.step(xc, xg, 0)
torch.cuda.synchronize()
gc.collect()
gc.disable()
base = torch.cuda.memory_allocated()
for i in range(N):
    runner.step(xc, xg, (i + 1) % 16)
torch.cuda.synchronize()
_ = torch.empty(1, device=dev)      # lets the allocator process the events of blocks freed on other streams
held = torch.cuda.memory_allocated() - base
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
kinds = collections.Counter(type(o).__name__ for o in gc.garbage)
tens = [o for o in gc.garbage if torch.is_tensor(o)]
print(f"{N} iterations with the collector off: {held / 1e6:.1f} MB more allocated than after the first; the collector then found {n} objects in cycles")
print("types:", dict(kinds.most_common(12)))
tot = 0
for t in sorted(tens, key=lambda t: -t.numel())[:12]:
    print("  tensor", tuple(t.shape), t.dtype, f"{t.numel() * t.element_size() / 1e6:.1f} MB", "grad_fn=" + type(t.grad_fn).__name__ if t.grad_fn is not None else "")
tot = sum(t.numel() * t.element_size() for t in tens if t.is_cuda)
print(f"device tensors only reachable from cycles: {len(tens)}, {tot / 1e6:.1f} MB (views counted at their own size)")
