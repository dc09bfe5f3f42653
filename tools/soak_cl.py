"""Race screen for the 16-bit path's schedule (lanes + the weight gradients' companion stream): N iterations of the bench workload from the same seeds, three times —
shipped schedule twice and the in-stream schedule once — and every iteration's losses plus the final parameters must be identical, bit for bit.
usage: python3 tools/soak_cl.py [config [iterations [batch]]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcvgan_amd import ops_cl, trainer
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.rng import PhiloxRng

name = sys.argv[1] if len(sys.argv) > 1 else "surreal-depth1"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = CONFIGS[name]
if len(sys.argv) > 3:
    cfg = cfg.scaled(batchsize=int(sys.argv[3]))
dev = torch.device("cuda:0")
B = cfg.batchsize
g = torch.Generator().manual_seed(4)
xc = (torch.rand(B, 3, cfg.video_length, 64, 64, generator=g) * 2 - 1).to(dev)
xg = (torch.rand(B, cfg.channel, cfg.video_length, 64, 64, generator=g) * 2 - 1).to(dev)


def run():
    torch.manual_seed(21)
    models = trainer.build_models(cfg, dev)
    r = PhiloxRng(13)
    for m in models.values():
        m._rng = r
    runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg), sync_losses=False)
    outs = []
    for i in range(iters):
        o = runner.step(xc, xg, i % cfg.video_length)
        outs.append(torch.stack([o[k].reshape(()) for k in sorted(o)]))
    torch.cuda.synchronize()
    return torch.stack(outs).cpu(), torch.cat([p.detach().reshape(-1) for m in models.values() for p in m.parameters()]).cpu()


ops_cl.enable(True)
a = run()
b = run()
ops_cl._WGRAD_SIDE = False
c = run()
ok1 = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
ok2 = torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
bad = (a[0] != b[0]).any(dim=1).nonzero().flatten().tolist()[:5], (a[0] != c[0]).any(dim=1).nonzero().flatten().tolist()[:5]
print(f"{name} B={B} {iters} iterations: shipped twice identical: {ok1}; shipped == in-stream: {ok2}; finite: {bool(torch.isfinite(a[0]).all())}; first differing iterations {bad}; last losses {a[0][-1].tolist()}")
sys.exit(0 if ok1 and ok2 else 1)
