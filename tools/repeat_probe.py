"""Which tensors of an iteration are not bitwise repeatable?  Runs the same seeded iteration R times and lists, per run, the state_dict
entries that differ from run 0 after iteration 1 (a gradient that differs moves its parameter; the differing set closest to the losses names
the layer whose backward is not repeatable).  Usage: python tools/repeat_probe.py [config] [B] [precision] [R] [side streams 1|0] [iterations]"""
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, trainer
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.rng import PhiloxRng

name = sys.argv[1] if len(sys.argv) > 1 else "surreal-depth1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = sys.argv[3] if len(sys.argv) > 3 else "f32x6"
R = int(sys.argv[4]) if len(sys.argv) > 4 else 6
LANES = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
ITERS = int(sys.argv[6]) if len(sys.argv) > 6 else 2
ONLY = [a.split("=")[1].split(",") for a in sys.argv if a.startswith("only=")]      # only=idis,vdis: `precision` on those modules, fp32 elsewhere
native.lib()
if mode == "bf16cl":          # the bf16 channels-last data path (ops_cl) instead of an MFMA precision of the fp32 tensors
    from dcvgan_amd import ops_cl
    ops_cl.enable(True)
else:
    native.set_precision("fp32" if ONLY else mode)
if "record=1" in sys.argv:       # experiment: every gradient entering a custom backward is recorded on the stream that will read it
    import inspect
    from dcvgan_amd import ops as _ops
    for _n, _cls in inspect.getmembers(_ops, inspect.isclass):
        if issubclass(_cls, torch.autograd.Function) and _cls is not torch.autograd.Function and "backward" in _cls.__dict__:
            def _wrapped(ctx, *grads, _o=_cls.backward):
                cur = torch.cuda.current_stream()
                for g_ in grads:
                    if torch.is_tensor(g_) and g_.is_cuda:
                        g_.record_stream(cur)
                return _o(ctx, *grads)
            _cls.backward = staticmethod(_wrapped)
dev = torch.device("cuda:0")
cfg = CONFIGS[name].scaled(batchsize=B)
for a in sys.argv:            # width.gdis=8: thinner (faster) module, to move the streams against one another without changing a kernel
    if a.startswith("width."):
        cfg = __import__("dataclasses").replace(cfg, width={**cfg.width, a[6:].split("=")[0]: int(a.split("=")[1])})
g = torch.Generator().manual_seed(3)
xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)


def run():
    torch.manual_seed(11)
    models = trainer.build_models(cfg, dev)
    r = PhiloxRng(5)
    for m in models.values():
        m._rng = r
    if "handoff=1" in sys.argv:      # experiment: gradients leaving a discriminator for the generators are recorded on the main stream before they are handed to autograd
        main_stream = torch.cuda.current_stream()

        class _Handoff(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                return x.view_as(x)

            @staticmethod
            def backward(ctx, dy):
                dy.record_stream(main_stream)
                return dy

        for k in ("idis", "vdis", "gdis"):
            def fwd(xg_, xc_, _o=models[k].forward):
                return _o(_Handoff.apply(xg_) if xg_.requires_grad else xg_, _Handoff.apply(xc_) if xc_.requires_grad else xc_)
            models[k].forward = fwd
    if ONLY:
        from dcvgan_amd import util
        for k in ONLY[0]:
            util.set_precision(models[k], mode)
    runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg), sync_losses=True, side_streams=LANES)
    loss, snap = [], {}
    for it in range(ITERS):
        loss.append(runner.step(xc, xg, 2 + it))
        snap.update({f"it{it}.{k}.{n}": v.detach().float().cpu().clone() for k, m in models.items() for n, v in m.state_dict().items()})
    return loss, snap


def signature(p):
    return hash(tuple(float(v.double().sum()) for v in p.values()) + tuple(float(v.double().abs().sum()) for v in p.values()))


l0, p0 = run()
print("run 0:", l0)
sigs = {signature(p0): 1}
for i in range(1, R):
    l, p = run()
    sigs[signature(p)] = sigs.get(signature(p), 0) + 1
    bad = [(k, float((p[k] - p0[k]).abs().max()), float(p0[k].abs().max())) for k in p0 if not torch.equal(p[k], p0[k])]
    print(f"run {i}: losses equal {l == l0}; {len(bad)} of {len(p0)} tensors differ" + ("" if l == l0 else f"  {l} vs {l0}"))
    groups = {}
    for k, d, m in bad:
        groups.setdefault(".".join(k.split(".")[:2]), []).append(d)
    for gk, ds in groups.items():
        print(f"    {gk:12s} {len(ds):3d} tensors differ, largest max|diff| {max(ds):.3e}")
    if "-v" in sys.argv:
        for k, d, m in bad:
            print(f"        {k:50s} max|diff| {d:.3e}  max|value| {m:.3e}")
print(f"SUMMARY {mode}: {R} runs, {len(sigs)} distinct results, the most common one {max(sigs.values())} times ({R - max(sigs.values())} runs off it)")
