"""Where does a multi-stream iteration first leave the single-stream bits?  Runs ITERS seeded iterations once on one stream (reference), then R times
with the discriminators on their side streams; every leaf-module output and every gradient arriving at such an output is cloned in a hook and compared
bit for bit after the run.  Prints, per run, the records that differ in execution order (forward records first, then backward in the order they fired).
Usage: python tools/race_trace.py [config] [B] [precision] [R] [iterations]"""
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, trainer
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.rng import PhiloxRng

name = sys.argv[1] if len(sys.argv) > 1 else "surreal-depth1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
R = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ITERS = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ONLY = [a.split("=")[1].split(",") for a in sys.argv if a.startswith("only=")]      # only=gdis: `precision` on those modules, fp32 elsewhere
native.lib()
native.set_precision("fp32" if ONLY else mode)
from dcvgan_amd import ops, util
dev = torch.device("cuda:0")
cfg = CONFIGS[name].scaled(batchsize=B)
g = torch.Generator().manual_seed(3)
xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)


def run(lanes):
    torch.manual_seed(11)
    models = trainer.build_models(cfg, dev)
    r = PhiloxRng(5)
    for m in models.values():
        m._rng = r
    rec, order, count = {}, [], {}

    def put(tag, t):
        n = count[tag] = count.get(tag, 0) + 1
        key = f"{tag}#{n}"
        rec[key] = t.detach().clone()
        order.append(key)

    def hook(tag):
        def f(mod, inp, out):
            if torch.is_tensor(out):
                put("fwd " + tag, out)
                if out.requires_grad:
                    out.register_hook(lambda gr, tag=tag: put("bwd " + tag, gr))
        return f

    if ONLY:
        for k in ONLY[0]:
            util.set_precision(models[k], mode)
    # the modules call the ops layer directly: record there (host order is the same on one stream and on several)
    saved = {}
    for opname in ("conv", "bn_act", "act", "noise_add", "temporal_diff", "cat_channels", "gan_loss", "gru_sequence", "copy_into", "softmax_channels"):
        orig = saved[opname] = getattr(ops, opname)

        def wrapped(*a, _orig=orig, _n=opname, **kw):
            out = _orig(*a, **kw)
            if torch.is_tensor(out):
                tag = _n + (" " + "x".join(map(str, out.shape)))
                put("fwd " + tag, out)
                if out.requires_grad:
                    out.register_hook(lambda gr, tag=tag: put("bwd " + tag, gr))
            return out
        setattr(ops, opname, wrapped)
    class _Tap(torch.autograd.Function):      # identity on a discriminator's input: its backward sees that discriminator's own contribution, before autograd sums the three
        @staticmethod
        def forward(ctx, x, tag):
            ctx.tag = tag
            return x.view_as(x)

        @staticmethod
        def backward(ctx, dy):
            put("bwd contribution " + ctx.tag, dy)
            return dy, None

    for k in ("idis", "vdis", "gdis"):
        def fwd(xg_, xc_, _o=models[k].forward, _k=k):
            return _o(_Tap.apply(xg_, _k + " xg") if xg_.requires_grad else xg_, _Tap.apply(xc_, _k + " xc") if xc_.requires_grad else xc_)
        models[k].forward = fwd
    runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg), sync_losses=True, side_streams=lanes)
    for it in range(ITERS):
        runner.step(xc, xg, 2 + it)
        order.append(f"--- end of iteration {it}")
        for k, m in models.items():
            for n, v in m.state_dict().items():
                put(f"state it{it} {k}.{n}", v)
    torch.cuda.synchronize()
    for k, v in saved.items():
        setattr(ops, k, v)
    return rec, order


ref, order0 = run(False)
print(f"{len(ref)} records per run")
for i in range(R):
    rec, order = run(True)
    bad = [k for k in order if k in ref and k in rec and not torch.equal(rec[k], ref[k])]
    missing = [k for k in ref if k not in rec]
    print(f"run {i}: {len(bad)} records differ, {len(missing)} missing")
    for k in bad[:12]:
        d = (rec[k].float() - ref[k].float()).abs()
        print(f"    {k:70s} {int((d > 0).sum()):9d} of {d.numel():9d} elements, max|diff| {float(d.max()):.3e} (max|value| {float(ref[k].float().abs().max()):.3e})")
    if bad:     # anatomy of the first differing record: where the changed elements sit says whose layout wrote them
        k = bad[0]
        a, b = rec[k].float().contiguous(), ref[k].float().contiguous()
        idx = (a != b).nonzero()
        flat = (a != b).reshape(-1).nonzero().reshape(-1)
        print(f"      shape {tuple(a.shape)} strides(rec) {tuple(rec[k].stride())}; per-dimension distinct coordinates: {[int(idx[:, j].unique().numel()) for j in range(idx.shape[1])]}")
        print(f"      coordinate ranges: {[(int(idx[:, j].min()), int(idx[:, j].max())) for j in range(idx.shape[1])]}")
        runs, start, prev = [], int(flat[0]), int(flat[0])
        for f in flat[1:].tolist():
            if f != prev + 1:
                runs.append((start, prev - start + 1)); start = f
            prev = f
        runs.append((start, prev - start + 1))
        print(f"      {len(runs)} runs of consecutive flat indices; first runs (start, length): {runs[:16]}")
        for c in idx[:6].tolist():
            print(f"      at {c}: got {float(a[tuple(c)]):+.6e}  expected {float(b[tuple(c)]):+.6e}")
