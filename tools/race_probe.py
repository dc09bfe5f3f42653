"""Forward passes of the three discriminators: alone on one stream (the reference bits), then T trials of all three side by side on their own streams
with the generators' forward on the main stream (the step's D-phase schedule).  Every module output of a trial is compared bit for bit with the serial pass;
the first module whose output differs names the kernel that is sensitive to what else runs on the card.
Usage: python tools/race_probe.py [config] [B] [precision] [T]"""
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, trainer
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.rng import PhiloxRng

name = sys.argv[1] if len(sys.argv) > 1 else "surreal-depth1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = sys.argv[3] if len(sys.argv) > 3 else "f32x6"
T = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ONLY = [a.split("=")[1].split(",") for a in sys.argv if a.startswith("only=")]
BWD = "bwd" in sys.argv          # also the backward pass (gradients of every parameter and of the two inputs are compared)
native.lib()
native.set_precision("fp32" if ONLY else mode)
dev = torch.device("cuda:0")
cfg = CONFIGS[name].scaled(batchsize=B)
g = torch.Generator().manual_seed(3)
xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)
torch.manual_seed(11)
models = trainer.build_models(cfg, dev)
rng = PhiloxRng(5)
for m in models.values():
    m._rng = rng
    m.train()
if ONLY:
    from dcvgan_amd import util
    for k in ONLY[0]:
        util.set_precision(models[k], mode)
USE = [a.split("=")[1].split(",") for a in sys.argv if a.startswith("lanes=")]
DNAMES = tuple(USE[0]) if USE else ("idis", "vdis", "gdis")
NOG = "nog" in sys.argv           # no generator forward on the main stream beside the lanes
dis = [models[k] for k in DNAMES]
rec = {}


def hook(tag):
    def f(mod, inp, out):
        if torch.is_tensor(out):
            rec[tag] = out.detach().clone()
    return f


for k in ("idis", "vdis", "gdis"):
    for n, mod in models[k].named_modules():
        if not list(mod.children()):
            mod.register_forward_hook(hook(f"{k}.{n}:{type(mod).__name__}"))


class _Tap(torch.autograd.Function):      # identity on a discriminator's input: its backward sees that module's own contribution, before autograd sums them
    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        rec["contribution " + ctx.tag] = dy.detach().clone()
        return dy, None


def call(d):
    k = [n for n in DNAMES if models[n] is d][0]
    a, b = (_Tap.apply(xg, k + " xg"), _Tap.apply(xc, k + " xc")) if (BWD and "tap" in sys.argv) else (xg, xc)
    return d(a[:, :, 2], b[:, :, 2]) if d is models["idis"] else d(a, b)


if BWD:
    xg.requires_grad_(True); xc.requires_grad_(True)


def grads():
    out = {f"{k}.{n}": p.grad.detach().clone() for k in DNAMES for n, p in models[k].named_parameters() if p.grad is not None}
    out["input xg"] = xg.grad.detach().clone(); out["input xc"] = xc.grad.detach().clone()
    for k in DNAMES:
        models[k].zero_grad()
    xg.grad = None; xc.grad = None
    return out


with torch.set_grad_enabled(BWD):
    state = (rng._seed_seen, rng._counter)      # every pass draws the same noise
    ys = [call(d) for d in dis]
    if BWD:
        sum(y.float().sum() for y in ys).backward()
        rec.update({"grad " + k: v for k, v in grads().items()})
    torch.cuda.synchronize()
    ref = dict(rec)
    lanes = [torch.cuda.Stream(dev) for _ in range(3)]
    main = torch.cuda.current_stream()
    bad_total = {}
    for t in range(T):
        rec.clear()
        rng._seed_seen, rng._counter = state
        ys = []
        for lane, d in zip(lanes, dis):
            lane.wait_stream(main)
            with torch.cuda.stream(lane):
                ys.append(call(d))
        if not NOG:
            with torch.no_grad():
                v = models["ggen"].sample_videos(B)
                models["cgen"].forward_videos(v)
        if BWD:
            for lane, y in zip(lanes, ys):
                main.wait_stream(lane); y.record_stream(main)
            sum(y.float().sum() for y in ys).backward()
            rec.update({"grad " + k: v_ for k, v_ in grads().items()})
        torch.cuda.synchronize()
        bad = [k for k in ref if k in rec and not torch.equal(rec[k], ref[k])]
        if bad:
            first = {}
            for k in bad:
                first.setdefault(k.split(".")[0] if not k.startswith("contribution") else k, k)
            if "-q" in sys.argv:
                for k in first.values():
                    bad_total[k] = bad_total.get(k, 0) + 1
                continue
            if "anatomy" in sys.argv:
                k = list(first.values())[0]
                a, b = rec[k].float().contiguous(), ref[k].float().contiguous()
                idx = (a != b).nonzero()
                flat = (a != b).reshape(-1).nonzero().reshape(-1).tolist()
                print(f"  {k}: shape {tuple(a.shape)}; distinct coordinates per dimension {[int(idx[:, j].unique().numel()) for j in range(idx.shape[1])]}; ranges {[(int(idx[:, j].min()), int(idx[:, j].max())) for j in range(idx.shape[1])]}")
                runs, start, prev = [], flat[0], flat[0]
                for f in flat[1:]:
                    if f != prev + 1:
                        runs.append((start, prev - start + 1)); start = f
                    prev = f
                runs.append((start, prev - start + 1))
                import collections
                print(f"  {len(runs)} runs; run lengths {dict(collections.Counter(l for _, l in runs))}; first {runs[:12]}")
                for j in range(idx.shape[1]):
                    u = idx[:, j].unique().tolist()
                    print(f"  dim {j}: {u[:40]}{' ...' if len(u) > 40 else ''}")
                other = {kk: vv for kk, vv in ref.items() if kk.startswith("contribution") and kk != k and tuple(vv.shape) == tuple(a.shape)}
                for c in idx[:10].tolist():
                    t_ = tuple(c)
                    extra = "".join(f"  [{kk}: {float(vv[t_]):+.6e}]" for kk, vv in other.items())
                    nb = ""
                    if len(t_) == 5 and t_[2] > 0:
                        nb = f"  [same tensor, frame-1: got {float(a[t_[0], t_[1], t_[2] - 1, t_[3], t_[4]]):+.6e} exp {float(b[t_[0], t_[1], t_[2] - 1, t_[3], t_[4]]):+.6e}]"
                    print(f"  at {c}: got {float(a[t_]):+.6e} expected {float(b[t_]):+.6e} diff {float(a[t_] - b[t_]):+.3e}{extra}{nb}")
            for k in first.values():
                d_ = (rec[k] - ref[k]).abs()
                print(f"trial {t}: first differing output {k}: {int((d_ > 0).sum())} of {d_.numel()} elements, max|diff| {float(d_.max()):.3e}, max|value| {float(ref[k].abs().max()):.3e}")
                bad_total[k] = bad_total.get(k, 0) + 1
print(f"{mode}: {T} trials;", "all outputs bit-equal to the serial pass" if not bad_total else f"modules that differed first: {bad_total}")
