"""32 x 128 x 128 discriminator stress shape (SURVEY §8(d) "D5"): vdis + gdis forward/backward on flow clips
(Cg = 2), HIP-event timed.  Usage: python tools/stress_d.py [B] [fp32|bf16|f32x6|bf16cl|fp16cl]
(bf16cl / fp16cl = the 16-bit channels-last data path with bf16 / fp16 elements: BASELINE configs[4] names an fp16 MFMA run of this shape;
 they also print the largest pre-BatchNorm magnitude — fp16's largest finite value is 65504 — and how many parameter-gradient elements came out zero)"""
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import discriminator as D, native

native.lib()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MODE = sys.argv[2] if len(sys.argv) > 2 else "fp32"
CL = MODE in ("bf16cl", "fp16cl")
if CL:
    from dcvgan_amd import layers, ops_cl
    ops_cl.enable(True, half=MODE[:4])
else:
    native.set_precision(MODE)
torch.manual_seed(0)
vdis = D.VideoDiscriminator(2, 3, True, 0.2, 64).to(dev)
gdis = D.GradientDiscriminator(2, 3, False, 0.2, 32).to(dev)
xc = (torch.rand(B, 3, 32, 128, 128, device=dev) * 2 - 1).requires_grad_(True)
xg = (torch.rand(B, 2, 32, 128, 128, device=dev) - 0.5).requires_grad_(True)


def step():
    for m in (vdis, gdis):
        m.zero_grad()
    yv, yg = vdis(xg, xc), gdis(xg, xc)
    (yv.mean() + yg.mean()).backward()
    return yv, yg


if CL:
    layers.PREBN_TAP = []
yv, yg = step()
torch.cuda.synchronize()
if CL:
    peak = max(float(t) for t in layers.PREBN_TAP); layers.PREBN_TAP = None
    gs = [p.grad for m in (vdis, gdis) for p in m.parameters() if p.grad is not None]
    zero = sum(int((g == 0).sum()) for g in gs); tot = sum(g.numel() for g in gs)
    print(f"{MODE} B={B}: largest pre-BatchNorm magnitude {peak:.2f}; parameter-gradient elements exactly zero {zero} of {tot} ({zero / tot:.2%}); all finite {all(bool(torch.isfinite(g).all()) for g in gs)}")
print("shapes", tuple(yv.shape), tuple(yg.shape), "finite", bool(torch.isfinite(yv).all() and torch.isfinite(yg).all()))
for _ in range(8):      # the clocks of a freshly started process are not the ones it holds under load: warm up before timing (3 timed steps right after start read up to 4x high)
    step()
NT = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(NT):
    step()
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / NT
gf = 3 * (55.1 + 13.6) * B   # fwd + dgrad + wgrad, SURVEY §8(d)
print(f"{MODE} B={B}: {ms:.2f} ms per fwd+bwd, ~{gf / ms:.1f} TFLOP/s, {B / ms * 1e3:.1f} clips/s, peak mem {torch.cuda.max_memory_allocated() / 1e9:.2f} GB")
