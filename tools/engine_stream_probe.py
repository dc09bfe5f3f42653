"""Pure PyTorch (no kernel of this repository): does autograd's sum of gradients that arrive from two side streams equal the single-stream sum, bit for bit?
A leaf x feeds two branches, each run on its own stream (one long, one short); x.grad after backward is compared with the same graph run on one stream.
Usage: python tools/engine_stream_probe.py [trials] [leaf|nonleaf]"""
import sys
import torch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
LEAF = (sys.argv[2] if len(sys.argv) > 2 else "leaf") == "leaf"
dev = torch.device("cuda:0")
torch.manual_seed(0)
x0 = torch.randn(16, 1, 16, 64, 64, device=dev)
w1 = [torch.randn(32, 1, 4, 4, 4, device=dev) * 0.1, torch.randn(64, 32, 4, 4, 4, device=dev) * 0.05]
w2 = [torch.randn(8, 1, 4, 4, 4, device=dev) * 0.1]


def branch(x, ws):
    for w in ws:
        x = torch.nn.functional.leaky_relu(torch.nn.functional.conv3d(x, w, stride=(1, 2, 2), padding=(0, 1, 1)), 0.2)
    return x


def run(streams):
    x = x0.clone().requires_grad_(True)
    src = x if LEAF else x * 1.0
    main = torch.cuda.current_stream()
    ys = []
    for s, ws in zip(streams, (w1, w2)):
        if s is None:
            ys.append(branch(src, ws))
        else:
            s.wait_stream(main)
            with torch.cuda.stream(s):
                ys.append(branch(src, ws))
    for s, y in zip(streams, ys):
        if s is not None:
            main.wait_stream(s); y.record_stream(main)
    (ys[0].sum() + ys[1].sum()).backward()
    torch.cuda.synchronize()
    return x.grad.clone()


ref = run((None, None))
assert torch.equal(ref, run((None, None))), "the single-stream run is not repeatable itself"
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
bad = 0
for t in range(T):
    g = run((s1, s2))
    if not torch.equal(g, ref):
        d = (g - ref).abs()
        bad += 1
        if bad <= 5:
            print(f"trial {t}: {int((d > 0).sum())} of {d.numel()} elements differ, max|diff| {float(d.max()):.3e} (max|value| {float(ref.abs().max()):.3e})")
print(f"torch {torch.__version__}: {'leaf' if LEAF else 'non-leaf'} input, two side streams: {bad} of {T} trials differ from the single-stream gradient")
