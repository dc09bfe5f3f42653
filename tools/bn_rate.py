"""BatchNorm passes of the bf16 channels-last path on one large layer (cgen.up5's output: 1600 x 64 x 64 x 64): ms and TB/s of forward apply, backward reduce + apply.
usage: python3 tools/bn_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcvgan_amd import ops, ops_cl, native
ops_cl.enable(True)
dev = torch.device("cuda:0")
for shape in ((1600, 64, 64, 64), (1600, 128, 16, 16), (100, 128, 13, 32, 32)):
    C = shape[1]
    x = ops_cl.cl_empty(shape, dev); x.copy_(torch.randn(shape, device=dev))
    x.requires_grad_(True)
    g = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dy = ops_cl.cl_empty(shape, dev); dy.copy_(torch.randn(shape, device=dev))
    def run():
        y = ops_cl.bn_act(x, g, b, rm, rv, True, ops.ACT_LEAKY, 0.2)
        y.backward(dy)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    nbytes = x.numel() // C * ops_cl.pitch_of(C) * 2
    print(shape, "fwd+bwd %.3f ms per call; tensor %.0f MB; 7 tensor passes -> %.2f TB/s" % (e0.elapsed_time(e1) / 10, nbytes / 1e6, 7 * nbytes / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12))
