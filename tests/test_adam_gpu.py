"""GPU: dcv_adam_step / dcv_adam_step_multi against torch.optim.Adam with the reference's wiring
(train.py:171-176: betas (0.5, 0.999), eps 1e-8, L2 weight decay 1e-5) — parameters and both moment
buffers, element by element, over 10 steps.  Covers > 24 tensors (the multi-tensor kernel's chunking),
ragged sizes around its 4096-element blocks, a tensor whose first gradient arrives at step 3 (own bias
correction: the single-tensor path), grad_scale = 1/8 (the data-parallel 1/world factor) and the trainer's
double ggen step (two steps on the same gradients, trainer.py:357-359)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SIZES = [(1,), (3, 5), (64,), (255,), (256,), (257,), (4095,), (4096,), (4097,), (8192,), (12289,), (128, 64, 4, 4),
         (7, 11, 13), (2, 2), (30, 10), (30,), (512, 50, 4, 4), (64, 1, 3, 3), (1, 256, 4, 4), (33,), (65,), (129,), (1000,), (4000,),
         (5000,), (9, 9, 9), (31, 31), (100, 100), (17,), (3,)]


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("grad_scale,double_step", [(1.0, False), (0.125, False), (1.0, True)], ids=["plain", "grad_scale_1_8", "double_step"])
def test_adam_matches_torch(grad_scale, double_step):
    from dcvgan_amd import optim
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7)
    assert len(SIZES) > 24
    ref = [torch.nn.Parameter(torch.randn(s, generator=gen) * 0.05) for s in SIZES]
    hip = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    late = 4   # this tensor gets its first gradient at step 3
    topt = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, weight_decay=1e-5)
    hopt = optim.Adam(hip, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, weight_decay=1e-5)
    hopt.grad_scale = grad_scale
    for step in range(1, 11):
        for i, (p, q) in enumerate(zip(ref, hip)):
            if i == late and step < 3:
                p.grad = None; q.grad = None
                continue
            # magnitudes from 1e-6 to 1: exercises eps against sqrt(v)
            g = torch.randn(p.shape, generator=gen) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=gen)))
            p.grad = g.clone()
            q.grad = (g / grad_scale).to(dev)       # the HIP side sees the un-averaged sum
        topt.step(); hopt.step()
        if double_step:
            topt.step(); hopt.step()
        torch.cuda.synchronize()
        for i, (p, q) in enumerate(zip(ref, hip)):
            if p.grad is None:
                assert torch.equal(p.detach(), q.detach().cpu())
                continue
            st, sh = topt.state[p], hopt.state[q]
            assert int(st["step"]) == sh["step"]
            # (g / grad_scale) * grad_scale is exact for a power of two, so the bars stay at 1e-6
            assert _rel(q.detach(), p.detach()) <= 1e-6, (step, i, "p")
            assert _rel(sh["exp_avg"], st["exp_avg"]) <= 1e-6, (step, i, "m")
            assert _rel(sh["exp_avg_sq"], st["exp_avg_sq"]) <= 1e-6, (step, i, "v")
    # total movement after 10 steps: relative L2 of (theta - theta_0)
    gen0 = torch.Generator().manual_seed(7)
    for i, s in enumerate(SIZES):
        t0 = torch.randn(s, generator=gen0) * 0.05
        d_ref = (ref[i].detach() - t0).double()
        d_hip = (hip[i].detach().cpu() - t0).double()
        assert float((d_hip - d_ref).norm() / d_ref.norm().clamp_min(1e-30)) <= 1e-4, i


def test_adam_skips_missing_grads_and_validates():
    from dcvgan_amd import native, optim
    dev = torch.device("cuda:0")
    a = torch.nn.Parameter(torch.ones(10, device=dev)); b = torch.nn.Parameter(torch.ones(10, device=dev))
    o = optim.Adam([a, b], lr=1e-2, betas=(0.5, 0.999))
    a.grad = torch.ones(10, device=dev)
    o.step()
    assert torch.equal(b.detach().cpu(), torch.ones(10)) and float(a.detach().sum()) < 10.0
    with pytest.raises(native.NativeError):
        c = torch.nn.Parameter(torch.ones(4)); c.grad = torch.ones(4)
        optim.Adam([c]).step()      # CPU tensors: no fallback
