"""GPU: the bf16-product mode of the large GEMM kernels (dcv_set_precision(1); BASELINE configs[2] / [4] name bf16 / fp16
MFMA variants — the reference itself is fp32-only, so this is a throughput mode with its own, looser, tolerance).
Tensors, weights, BatchNorm statistics and optimiser state stay fp32; only the MFMA fragments are rounded to bf16
(8 significant bits, RNE), products are exact and accumulated in fp32.  Expected relative L2 error of a convolution:
~2^-8 / sqrt(3) * sqrt(2) = 3e-3 (two rounded operands); asserted < 1e-2 — and > 2e-4, which proves the bf16 kernels ran."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16(request):
    from dcvgan_amd import native
    native.lib()
    native.set_precision("bf16")
    yield torch.device("cuda:0")
    native.set_precision("fp32")


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# name, transposed, dims, Cin, Cout, k, s, p, input spatial, N  — geometries that take the LDS-DMA kernels
CASES = [
    ("conv2d_4s2p1_32_oc40", False, 2, 8, 40, 4, 2, 1, (32, 32), 3),
    ("conv2d_4s2p1_32_oc130", False, 2, 16, 130, 4, 2, 1, (32, 32), 2),
    ("conv2d_4s2p1_8_oc72", False, 2, 20, 72, 4, 2, 1, (8, 8), 7),
    ("convT2d_4s2p1_16_oc36", True, 2, 12, 36, 4, 2, 1, (16, 16), 3),
    ("convT2d_4s2p1_8_oc132", True, 2, 16, 132, 4, 2, 1, (8, 8), 5),
    ("conv3d_4s122_16_oc70", False, 3, 8, 70, 4, (1, 2, 2), (0, 1, 1), (6, 16, 16), 2),
    ("conv2d_4s2p1_wgrad_dma", False, 2, 16, 128, 4, 2, 1, (16, 16), 9),
    ("convT2d_4s2p1_wgrad_dma", True, 2, 128, 8, 4, 2, 1, (8, 8), 21),
    ("conv3d_4s122_wgrad_dma", False, 3, 8, 128, 4, (1, 2, 2), (0, 1, 1), (6, 16, 16), 3),
    ("conv2d_4s2p1_wgrad_dma64", False, 2, 16, 64, 4, 2, 1, (16, 16), 9),
    ("conv3d_dstep_128", False, 3, 64, 128, 4, (1, 2, 2), (0, 1, 1), (7, 16, 16), 4),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_bf16_products(bf16, case):
    from dcvgan_amd import native, ops
    dev = bf16
    name, tr, nd, cin, cout, k, s, p, sp, n = case
    g = torch.Generator().manual_seed(hash(name) % 10000)
    s_t = (s,) * nd if isinstance(s, int) else s
    p_t = (p,) * nd if isinstance(p, int) else p
    w = (torch.randn(((cin, cout) if tr else (cout, cin)) + (k,) * nd, generator=g) * 0.2).requires_grad_(True)
    x = torch.randn((n, cin) + sp, generator=g).requires_grad_(True)
    fn = {(False, 2): F.conv2d, (False, 3): F.conv3d, (True, 2): F.conv_transpose2d}[(tr, nd)]
    y_ref = fn(x, w, None, s_t, p_t)
    cot = torch.randn(y_ref.shape, generator=g)
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot).sum(), [x, w])
    xd, wd = x.detach().to(dev).requires_grad_(True), w.detach().to(dev).requires_grad_(True)
    y = ops.conv(xd, wd, ops.conv_geom(wd, s_t, p_t, tr))
    kernel = native.lib().dcv_debug_last_kernel().decode()
    gx, gw = torch.autograd.grad((y * cot.to(dev)).sum(), [xd, wd])
    errs = [rel(y, y_ref), rel(gx, gx_ref), rel(gw, gw_ref)]
    assert max(errs) < 1e-2, (name, errs)
    if "dma" in kernel:
        assert "bf16" in kernel and errs[0] > 2e-4, (kernel, errs)   # the bf16 instance ran (an fp32 one would sit at ~1e-7)


@pytest.mark.parametrize("name", ["isogd-depth", "surreal-depth1", "isogd-flow"])
def test_training_iteration_in_bf16_mode(bf16, name):
    """One full-width iteration at B = 4 of each GPU config (surreal-depth1 is the one BASELINE configs[2] names for bf16; isogd-flow the
    one configs[4] names for the 16-bit MFMA path): finite losses, parameters move, forward within 3e-2 of the fp32 mode."""
    from dcvgan_amd import native, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    dev = bf16
    cfg = CONFIGS[name].scaled(batchsize=4, num_gen_update=1)
    torch.manual_seed(5)
    models = trainer.build_models(cfg, dev)

    def fakes():
        r = PhiloxRng(77)
        for m in models.values():
            m._rng = r
        with torch.no_grad():
            xg = models["ggen"].sample_videos(4)
            return xg, models["cgen"].forward_videos(xg)

    xg_b, xc_b = fakes()
    native.set_precision("fp32")
    xg_f, xc_f = fakes()
    native.set_precision("bf16")
    assert 1e-4 < rel(xc_b, xc_f) < 3e-2 and rel(xg_b, xg_f) < 3e-2
    before = torch.cat([p.detach().reshape(-1) for p in models["cgen"].parameters()]).clone()
    runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg), sync_losses=True)
    g = torch.Generator().manual_seed(1)
    xc = (torch.rand(4, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(4, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)
    out = runner.step(xc, xg, 3)
    assert all(v == v and abs(v) < 100 for v in out.values()), out
    after = torch.cat([p.detach().reshape(-1) for p in models["cgen"].parameters()])
    assert float((after != before).float().mean()) > 0.9


def test_precision_is_a_per_module_switch():
    """dcv_conv_geom.mfma / util.set_precision(module, ...): one module's convolutions run bf16 products while the process default stays fp32 — and a module pinned to
    "fp32" stays fp32 under a bf16 process default.  The kernel instance names say which ran; the packed-weight cache keeps the two formats apart."""
    from dcvgan_amd import layers, native, ops, util
    native.lib()
    dev = torch.device("cuda:0")
    assert native.lib().dcv_get_precision() == 0
    last = lambda: native.lib().dcv_debug_last_kernel().decode()
    g = torch.Generator().manual_seed(3)
    a = torch.nn.Conv2d(16, 128, 4, 2, 1, bias=False).to(dev); b = torch.nn.Conv2d(16, 128, 4, 2, 1, bias=False).to(dev)
    with torch.no_grad():
        b.weight.copy_(a.weight)
    x = torch.randn(9, 16, 16, 16, generator=g).to(dev)
    util.set_precision(a, "bf16")
    assert layers.geom_of(a).mfma == 2 and layers.geom_of(b).mfma == 0
    xa = x.clone().requires_grad_(True); xb = x.clone().requires_grad_(True)
    ya = ops.conv(xa, a.weight, layers.geom_of(a)); ka = last()
    yb = ops.conv(xb, b.weight, layers.geom_of(b)); kb = last()
    assert "bf16" in ka and "bf16" not in kb, (ka, kb)
    e = rel(ya, yb)
    assert 2e-4 < e < 1e-2, e                                   # same weights and input: the two results differ by bf16 rounding, no more
    cot = torch.randn(ya.shape, generator=g).to(dev)
    ga = torch.autograd.grad((ya * cot).sum(), [xa, a.weight])
    gb = torch.autograd.grad((yb * cot).sum(), [xb, b.weight])
    # data and weight gradients follow the module's switch too (they run on autograd's thread, whose kernel-name record this thread cannot read: the
    # bf16 rounding of their results is the evidence — the fp32 instances would agree to ~1e-7)
    assert 2e-4 < rel(ga[0], gb[0]) < 1e-2 and 2e-4 < rel(ga[1], gb[1]) < 1e-2
    # the other way round: process default bf16, module pinned to fp32
    native.set_precision("bf16")
    try:
        util.set_precision(a, "fp32")
        with torch.no_grad():
            y1 = ops.conv(x, a.weight, layers.geom_of(a)); k1 = last()
            y2 = ops.conv(x, b.weight, layers.geom_of(b)); k2 = last()
        assert "bf16" not in k1 and "bf16" in k2, (k1, k2)
        assert rel(y1, yb) < 1e-6                                # the pinned module reproduces the fp32 result (its fp32 pack was not confused with the bf16 one)
        util.set_precision(a, None)
        assert layers.geom_of(a).mfma == 0
    finally:
        native.set_precision("fp32")
