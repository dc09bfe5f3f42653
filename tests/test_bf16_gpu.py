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


# --------------------------------------------------------------------------- #
# fp32 emulated on the bf16 matrix pipe ("f32x6", dcv_set_precision(2); round 4, experimental — never the default)
# --------------------------------------------------------------------------- #
@pytest.fixture()
def f32x6(request):
    from dcvgan_amd import native
    native.lib()
    native.set_precision("f32x6")
    yield torch.device("cuda:0")
    native.set_precision("fp32")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_f32x6(f32x6, case):
    """Every operand is split EXACTLY into three bf16 pieces and the six products of total order <= 2 are accumulated in fp32: the results are
    fp32-grade — held to 2e-6 relative L2 against an fp64 evaluation (measured ~2e-7, the native fp32 MFMA kernels' own level; the bf16-product
    mode sits at 3e-3) for forward, data gradient and weight gradient, on every geometry that takes the LDS-DMA kernels."""
    from dcvgan_amd import native, ops
    dev = f32x6
    name, tr, nd, cin, cout, k, s, p, sp, n = case
    g = torch.Generator().manual_seed(hash(name) % 10000)
    s_t = (s,) * nd if isinstance(s, int) else s
    p_t = (p,) * nd if isinstance(p, int) else p
    w = (torch.randn(((cin, cout) if tr else (cout, cin)) + (k,) * nd, generator=g) * 0.2).requires_grad_(True)
    x = torch.randn((n, cin) + sp, generator=g).requires_grad_(True)
    fn = {(False, 2): F.conv2d, (False, 3): F.conv3d, (True, 2): F.conv_transpose2d}[(tr, nd)]
    x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    y_ref = fn(x64, w64, None, s_t, p_t)
    cot = torch.randn(y_ref.shape, generator=g)
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot.double()).sum(), [x64, w64])
    xd, wd = x.detach().to(dev).requires_grad_(True), w.detach().to(dev).requires_grad_(True)
    y = ops.conv(xd, wd, ops.conv_geom(wd, s_t, p_t, tr))
    kernel = native.lib().dcv_debug_last_kernel().decode()
    gx, gw = torch.autograd.grad((y * cot.to(dev)).sum(), [xd, wd])
    errs = [rel(y, y_ref), rel(gx, gx_ref), rel(gw, gw_ref)]
    assert max(errs) < 2e-6, (name, kernel, errs)
    if "dma" in kernel:
        assert "f32x6" in kernel, kernel


def test_pack_precision_stamp():
    """A caller-owned packed-weight buffer is stamped with the precision it was packed for (dcv_wpack.precision): the same weight tensor run at fp32,
    then f32x6, then fp32 again gives the fp32 result both times (the cache keeps one pack per precision), and a ready pack handed to a call of
    another precision is refused by the library itself (DCV_EINVAL), not read."""
    import ctypes as C
    from dcvgan_amd import native, ops
    from dcvgan_amd.native import WPack, check, dims5, lib, ptr, stream_ptr
    dev = torch.device("cuda:0")
    g0 = torch.Generator().manual_seed(11)
    w = (torch.randn(128, 16, 4, 4, generator=g0) * 0.1).to(dev)
    x = torch.randn(9, 16, 16, 16, generator=g0).to(dev)
    with torch.no_grad():
        y1 = ops.conv(x, w, ops.conv_geom(w, (2, 2), (1, 1), False))
        native.set_precision("f32x6")
        try:
            y2 = ops.conv(x, w, ops.conv_geom(w, (2, 2), (1, 1), False))
        finally:
            native.set_precision("fp32")
        y3 = ops.conv(x, w, ops.conv_geom(w, (2, 2), (1, 1), False))
    assert torch.equal(y1, y3) and rel(y2, y1) < 1e-6
    L = lib()
    g = ops.conv_geom(w, (2, 2), (1, 1), False)
    y = torch.empty_like(y1)
    xd, yd = dims5(x), dims5(y)
    nb = L.dcv_conv_packed_bytes(C.byref(g), C.byref(xd), C.byref(yd), 0)
    buf = torch.empty(nb, dtype=torch.uint8, device=dev)
    ws = torch.empty(L.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 0) + 256, dtype=torch.uint8, device=dev)
    pk = WPack(buf.data_ptr(), nb, 0, 1)
    check(L.dcv_conv_forward(C.byref(g), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), 0, 0.0, C.byref(pk), ptr(ws), ws.numel(), stream_ptr()), "pack")
    assert torch.equal(y, y1)
    pk.ready = 1
    g_x6 = ops.conv_geom(w, (2, 2), (1, 1), False, "f32x6")
    n0 = native.launch_count()
    rc = L.dcv_conv_forward(C.byref(g_x6), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), 0, 0.0, C.byref(pk), ptr(ws), ws.numel(), stream_ptr())
    assert rc == -1 and b"dcv_wpack.precision" in L.dcv_last_error() and native.launch_count() == n0


def test_f32x6_error_is_not_one_sided(f32x6):
    """The bf16 MFMA's accumulate drops what it drops toward minus infinity; the emulation's six small-addend MFMAs per 16 k made that a visible bias of the OUTPUT
    (mean error ~400 standard errors below zero at K = 8192; per-channel sums of an output 20x further from fp64 than the native kernels': profiles/r04_f32x6_bias.txt).
    The kernels now alternate the sign of what they accumulate in phases of 8 K steps (X6_PHASE, csrc/conv_mfma.hip).  Forward and data gradient of the video
    discriminator's conv3d 128 -> 256 (K = 8192) against torch's fp64: the mean signed error within 40 standard errors of zero, channel sums to 2e-6."""
    from dcvgan_amd import ops
    dev = f32x6
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 128, 10, 16, 16, generator=g); w = torch.randn(256, 128, 4, 4, 4, generator=g) * 0.05
    with torch.backends.cudnn.flags(enabled=False):        # torch's own fp64 convolution on the device
        xr = x.double().to(dev).requires_grad_(True)
        y64 = F.conv3d(xr, w.double().to(dev), None, (1, 2, 2), (0, 1, 1))
        dy = torch.randn(y64.shape, generator=g)
        (dx64,) = torch.autograd.grad(y64, [xr], dy.double().to(dev))
    xd = x.to(dev).requires_grad_(True)
    y = ops.conv(xd, w.to(dev), ops.conv_geom(w.to(dev), (1, 2, 2), (0, 1, 1), False))
    (dx,) = torch.autograd.grad(y, [xd], dy.to(dev))
    for name, a, b in (("forward", y.detach(), y64.detach()), ("data gradient", dx, dx64)):
        e = a.double() - b
        bias = float(e.mean() / e.std() * e.numel() ** 0.5)
        dims = (0, 2, 3, 4)
        sums = float((a.double().sum(dims) - b.sum(dims)).norm() / b.sum(dims).norm())
        assert float(e.norm() / b.norm()) < 2e-6, name
        assert abs(bias) < 40 and sums < 2e-6, (name, bias, sums)
