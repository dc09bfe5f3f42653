"""GPU: the bf16 channels-last ("CL16") data path (dcvgan_amd/ops_cl.py, csrc/conv_cl16.hip, csrc/cl_elementwise.hip) — BASELINE configs[2] / [4] name
16-bit MFMA variants; the reference is fp32-only, so these are TOLERANCE tests of a throughput path, not parity tests.

Operands are rounded to bf16 first and the SAME rounded values go through torch's fp32 CPU ops, so what is measured is the kernels' own arithmetic: fp32
accumulation of exact bf16 products, then one rounding of the result to bf16 (relative 2^-9 per element: ~2.3e-3 relative L2) for activations and data
gradients — asserted < 5e-3 — and NO rounding for weight gradients (fp32 out): asserted < 2e-5.  Every case also runs inside NaN guard bands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def r16(t):
    return t.to(torch.bfloat16).float()


# name, transposed, dims, Cin, Cout, k, s, p, input spatial, N
CASES = [
    ("conv2d_4s2p1_64_128", False, 2, 64, 128, 4, 2, 1, (16, 16), 3),
    ("conv2d_4s2p1_32_40", False, 2, 32, 40, 4, 2, 1, (32, 32), 2),
    ("conv2d_4s2p1_96_192", False, 2, 96, 192, 4, 2, 1, (8, 8), 5),
    ("conv2d_4s2p1_thin3_32", False, 2, 3, 32, 4, 2, 1, (64, 64), 2),
    ("conv2d_4s2p1_256_1", False, 2, 256, 1, 4, 2, 1, (8, 8), 3),
    ("conv2d_3s1p1_thin1_64", False, 2, 1, 64, 3, 1, 1, (64, 64), 2),
    ("conv2d_3s1p1_thin2_64_n9", False, 2, 2, 64, 3, 1, 1, (32, 64), 9),     # fused thin-source kernel: two bands per image, images past a multiple of 8
    ("convT2d_4s2p1_128_64", True, 2, 128, 64, 4, 2, 1, (16, 16), 3),
    ("convT2d_4s2p1_64_96_w8_n5", True, 2, 64, 96, 4, 2, 1, (8, 8), 5),        # patch-staged kernel: two images per patch, an odd image count, 96 = 1.5 channel tiles
    ("convT2d_4s2p1_96_192_w4_n11", True, 2, 96, 192, 4, 2, 1, (4, 4), 11),    # ... eight 4 x 4 images per patch, three K blocks
    ("convT2d_4s2p1_32_64_w32", True, 2, 32, 64, 4, 2, 1, (32, 32), 2),        # ... four rows of a 32-wide image per patch, one K block
    ("convT2d_4s2p1_64_64_h8w16", True, 2, 64, 64, 4, 2, 1, (8, 16), 3),       # ... a rectangular image
    ("convT2d_4s2p1_64_64_h6w8", True, 2, 64, 64, 4, 2, 1, (6, 8), 2),         # a height the patch plan does not take: the tiled gather
    ("convT2d_4s2p1_288_256", True, 2, 266, 256, 4, 2, 1, (1, 1), 7),
    ("convT2d_4s1p0_latent", True, 2, 50, 128, 4, 1, 0, (1, 1), 9),
    ("convT2d_4s2p1_96_1", True, 2, 96, 1, 4, 2, 1, (32, 32), 2),
    ("convT2d_3s1p1_128_3", True, 2, 128, 3, 3, 1, 1, (64, 64), 2),
    ("convT2d_3s1p1_128_3_n9", True, 2, 128, 3, 3, 1, 1, (64, 64), 9),       # fused thin-destination kernel: images past a multiple of 8 (XCD slots without an image)
    ("convT2d_3s1p1_32_2", True, 2, 32, 2, 3, 1, 1, (64, 64), 3),            # its 32-channel instance
    ("convT2d_3s1p1_64_1_h32", True, 2, 64, 1, 3, 1, 1, (32, 64), 2),        # its 64-channel instance, two bands per image
    ("conv3d_4s122_64_128", False, 3, 64, 128, 4, (1, 2, 2), (0, 1, 1), (7, 16, 16), 2),
    ("conv3d_4s122_thin3_32", False, 3, 3, 32, 4, (1, 2, 2), (0, 1, 1), (16, 64, 64), 1),
    ("conv3d_4s122_thin1_32", False, 3, 1, 32, 4, (1, 2, 2), (0, 1, 1), (15, 64, 64), 1),
    ("conv3d_4s122_thin2_32_n3", False, 3, 2, 32, 4, (1, 2, 2), (0, 1, 1), (6, 64, 64), 3),
    ("conv3d_4s122_256_1", False, 3, 256, 1, 4, (1, 2, 2), (0, 1, 1), (7, 8, 8), 2),
]


def guarded_cl(shape, fill=None):
    """A CL16 tensor inside a NaN-filled allocation (8192 NaN elements on either side, same strides as ops_cl.cl_empty's)."""
    from dcvgan_amd import ops_cl
    n, c, sp = shape[0], shape[1], tuple(shape[2:])
    p = ops_cl.pitch_of(c)
    numel = n * p
    for s in sp:
        numel *= s
    G = 8192
    store = torch.full((numel + 2 * G,), float("nan"), dtype=torch.bfloat16, device=DEV)
    body = store[G:G + numel].view((n,) + sp + (p,))
    body.zero_()
    perm = (0, len(sp) + 1) + tuple(range(1, len(sp) + 1))
    t = body.permute(*perm)[:, :c]
    if fill is not None:
        t.copy_(fill)
    return t, store, G


def margins_intact(store, G):
    return bool(torch.isnan(store[:G].float()).all()) and bool(torch.isnan(store[-G:].float()).all())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_cl16(case):
    from dcvgan_amd import native, ops, ops_cl
    native.lib()
    name, tr, nd, cin, cout, k, s, p, sp, n = case
    g = torch.Generator().manual_seed(hash(name) % 10000)
    s_t = (s,) * nd if isinstance(s, int) else s
    p_t = (p,) * nd if isinstance(p, int) else p
    w = r16(torch.randn(((cin, cout) if tr else (cout, cin)) + (k,) * nd, generator=g) * 0.1).requires_grad_(True)
    x = r16(torch.randn((n, cin) + sp, generator=g)).requires_grad_(True)
    fn = {(False, 2): F.conv2d, (False, 3): F.conv3d, (True, 2): F.conv_transpose2d}[(tr, nd)]
    y_ref = fn(x, w, None, s_t, p_t)
    cot = r16(torch.randn(y_ref.shape, generator=g))
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot).sum(), [x, w])
    xc, xstore, G = guarded_cl(x.shape, x.detach().to(DEV))
    xc.requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    y = ops_cl.conv(xc, wd, ops.conv_geom(wd, s_t, p_t, tr))
    kn = native.lib().dcv_debug_last_kernel().decode()
    assert y.dtype == torch.bfloat16 and ("cl_gather" in kn or "cl_thin3x3" in kn or "cl_widen3x3" in kn or "cl_stem3d" in kn or "cl_patch_convt" in kn), kn
    if name.startswith("conv3d_4s122_thin"):
        assert "cl_stem3d" in kn, kn            # the fused 3-D stem form
    if name.startswith("conv2d_3s1p1_thin"):
        assert "cl_widen3x3" in kn, kn          # the fused thin-source form
    if name.startswith("convT2d_3s1p1"):
        assert "cl_thin3x3" in kn, kn           # the fused thin-destination form is the one these shapes select
    cc, cstore, G2 = guarded_cl(cot.shape, cot.to(DEV))
    gx, gw = torch.autograd.grad(y, [xc, wd], cc)
    torch.cuda.synchronize()
    errs = [rel(y.float(), y_ref), rel(gx.float(), gx_ref), rel(gw, gw_ref)]
    assert errs[0] < 5e-3 and errs[1] < 5e-3 and errs[2] < 2e-5, (name, errs)
    assert margins_intact(xstore, G) and margins_intact(cstore, G2)
    # the channels between C and C rounded up to 8 are exactly zero: readers consume whole 16-byte granules up to there (and nothing beyond)
    for t in (y, gx):
        c8 = (t.shape[1] + 7) // 8 * 8
        if c8 > t.shape[1]:
            st = torch.as_strided(t.detach(), t.shape[:1] + (c8,) + t.shape[2:], t.stride())
            assert float(st[:, t.shape[1]:].float().abs().max()) == 0.0


def test_conv_cl16_into_concat_slices_and_accumulated_fp32_boundary():
    """Two producers write the two channel slices of one channels-last buffer; the consumer convolution reads the whole buffer; conversions at
    the boundary: fp32 strided views in (a (B,C,T,H,W) permuted generator output), fp32 contiguous out, gradients back to fp32."""
    from dcvgan_amd import native, ops, ops_cl
    native.lib()
    g = torch.Generator().manual_seed(5)
    B, T = 2, 6
    vid = r16(torch.randn(B, T, 3, 16, 16, generator=g)).permute(0, 2, 1, 3, 4)         # non-contiguous (B,3,T,16,16) view, as generator.py:433 returns
    geo = r16(torch.randn(B, 1, T, 16, 16, generator=g))
    wc = r16(torch.randn(32, 3, 4, 4, 4, generator=g) * 0.1); wg = r16(torch.randn(32, 1, 4, 4, 4, generator=g) * 0.1)
    s3, p3 = (1, 2, 2), (0, 1, 1)
    xv, xg = vid.clone().requires_grad_(True), geo.clone().requires_grad_(True)
    ws = [t.clone().requires_grad_(True) for t in (wc, wg)]
    hc = F.leaky_relu(F.conv3d(xv, ws[0], None, s3, p3), 0.2); hg = F.leaky_relu(F.conv3d(xg, ws[1], None, s3, p3), 0.2)
    # HIP
    dv = vid.to(DEV).requires_grad_(True); dg = geo.to(DEV).requires_grad_(True)
    dws = [t.to(DEV).requires_grad_(True) for t in (wc, wg)]
    cat = ops_cl.ConcatBuffer(B, 32, 32, (hc.shape[2], 8, 8), DEV)
    a = ops_cl.conv(ops_cl.from_f32(dv), dws[0], ops.conv_geom(dws[0], s3, p3, False), ops.ACT_LEAKY, 0.2, out=cat.first)
    b = ops_cl.conv(ops_cl.from_f32(dg), dws[1], ops.conv_geom(dws[1], s3, p3, False), ops.ACT_LEAKY, 0.2, out=cat.second)
    h = cat.join(a, b)
    out = ops_cl.to_f32(h)
    ref = torch.cat([hc, hg], 1)
    assert out.is_contiguous() and rel(out, ref) < 5e-3
    cot = r16(torch.randn(ref.shape, generator=g))
    gr = torch.autograd.grad((ref * cot).sum(), [xv, xg] + ws)
    gh = torch.autograd.grad(out, [dv, dg] + dws, cot.to(DEV))
    assert gh[0].dtype == torch.float32 and gh[0].shape == vid.shape
    for u, v, bar in zip(gh, gr, (8e-3, 8e-3, 3e-3, 3e-3)):   # data gradients: two bf16 roundings on the way; weight gradients see the rounded activation derivative
        assert rel(u, v) < bar, (rel(u, v), bar)


@pytest.mark.parametrize("shape,act,drop", [((6, 64, 8, 8), (1, 0.2), False), ((4, 96, 4, 4), (1, 0.0), True), ((2, 32, 5, 16, 16), (1, 0.2), False), ((3, 128, 1, 1), (0, 0.0), False)])
def test_bn_act_cl16(shape, act, drop):
    """BatchNorm (+ Dropout2d mask) (+ (Leaky)ReLU), training mode: output and input gradient within bf16 rounding of torch's fp32 result on the same
    (bf16-valued) input; gamma / beta gradients, batch statistics and running statistics are fp32 quantities: 2e-3 (they sum bf16-rounded dy)."""
    from dcvgan_amd import native, ops, ops_cl
    native.lib()
    g = torch.Generator().manual_seed(sum(shape))
    Cn = shape[1]
    x = r16(torch.randn(shape, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gamma = (torch.rand(Cn, generator=g) + 0.5).requires_grad_(True); beta = (torch.randn(Cn, generator=g) * 0.2).requires_grad_(True)
    rm, rv = torch.zeros(Cn), torch.ones(Cn)
    mask = ((torch.rand(shape[0], Cn, generator=g) > 0.5).float() * 2.0) if drop else None
    z = F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if mask is not None:
        z = z * mask.view(shape[0], Cn, *([1] * (len(shape) - 2)))
    y_ref = F.leaky_relu(z, act[1]) if act[0] else z
    cot = r16(torch.randn(shape, generator=g))
    gr = torch.autograd.grad((y_ref * cot).sum(), [x, gamma, beta])
    xc = ops_cl.cl_empty(shape, DEV, zero=True); xc.copy_(x.detach().to(DEV)); xc.requires_grad_(True)
    gd, bd = gamma.detach().to(DEV).requires_grad_(True), beta.detach().to(DEV).requires_grad_(True)
    rmd, rvd = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    md = mask.to(DEV).view(shape[0], Cn, 1, 1) if mask is not None else None
    y = ops_cl.bn_act(xc, gd, bd, rmd, rvd, True, ops.ACT_LEAKY if act[0] else ops.ACT_NONE, act[1], md, num_batches_tracked=nbt)
    cc = ops_cl.cl_empty(shape, DEV, zero=True); cc.copy_(cot.to(DEV))
    gh = torch.autograd.grad(y, [xc, gd, bd], cc)
    assert rel(y.float(), y_ref) < 5e-3 and rel(gh[0].float(), gr[0]) < 8e-3, (rel(y.float(), y_ref), rel(gh[0].float(), gr[0]))
    assert rel(gh[1], gr[1]) < 2e-3 and rel(gh[2], gr[2]) < 2e-3, (rel(gh[1], gr[1]), rel(gh[2], gr[2]))
    assert rel(rmd, rm) < 1e-5 and rel(rvd, rv) < 1e-5 and int(nbt) == 1
    # eval mode uses the running statistics
    ye = ops_cl.bn_act(xc.detach(), gd.detach(), bd.detach(), rmd, rvd, False, ops.ACT_NONE, 0.0)
    assert rel(ye.float(), F.batch_norm(x.detach(), rm, rv, gamma.detach(), beta.detach(), False, 0.1, 1e-5)) < 5e-3


def test_noise_and_activation_cl16():
    from dcvgan_amd import native, ops, ops_cl
    native.lib()
    x = ops_cl.cl_empty((4, 64, 6, 8, 8), DEV, zero=True); x.copy_(torch.randn(4, 64, 6, 8, 8, device=DEV))
    y = ops_cl.noise_add(x, 0.2, None, 1234, 7)
    d = (y.float() - x.float())
    assert abs(float(d.mean())) < 5e-3 and abs(float(d.std()) - 0.2) < 1e-2          # N(0, 0.2^2), up to bf16 rounding of the sum
    y2 = ops_cl.noise_add(x, 0.2, None, 1234, 7)
    assert torch.equal(y, y2) and not torch.equal(y, ops_cl.noise_add(x, 0.2, None, 1234, 8))
    s = torch.randn(4, 64, 6, 8, 8, device=DEV)
    assert rel(ops_cl.noise_add(x, 0.5, s).float(), x.float() + 0.5 * s.to(torch.bfloat16).float()) < 4e-3
    xr = x.clone().requires_grad_(True)
    t = ops_cl.act(xr, ops.ACT_TANH)
    (gt,) = torch.autograd.grad(t, xr, torch.ones_like(t))
    assert rel(t.float(), torch.tanh(x.float())) < 4e-3 and rel(gt.float(), 1 - torch.tanh(x.float()) ** 2) < 1e-2


@pytest.mark.parametrize("name", ["surreal-depth1", "isogd-flow", "isogd-depth"])
def test_models_cl16_against_fp32_path(name):
    """Every model of a config at a quarter of its width, same weights and random draws (the discriminators' Noise layers switched off for this
    comparison: the two paths index their Philox streams differently, so their noise realisations differ): the CL16 path's outputs within 3e-2 / 5e-2 of
    the fp32 HIP path's (bf16 storage of ~20 layers' activations), parameter gradients of a generator-loss backward within 1e-1 relative L2 per model."""
    import os
    from dcvgan_amd import native, ops_cl, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    native.lib()
    cfg = CONFIGS[name].scaled(batchsize=2, width_div=4)
    torch.manual_seed(3)
    models = trainer.build_models(cfg, DEV)
    for m in models.values():
        for sub in m.modules():
            if hasattr(sub, "use_noise"):
                sub.use_noise = False
    loss = trainer.build_loss(cfg)

    def run():
        r = PhiloxRng(99)
        for m in models.values():
            m._rng = r
            m.zero_grad()
        xg = models["ggen"].sample_videos(2)
        xc = models["cgen"].forward_videos(xg)
        ys = [models["idis"](xg[:, :, 3], xc[:, :, 3]), models["vdis"](xg, xc), models["gdis"](xg, xc)]
        loss.compute_gen_loss(*ys).backward()
        grads = {n: torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).clone() for n, m in models.items()
                 if any(p.grad is not None for p in m.parameters())}
        return xg.detach().clone(), xc.detach().clone(), [y.detach().clone() for y in ys], grads

    ref = run()
    ops_cl.enable(True)
    try:
        n0 = native.launch_count()
        got = run()
        assert native.launch_count() - n0 > 100
    finally:
        ops_cl.enable(False)
    # control: the OTHER 16-bit mode of this library (bf16 MFMA products on fp32 tensors, conv_mfma.hip) — an independent implementation whose distance
    # from the fp32 path shows what 8-bit operand rounding does to deep (Leaky)ReLU gradients by itself (branch flips of near-zero pre-activations)
    native.set_precision("bf16")
    try:
        ctl = run()
    finally:
        native.set_precision("fp32")
    assert got[0].dtype == torch.float32 and got[0].shape == ref[0].shape and got[0].stride() == ref[0].stride()      # the boundary is unchanged
    rep = {"xg": rel(got[0], ref[0]), "xc": rel(got[1], ref[1])}
    for k, a, b in zip(("y_idis", "y_vdis", "y_gdis"), got[2], ref[2]):
        assert a.shape == b.shape
        rep[k] = rel(a, b)
    cosn = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()))
    for n in ref[3]:
        rep["grad_" + n] = rel(got[3][n], ref[3][n])
        rep["gradcos_" + n] = cosn(got[3][n], ref[3][n])
        rep["gradnorm_ratio_" + n] = float(got[3][n].double().norm() / ref[3][n].double().norm())
        rep["control_bf16products_grad_" + n] = rel(ctl[3][n], ref[3][n])
    rep["control_bf16products_xc"] = rel(ctl[1], ref[1])
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, f"cl16_models_{name}.txt"), "w") as f:
            f.write("\n".join(f"{k} {v:.3e}" for k, v in rep.items()) + "\n")
    assert rep["xg"] < 3e-2 and rep["xc"] < 4e-2, rep
    assert max(rep[k] for k in ("y_idis", "y_vdis", "y_gdis")) < 1e-1, rep
    # gradients: the discriminators' (5 layers from the loss) within 0.15; the generators' (20-40 (Leaky)ReLU layers deep) are dominated by branch flips of
    # pre-activations within bf16 rounding of zero (~0.3 % of the elements per layer): direction and size must hold (cos > 0.85, norm within 15 %) and the
    # distance must not exceed 1.5x what the independent bf16-product mode shows on the same run
    for n in ref[3]:
        if n.endswith("dis"):
            assert rep["grad_" + n] < 0.15, (n, rep)
        else:
            assert rep["gradcos_" + n] > 0.85 and 0.85 < rep["gradnorm_ratio_" + n] < 1.15, (n, rep)
            assert rep["grad_" + n] < max(0.2, 1.5 * rep["control_bf16products_grad_" + n]), (n, rep)


@pytest.mark.parametrize("name", ["surreal-depth1", "isogd-flow", "isogd-depth"])
def test_training_iteration_cl16(name):
    """One FULL-WIDTH iteration at B = 4 of each GPU config on the CL16 path beside the same iteration on the fp32 HIP path (same initial weights, same Philox
    streams): finite losses, every model's parameters move, and every loss within 5 % of the fp32 path's (measured 0.2-2 %; 15 % for the two configs whose
    discriminators add Noise — the two paths index the Philox stream differently and draw different realisations).  The measured values are written to
    gpurun_out/cl16_iteration_<config>.txt (committed as profiles/r04_cl16_tolerance.txt)."""
    from dcvgan_amd import native, ops_cl, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    native.lib()
    cfg = CONFIGS[name].scaled(batchsize=4, num_gen_update=1)
    g = torch.Generator().manual_seed(1)
    xc = (torch.rand(4, 3, 16, 64, 64, generator=g) * 2 - 1).to(DEV); xg = (torch.rand(4, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(DEV)
    outs = {}
    for mode in (False, True):
        torch.manual_seed(5)
        models = trainer.build_models(cfg, DEV)
        r = PhiloxRng(77)
        for m in models.values():
            m._rng = r
        before = {n: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone() for n, m in models.items()}
        ops_cl.enable(mode)
        try:
            runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg), sync_losses=True)
            outs[mode] = runner.step(xc, xg, 3)
        finally:
            ops_cl.enable(False)
        assert all(v == v and abs(v) < 100 for v in outs[mode].values()), outs[mode]
        for n, m in models.items():
            after = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
            assert float((after != before[n]).float().mean()) > 0.5, n
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, f"cl16_iteration_{name}.txt"), "w") as f:
            for k in outs[True]:
                f.write(f"{name} B=4 full width  {k}: bf16cl {outs[True][k]:.6f}  fp32 {outs[False][k]:.6f}  relative {abs(outs[True][k] - outs[False][k]) / max(1.0, abs(outs[False][k])):.2e}"
                        f"  (bar {5e-2 if name == 'surreal-depth1' else 1.5e-1})\n")
    for k in outs[True]:
        assert abs(outs[True][k] - outs[False][k]) < (5e-2 if name == "surreal-depth1" else 1.5e-1) * max(1.0, abs(outs[False][k])), (k, outs[True][k], outs[False][k])


def test_stress_shape_32x128x128_cl16():
    """BASELINE configs[4]'s 32-frame 128 x 128 clips (the discriminators are size-agnostic): vdis + gdis forward and backward on the CL16
    path at B = 2, logits within 5e-2 of the fp32 HIP path's, input gradients cos > 0.98, parameter gradients within 0.15."""
    from dcvgan_amd import discriminator as D, native, ops_cl
    native.lib()
    torch.manual_seed(0)
    vdis = D.VideoDiscriminator(2, 3, False, 0.2, 64).to(DEV)
    gdis = D.GradientDiscriminator(2, 3, False, 0.2, 32).to(DEV)
    g = torch.Generator().manual_seed(4)
    xc0 = (torch.rand(2, 3, 32, 128, 128, generator=g) * 2 - 1).to(DEV)
    xg0 = (torch.rand(2, 2, 32, 128, 128, generator=g) - 0.5).to(DEV)

    def run():
        xc, xg = xc0.clone().requires_grad_(True), xg0.clone().requires_grad_(True)
        for m in (vdis, gdis):
            m.zero_grad()
        yv, yg = vdis(xg, xc), gdis(xg, xc)
        (yv.mean() + yg.mean()).backward()
        return yv.detach(), yg.detach(), xc.grad.clone(), xg.grad.clone(), torch.cat([p.grad.reshape(-1) for m in (vdis, gdis) for p in m.parameters()]).clone()

    ref = run()
    ops_cl.enable(True)
    try:
        got = run()
    finally:
        ops_cl.enable(False)
    assert got[0].shape == ref[0].shape == (2, 20, 8, 8) and got[1].shape == ref[1].shape == (2, 19, 8, 8)
    assert rel(got[0], ref[0]) < 5e-2 and rel(got[1], ref[1]) < 5e-2, (rel(got[0], ref[0]), rel(got[1], ref[1]))
    cosn = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()))
    assert cosn(got[2], ref[2]) > 0.98 and cosn(got[3], ref[3]) > 0.98, (cosn(got[2], ref[2]), cosn(got[3], ref[3]))
    assert rel(got[4], ref[4]) < 0.15, rel(got[4], ref[4])


def test_iteration_cl16_is_bitwise_reproducible():
    """Two runs of the same two iterations from the same seeds on the CL16 path give bit-identical losses and parameters: no atomics, every reduction
    (BatchNorm partials, weight-gradient slabs) in a fixed order — also with the discriminators on their own streams."""
    from dcvgan_amd import native, ops_cl, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    native.lib()
    cfg = CONFIGS["isogd-flow"].scaled(batchsize=2, width_div=2)
    g = torch.Generator().manual_seed(1)
    xc = (torch.rand(2, 3, 16, 64, 64, generator=g) * 2 - 1).to(DEV); xg = (torch.rand(2, 2, 16, 64, 64, generator=g) - 0.5).to(DEV)
    res = []
    ops_cl.enable(True)
    try:
        for _ in range(2):
            torch.manual_seed(9)
            models = trainer.build_models(cfg, DEV)
            r = PhiloxRng(5)
            for m in models.values():
                m._rng = r
            runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg), sync_losses=True)
            outs = [runner.step(xc, xg, 3), runner.step(xc, xg, 8)]
            res.append((outs, torch.cat([p.detach().reshape(-1) for m in models.values() for p in m.parameters()]).clone()))
    finally:
        ops_cl.enable(False)
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("mode", ["in_stream", "no_twins"])
def test_iteration_cl16_schedules_and_boundary_shortcuts_change_no_bit(mode):
    """The shipped iteration hands the main chain's weight gradients to a companion stream (ops_cl._wgrad_on_side) and passes bf16 twins across the module boundaries
    (ops_cl.twin_of).  Neither may change a bit of the forward, and the companion not one of anything: against the in-stream schedule three iterations give identical losses
    and parameters; without the twins the LOSSES are identical (the same forward bits) and the
    parameters agree to bf16 rounding of the fake clips' gradient sum (one rounding more on the short cut)."""
    from dcvgan_amd import native, ops_cl, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    native.lib()
    cfg = CONFIGS["surreal-depth1"].scaled(batchsize=2, width_div=2)
    g = torch.Generator().manual_seed(3)
    xc = (torch.rand(2, 3, 16, 64, 64, generator=g) * 2 - 1).to(DEV); xg = (torch.rand(2, 1, 16, 64, 64, generator=g) * 2 - 1).to(DEV)

    def run():
        torch.manual_seed(11)
        models = trainer.build_models(cfg, DEV)
        r = PhiloxRng(7)
        for m in models.values():
            m._rng = r
        runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg), sync_losses=True)
        outs = [runner.step(xc, xg, t) for t in (3, 8, 1)]
        return outs, torch.cat([p.detach().reshape(-1) for m in models.values() for p in m.parameters()]).clone()
    saved = (ops_cl._WGRAD_SIDE, ops_cl._TWINS)
    ops_cl.enable(True)
    try:
        base = run()
        if mode == "in_stream":
            ops_cl._WGRAD_SIDE = False
        else:
            ops_cl._TWINS = False
        other = run()
    finally:
        ops_cl._WGRAD_SIDE, ops_cl._TWINS = saved
        ops_cl.enable(False)
    if mode == "no_twins":
        assert base[0][0] == other[0][0], (base[0][0], other[0][0])      # first iteration: identical forward bits on both routes
        assert rel(base[1], other[1]) < 2e-2, rel(base[1], other[1])
    else:
        assert base[0] == other[0], (base[0], other[0])
        assert torch.equal(base[1], other[1])


def test_twin_is_dropped_when_the_fp32_tensor_is_modified_in_place():
    """to_f32() remembers its CL16 source and from_f32() hands that back instead of converting.  If the fp32 tensor was written in place in between
    (`videos.mul_(0.5)` between generator and discriminator) the source no longer holds its values: the twin must be dropped — forward AND the gradient equal
    the always-convert route (DCV_CL_NO_TWINS) bit for bit."""
    from dcvgan_amd import native, ops_cl
    native.lib()
    saved = ops_cl._TWINS
    ops_cl.enable(True)
    try:
        g = torch.Generator().manual_seed(5)
        src32 = torch.randn(2, 8, 4, 16, 16, generator=g).to(DEV)
        got = {}
        for twins in (True, False):
            ops_cl._TWINS = twins
            leaf = src32.clone().requires_grad_(True)
            x16 = ops_cl.from_f32(leaf)
            y = ops_cl.to_f32(x16)
            if twins:
                assert ops_cl.twin_of(y) is x16
            v = y.permute(0, 2, 1, 3, 4)                      # a boundary view, as the generators make
            v = ops_cl.carry_twin(v, y, lambda t: t.permute(0, 2, 1, 3, 4))
            y.mul_(0.5)                                       # in place, through the base of the view
            assert ops_cl.twin_of(y) is None and ops_cl.twin_of(v) is None
            z16 = ops_cl.from_f32(y)
            assert z16 is not x16
            z = ops_cl.to_f32(z16)
            (gx,) = torch.autograd.grad((z * z).sum(), leaf)
            got[twins] = (z.detach().clone(), gx.clone())
        assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][1], got[False][1])
        assert torch.equal(got[True][0], (r16(r16(src32) * 0.5)))   # the modification is in the values the consumer saw
        # and an untouched tensor keeps the short cut
        ops_cl._TWINS = True
        x16 = ops_cl.from_f32(src32)
        assert ops_cl.from_f32(ops_cl.to_f32(x16)) is x16
    finally:
        ops_cl._TWINS = saved
        ops_cl.enable(False)


@pytest.mark.parametrize("case", [("conv2d_64_128", False, 2, 64, 128, 4, 2, 1, (16, 16), 5), ("convT2d_128_64", True, 2, 128, 64, 4, 2, 1, (16, 16), 3), ("convT2d_64_96_w8_n5", True, 2, 64, 96, 4, 2, 1, (8, 8), 5),
                                  ("conv3d_64_128", False, 3, 64, 128, 4, (1, 2, 2), (0, 1, 1), (7, 16, 16), 2), ("conv2d_thin3_32", False, 2, 3, 32, 4, 2, 1, (64, 64), 2),
                                  ("convT2d_latent_50_128", True, 2, 50, 128, 4, 1, 0, (1, 1), 9)], ids=lambda c: c[0])
def test_bn_sums_from_the_conv_epilogue_cl16(case):
    """conv -> BatchNorm pairs: the convolution's epilogue leaves per-tile {sum, sum^2} of the STORED bf16 values and the BatchNorm op skips its own pass over them.
    Same statistics as that pass (fp64 combine of fp32 partial sums against fp64 combine of bounded fp32 runs): running statistics to 1e-6, and the normalised
    output differs in at most a few last-bit roundings."""
    from dcvgan_amd import native as N, ops, ops_cl
    name, tr, dims, cin, cout, k, s, p, sp, n = case
    g0 = torch.Generator().manual_seed(7)
    kk = (k,) * dims if isinstance(k, int) else k
    ss = (s,) * dims if isinstance(s, int) else s
    pp = (p,) * dims if isinstance(p, int) else p
    x = ops_cl.from_f32(torch.randn((n, cin) + sp, generator=g0).to(DEV))
    w = (torch.randn(((cin, cout) if tr else (cout, cin)) + kk, generator=g0) * 0.05).to(DEV)
    gamma = (torch.rand(cout, generator=g0) + 0.5).to(DEV); beta = (torch.randn(cout, generator=g0) * 0.1).to(DEV)
    geom = ops.conv_geom(w, ss, pp, tr)
    box = []
    y = ops_cl.conv(x, w, geom, bn_stats=box)
    assert len(box) == 1 and box[0][1] > 0, "the epilogue produced no sums for this geometry"
    y_plain = ops_cl.conv(x, w, geom)                  # without the extra epilogue (round 5: this one may take the split-K form where the position tiles are few: the
    assert rel(y_plain.float(), y.float()) < 4e-3      # same sums in another order, equal to one bf16 rounding; bit-equal where it does not)
    res = []
    for fused in (True, False):                        # the SAME convolution output through BatchNorm with the epilogue's sums and with BatchNorm's own pass
        rm, rv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        z = ops_cl.bn_act(y, gamma, beta, rm, rv, True, ops.ACT_LEAKY, 0.2, partials=box[0] if fused else None, num_batches_tracked=nbt)
        res.append((y.float().clone(), z.float().clone(), rm.clone(), rv.clone(), int(nbt)))
    (y1, z1, rm1, rv1, n1), (y0, z0, rm0, rv0, n0) = res
    assert n1 == n0 == 1
    assert rel(rm1, rm0) < 1e-6 and rel(rv1, rv0) < 1e-6, (rel(rm1, rm0), rel(rv1, rv0))
    assert rel(z1, z0) < 1e-4 and float((z1 != z0).float().mean()) < 2e-2, (rel(z1, z0), float((z1 != z0).float().mean()))


def test_gated_skip_gradient_cl16():
    """Inconv -> DownBlock 0 on the bf16 path: DownBlock 0's data gradient accumulates into the concat buffer's gradient slice AND applies Inconv's LeakyReLU
    derivative in its epilogue (dcv_cl_conv_backward_data_gated; ops.GradSlot) — against the three-step form (accumulate, then a derivative pass).  The fused form
    rounds once less, so the two agree to bf16 rounding, not bit for bit."""
    from dcvgan_amd import native, ops, ops_cl, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    native.lib()
    cfg = CONFIGS["isogd-depth"].scaled(batchsize=2, width_div=2)
    x = (torch.rand(6, 1, 64, 64, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(DEV)

    def run(gated):
        old = ops._GATED_DGRAD
        ops._GATED_DGRAD = gated
        ops_cl.enable(True)
        try:
            torch.manual_seed(9)
            cgen = trainer.build_models(cfg, DEV)["cgen"]
            cgen._rng = PhiloxRng(3)
            cgen.train()
            n0 = native.launch_count()
            z = torch.randn(6, cgen.dim_z, 1, 1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
            y = cgen(x, z)
            (y * torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)).sum().backward()
            torch.cuda.synchronize()
            return {n: p.grad.detach().clone() for n, p in cgen.named_parameters()}, native.launch_count() - n0
        finally:
            ops._GATED_DGRAD = old
            ops_cl.enable(False)

    run(True)                                          # (the first pass also builds the weight gradients' position tables, kept per geometry)
    ga, la = run(True)
    gb, lb = run(False)
    assert la == lb - 1, (la, lb)                      # exactly the derivative pass is gone
    for n in ga:
        assert rel(ga[n], gb[n]) < 2e-2, (n, rel(ga[n], gb[n]))
    assert rel(ga["inconv.main.0.weight"], gb["inconv.main.0.weight"]) < 1e-2


def test_out_view_that_is_not_a_multiple_of_8_channels_must_be_trailing():
    """the kernels write whole 8-channel groups; a destination view with C % 8 != 0 followed by live channels would have them zeroed silently: refused."""
    from dcvgan_amd import native, ops, ops_cl
    native.lib()
    buf = ops_cl.cl_empty((2, 32, 8, 8), DEV, zero=True)
    x = ops_cl.from_f32(torch.randn(2, 8, 8, 8, device=DEV))
    w = torch.randn(12, 8, 3, 3, device=DEV) * 0.1
    g = ops.conv_geom(w, (1, 1), (1, 1), False)
    with pytest.raises(native.NativeError):
        ops_cl.conv(x, w, g, out=buf[:, 8:20])                    # 12 channels in the middle of a 32-channel pixel
    cat = ops_cl.ConcatBuffer(2, 8, 12, (8, 8), DEV)              # ... the trailing member of a concatenation may have any width
    y = ops_cl.conv(x, w, g, out=cat.second)
    assert y.shape[1] == 12 and torch.isfinite(y.float()).all()
