"""GPU: the fp16 build of the 16-bit channels-last path (dcv_clf16_*: the same kernels with _Float16 elements and v_mfma_f32_32x32x16_f16) — BASELINE configs[4] names
"fp16 MFMA" for the 32 x 128 x 128 discriminator shape.  Same scheme as tests/test_cl16_gpu.py: operands rounded to fp16 first, the same values through torch's fp32 CPU
ops; one fp16 rounding of the result is 2^-11 relative (~2.8e-4 relative L2): activations / data gradients < 1e-3, weight gradients (fp32 out) < 2e-5."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def r16(t):
    return t.to(torch.float16).float()


@pytest.fixture()
def fp16():
    from dcvgan_amd import native, ops_cl
    native.lib()
    ops_cl.enable(False, half="fp16")       # element type fp16; the models' switch stays off (these tests call the ops directly)
    try:
        yield
    finally:
        ops_cl.enable(False, half="bf16")


CASES = [
    ("conv3d_4s122_64_128", False, 3, 64, 128, 4, (1, 2, 2), (0, 1, 1), (7, 16, 16), 2),
    ("conv3d_4s122_thin2_32", False, 3, 2, 32, 4, (1, 2, 2), (0, 1, 1), (9, 32, 32), 2),
    ("conv3d_4s122_thin3_32_stem", False, 3, 3, 32, 4, (1, 2, 2), (0, 1, 1), (6, 64, 64), 2),
    ("conv3d_4s122_256_1", False, 3, 256, 1, 4, (1, 2, 2), (0, 1, 1), (7, 8, 8), 2),
    ("conv2d_4s2p1_64_128", False, 2, 64, 128, 4, 2, 1, (16, 16), 3),
    ("convT2d_4s2p1_128_64", True, 2, 128, 64, 4, 2, 1, (16, 16), 3),
    ("convT2d_3s1p1_128_3", True, 2, 128, 3, 3, 1, 1, (64, 64), 2),
    ("conv2d_3s1p1_thin1_64", False, 2, 1, 64, 3, 1, 1, (64, 64), 2),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_fp16(fp16, case):
    from dcvgan_amd import ops, ops_cl
    name, tr, nd, cin, cout, k, s, p, sp, n = case
    g = torch.Generator().manual_seed(abs(hash(name)) % 10000)
    s_t = (s,) * nd if isinstance(s, int) else s
    p_t = (p,) * nd if isinstance(p, int) else p
    w = r16(torch.randn(((cin, cout) if tr else (cout, cin)) + (k,) * nd, generator=g) * 0.1).requires_grad_(True)
    x = r16(torch.randn((n, cin) + sp, generator=g)).requires_grad_(True)
    fn = {(False, 2): F.conv2d, (False, 3): F.conv3d, (True, 2): F.conv_transpose2d}[(tr, nd)]
    y_ref = fn(x, w, None, s_t, p_t)
    cot = r16(torch.randn(y_ref.shape, generator=g))
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot).sum(), [x, w])
    with torch.no_grad():
        xc = ops_cl.from_f32(x.detach().to(DEV))
        cc = ops_cl.from_f32(cot.to(DEV))
    assert xc.dtype == torch.float16
    xc.requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    y = ops_cl.conv(xc, wd, ops.conv_geom(wd, s_t, p_t, tr))
    assert y.dtype == torch.float16
    gx, gw = torch.autograd.grad(y, [xc, wd], cc)
    torch.cuda.synchronize()
    errs = [rel(y.float(), y_ref), rel(gx.float(), gx_ref), rel(gw, gw_ref)]
    assert errs[0] < 1e-3 and errs[1] < 1e-3 and errs[2] < 2e-5, (name, errs)


def test_bn_act_fp16(fp16):
    from dcvgan_amd import ops, ops_cl
    g = torch.Generator().manual_seed(2)
    x = r16(torch.randn(3, 64, 5, 8, 8, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(64, generator=g) + 0.5).requires_grad_(True); beta = (torch.randn(64, generator=g) * 0.1).requires_grad_(True)
    y_ref = F.leaky_relu(F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5), 0.2)
    cot = r16(torch.randn(y_ref.shape, generator=g))
    gx_ref, gg_ref, gb_ref = torch.autograd.grad((y_ref * cot).sum(), [x, gamma, beta])
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    with torch.no_grad():
        xc = ops_cl.from_f32(x.detach().to(DEV)); cc = ops_cl.from_f32(cot.to(DEV))
    xc.requires_grad_(True)
    gd, bd = gamma.detach().to(DEV).requires_grad_(True), beta.detach().to(DEV).requires_grad_(True)
    y = ops_cl.bn_act(xc, gd, bd, rm, rv, True, ops.ACT_LEAKY, 0.2)
    gx, gg, gb = torch.autograd.grad(y, [xc, gd, bd], cc)
    torch.cuda.synchronize()
    assert rel(y.float(), y_ref) < 1e-3 and rel(gx.float(), gx_ref) < 2e-3 and rel(gg, gg_ref) < 1e-3 and rel(gb, gb_ref) < 1e-3


def test_stress_shape_fp16_beside_fp32_and_what_fp16_loses():
    """vdis + gdis on 32 x 128 x 128 flow clips in fp16, beside the fp32 HIP path from the same weights: logits, the largest pre-BatchNorm magnitude
    (fp16's largest finite value is 65504) and the gradients — the cotangent of a mean() over the logits is ~1e-4 per element and shrinks further on the way down,
    below fp16's smallest normal number (6.1e-5): what arrives at the first layers is what survives as subnormals."""
    from dcvgan_amd import discriminator as D, layers, native, ops_cl
    native.lib()
    B = 2
    torch.manual_seed(0)
    vdis = D.VideoDiscriminator(2, 3, False, 0.0, 64).to(DEV)
    gdis = D.GradientDiscriminator(2, 3, False, 0.0, 32).to(DEV)
    xc = torch.rand(B, 3, 32, 128, 128, device=DEV) * 2 - 1; xg = torch.rand(B, 2, 32, 128, 128, device=DEV) - 0.5

    def run(half):
        ops_cl.enable(half is not None, half=half or "bf16")
        taps = []
        layers.PREBN_TAP = taps if half else None
        try:
            for m in (vdis, gdis):
                m.zero_grad()
            yv, yg = vdis(xg, xc), gdis(xg, xc)
            (yv.float().mean() + yg.float().mean()).backward()
            torch.cuda.synchronize()
            grads = {f"{n}.{k}": p.grad.detach().clone() for n, m in (("vdis", vdis), ("gdis", gdis)) for k, p in m.named_parameters()}
            return yv.float(), yg.float(), grads, (max(float(t) for t in taps) if taps else None)
        finally:
            layers.PREBN_TAP = None
            ops_cl.enable(False, half="bf16")

    yv32, yg32, g32, _ = run(None)
    yvh, ygh, gh, peak = run("fp16")
    yvb, ygb, gb, peak_b = run("bf16")
    assert torch.isfinite(yvh).all() and torch.isfinite(ygh).all() and peak is not None and peak < 65504 / 16, peak       # headroom of >= 4 binades at random init
    assert rel(yvh, yv32) < 2e-2 and rel(ygh, yg32) < 2e-2 and rel(yvh, yv32) < rel(yvb, yv32) + 1e-3                          # 10 mantissa bits against bf16's 7
    cos = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()).clamp_min(1e-300))
    worst_h = min(cos(gh[k], g32[k]) for k in g32 if g32[k].numel() > 64)
    worst_b = min(cos(gb[k], g32[k]) for k in g32 if g32[k].numel() > 64)
    print(f"stress shape B = {B}: pre-BatchNorm peak fp16 {peak:.1f} (bf16 {peak_b:.1f}); worst gradient cosine vs the fp32 path: fp16 {worst_h:.4f}, bf16 {worst_b:.4f}")
    assert all(torch.isfinite(v).all() for v in gh.values())
    assert worst_b > 0.97
