"""GPU: the sizes of the two B = 100 bench configs (surreal-depth1: ggen ngf 96 — output-channel counts 768 / 384 / 192 / 96 on padded
128-wide tiles, F = 1600 frames; isogd-flow: two geometry channels — strided generator outputs, 2-channel stems and heads), which select
kernel variants and tile counts that B <= 16 (the oracle-checked sizes) and B = 70 (tests/test_b70_gpu.py) never reach, and where a wrong
in-range offset inside a raw-buffer descriptor would read a neighbour silently.
  * every ggen layer at ngf 96 / F = 1600, the Cg = 2 stems and heads at B = 100 (on the non-contiguous views the models really pass),
    the heaviest cgen / vdis layers at F = 1600 / B = 100: forward, data gradient, weight gradient against torch.nn.functional on the host
    (1e-5 / 2e-5; north_star's tolerance is 1e-3);
  * a batch-split identity over ALL FIVE models of surreal-depth1 and isogd-flow: in eval mode rows 0..15 of a B = 100 pass — ggen and
    cgen outputs, the three discriminators' logits, every parameter gradient for a cotangent that is zero outside those rows — equal the
    B = 16 pass (the size tests/test_fullwidth_gpu.py verifies against the oracle)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
B = 100
Fr = B * 16
S3, P3 = (1, 2, 2), (0, 1, 1)
# name, transposed, cin, cout, kernel, stride, padding, input shape, input layout ("" = contiguous)
LAYERS = [
    ("ggen96.0 convT 50->768 1x1->4", True, 50, 768, (4, 4), (1, 1), (0, 0), (Fr, 50, 1, 1), ""),
    ("ggen96.3 convT 768->384 @4", True, 768, 384, (4, 4), (2, 2), (1, 1), (Fr, 768, 4, 4), ""),
    ("ggen96.6 convT 384->192 @8", True, 384, 192, (4, 4), (2, 2), (1, 1), (Fr, 384, 8, 8), ""),
    ("ggen96.9 convT 192->96 @16", True, 192, 96, (4, 4), (2, 2), (1, 1), (Fr, 192, 16, 16), ""),
    ("ggen96.12 convT 96->1 @32", True, 96, 1, (4, 4), (2, 2), (1, 1), (Fr, 96, 32, 32), ""),
    ("ggen.12 convT 64->2 @32 (flow head)", True, 64, 2, (4, 4), (2, 2), (1, 1), (Fr, 64, 32, 32), ""),
    ("cgen.in conv 2->64 3x3 @64 (flow frames)", False, 2, 64, (3, 3), (1, 1), (1, 1), (Fr, 2, 64, 64), ""),
    ("idis.g conv 2->32 on frame t of the video view", False, 2, 32, (4, 4), (2, 2), (1, 1), (B, 2, 64, 64), "frame"),
    ("vdis.g conv3d 2->32 on the (B,C,T,H,W) view", False, 2, 32, (4, 4, 4), S3, P3, (B, 2, 16, 64, 64), "video"),
    ("gdis.1 conv3d 2->32 stem", False, 2, 32, (4, 4, 4), S3, P3, (B, 2, 15, 64, 64), ""),
    ("vdis.c conv3d 3->32 on the (B,C,T,H,W) view", False, 3, 32, (4, 4, 4), S3, P3, (B, 3, 16, 64, 64), "video"),
    ("vdis.1 conv3d 64->128", False, 64, 128, (4, 4, 4), S3, P3, (B, 64, 13, 32, 32), ""),
    ("vdis.5 conv3d 128->256", False, 128, 256, (4, 4, 4), S3, P3, (B, 128, 10, 16, 16), ""),
    ("gdis.5 conv3d 32->64", False, 32, 64, (4, 4, 4), S3, P3, (B, 32, 12, 32, 32), ""),
    ("cgen.up5 convT 128->64 @32", True, 128, 64, (4, 4), (2, 2), (1, 1), (Fr, 128, 32, 32), ""),
    ("cgen.down0 conv 64->64 @64", False, 64, 64, (4, 4), (2, 2), (1, 1), (Fr, 64, 64, 64), ""),
    ("cgen.down1 conv 64->128 @32", False, 64, 128, (4, 4), (2, 2), (1, 1), (Fr, 64, 32, 32), ""),
    ("cgen.up2 convT 512->256 @4", True, 512, 256, (4, 4), (2, 2), (1, 1), (Fr, 512, 4, 4), ""),
    ("cgen.out convT 128->3 3x3 @64", True, 128, 3, (3, 3), (1, 1), (1, 1), (Fr, 128, 64, 64), ""),
]


def rel(a, b):
    a = a.detach().cpu(); b = b.detach().cpu()
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


GUARD = 8192     # elements of NaN on either side of every operand


def guarded(t):
    """The same values inside a NaN-filled allocation, same shape and strides: a read that strays outside the tensor — a halo granule that
    should have been padding, an offset past the last sample — puts a NaN into the result instead of whatever the allocator left next door
    (the raw-buffer descriptors range-check against 2 GB, not against the tensor).  -> (view, storage)"""
    base = t
    while base._base is not None:
        base = base._base
    big = torch.full((base.numel() + 2 * GUARD,), float("nan"), device=t.device, dtype=t.dtype)
    big[GUARD:GUARD + base.numel()] = base.reshape(-1)
    return torch.as_strided(big, t.shape, t.stride(), GUARD + t.storage_offset()), big


def margins_intact(big, n):
    return bool(torch.isnan(big[:GUARD]).all() and torch.isnan(big[GUARD + n:]).all())


@pytest.fixture(scope="module")
def dev():
    from dcvgan_amd import native
    native.lib()
    return torch.device("cuda:0")


def _input(xs, layout, dev, g):
    """The tensor the model really hands the layer: generators emit (B, T, C, H, W) buffers viewed as (B, C, T, H, W)."""
    if layout == "video":       # discriminator.py:211-231 on generator.py:141's permuted view
        b, c, t, h, w = xs
        return torch.randn((b, t, c, h, w), device=dev, generator=g).permute(0, 2, 1, 3, 4)
    if layout == "frame":       # trainer.py:307: xg_fake[:, :, t_rand] of that view
        b, c, h, w = xs
        return torch.randn((b, 16, c, h, w), device=dev, generator=g).permute(0, 2, 1, 3, 4)[:, :, 7]
    return torch.randn(xs, device=dev, generator=g)


@pytest.mark.parametrize("case", LAYERS, ids=[c[0].split(" ")[0] for c in LAYERS])
def test_layer_at_b100(dev, case):
    from dcvgan_amd import ops
    name, tr, cin, cout, k, s, p, xs, layout = case
    g = torch.Generator(device=dev).manual_seed(13)
    x0 = _input(xs, layout, dev, g)
    xd, xbig = guarded(x0)
    assert xd.is_contiguous() == (layout == "") and xd.stride() == x0.stride()
    xd.requires_grad_(True)
    wd, wbig = guarded(torch.randn(((cin, cout) if tr else (cout, cin)) + k, device=dev, generator=g) * 0.05)
    wd.requires_grad_(True)
    y = ops.conv(xd, wd, ops.conv_geom(wd, s, p, tr))
    cotd, cbig = guarded(torch.randn(y.shape, device=dev, generator=g))
    gx, gw = torch.autograd.grad((y * cotd).sum(), [xd, wd])
    assert bool(torch.isfinite(y).all() and torch.isfinite(gx).all() and torch.isfinite(gw).all()), name        # nothing outside the operands was read
    assert margins_intact(xbig, xbig.numel() - 2 * GUARD) and margins_intact(wbig, wd.numel()) and margins_intact(cbig, cotd.numel()), name   # ... or written
    x, w, cot = xd.detach().cpu().contiguous().requires_grad_(True), wd.detach().cpu().requires_grad_(True), cotd.cpu()
    fn = F.conv_transpose2d if tr else (F.conv3d if len(k) == 3 else F.conv2d)
    y_ref = fn(x, w, None, s, p)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert rel(y, y_ref) < 1e-5, name
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot).sum(), [x, w])
    assert rel(gx, gx_ref) < 1e-5, name
    assert rel(gw, gw_ref) < 2e-5, name     # sums over up to 6.5 M positions: the host's own fp32 sum is the looser side


@pytest.mark.parametrize("name", ["surreal-depth1", "isogd-flow"])
def test_batch_split_identity_b100(dev, name):
    from dcvgan_amd import layers, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import InjectedRng
    from oracle import dcvgan_oracle as O
    from oracle.stepcheck import KINK_EPS, KINK_FRAC, ReplayRng64
    from tests import fullwidth as FW
    cfg = CONFIGS[name]
    assert cfg.batchsize == B
    torch.manual_seed(int(os.environ.get("DCV_TEST_SEED", "78")))      # (the environment override is for seed sweeps with DCV_REPORT_DIR)
    models = trainer.build_models(cfg, dev)
    g = torch.Generator(device=dev).manual_seed(4)
    for m in models.values():      # non-trivial running statistics, then eval mode: every sample is processed on its own
        for mod in m.modules():
            if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                mod.running_mean.copy_(torch.randn(mod.num_features, device=dev, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.num_features, device=dev, generator=g) + 0.5)
        m.eval()
    ggen, cgen, idis, vdis, gdis = (models[k] for k in ("ggen", "cgen", "idis", "vdis", "gdis"))
    t, Cg = 9, cfg.channel
    for k in filter(None, os.environ.get("DCV_FP32_MODULES", "").split(",")):      # diagnosis: these modules at native fp32 whatever the process default
        from dcvgan_amd import util
        util.set_precision(models[k], "fp32")

    def draws(n):
        """The same random numbers for sample i whatever the batch, in the models' draw order: ggen z_content, h0, e_1..e_16
        (generator.py:110-116); cgen z; then the Noise layers of idis (both stems, trunk x3) and vdis (trunk x3) when enabled."""
        gg = torch.Generator(device=dev).manual_seed(10)
        shapes = [(cfg.dim_z_content,), (cfg.dim_z_motion,)] + [(cfg.dim_z_motion,)] * 16 + [(cfg.dim_z_color,)]
        if cfg.use_noise["idis"]:
            shapes += [(Cg, 64, 64), (3, 64, 64), (64, 32, 32), (128, 16, 16), (256, 8, 8)]
        if cfg.use_noise["vdis"]:
            shapes += [(64, 13, 32, 32), (128, 10, 16, 16), (256, 7, 8, 8)]
        assert not cfg.use_noise["gdis"]
        return [("normal", torch.randn((B,) + s, device=dev, generator=gg)[:n].contiguous()) for s in shapes]

    def cots(ys):
        return [torch.cos(torch.arange(y[:16].numel(), dtype=torch.float32) * 0.3).view(y[:16].shape) for y in ys]

    def run(n):
        for m in models.values():
            m.zero_grad()
        r = InjectedRng(draws(n))
        for m in models.values():
            m._rng = r
        layers.KINK_TAP = kinks = []
        xg = ggen.sample_videos(n)
        xc = cgen.forward_videos(xg)
        yi, yv, yg = idis(xg[:, :, t], xc[:, :, t]), vdis(xg, xc), gdis(xg, xc)
        layers.KINK_TAP = None
        assert r.pos == len(r.log)
        tot = 0
        for y, cot in zip((yi, yv, yg), cots((yi, yv, yg))):
            tot = tot + (y[:16] * cot.to(dev)).sum()          # the cotangent is zero for rows >= 16
        tot.backward()
        grads = {(mn, k): p.grad.detach().cpu().clone() for mn, m in models.items() for k, p in m.named_parameters()}
        # rows 0..15 of every activation pattern (2-D layers see B * 16 frames, sample-major)
        k16 = [m[: m.shape[0] * 16 // n].clone() for m in kinks]
        return xg.detach()[:16].cpu(), xc.detach()[:16].cpu(), [y.detach()[:16].cpu() for y in (yi, yv, yg)], grads, k16

    def oracle64(kinks):
        """The same 16 samples through the pinned oracle in fp64 (eval mode: per-sample arithmetic), differentiating with `kinks`."""
        st = FW.states_of(models, torch.float64)
        rng = ReplayRng64([(k, v.cpu()) for k, v in draws(16)])
        with O.KinkTape(kinks) as tape:
            xg = O.ggen_sample_videos(st["ggen"], 16, 16, cfg.dim_z_content, cfg.dim_z_motion, Cg, rng, False)
            xc = O.cgen_forward_videos(st["cgen"], xg, cfg.dim_z_color, rng, False)
            ys = (O.idis_forward(st["idis"], xg[:, :, t], xc[:, :, t], cfg.use_noise["idis"], cfg.noise_sigma["idis"], rng, False),
                  O.vdis_forward(st["vdis"], xg, xc, cfg.use_noise["vdis"], cfg.noise_sigma["vdis"], rng, False),
                  O.gdis_forward(st["gdis"], xg, xc, cfg.use_noise["gdis"], cfg.noise_sigma["gdis"], rng, False))
        assert tape.pos == len(kinks) and rng.pos == len(rng.log)
        sum((y * c.double()).sum() for y, c in zip(ys, cots(ys))).backward()
        grads = {(mn, k): p.grad.detach() for mn in st for k, p in st[mn].items() if p.requires_grad}
        flips = sum(m[0] for m in tape.mismatch); total = sum(m[1] for m in tape.mismatch); far = max(m[2] for m in tape.mismatch)
        oracle64.per_activation = [tuple(m[:3]) for m in tape.mismatch]
        return xg.detach(), xc.detach(), [y.detach() for y in ys], grads, (flips, total, far)

    xg16, xc16, ys16, gr16, k16 = run(16)
    xgB, xcB, ysB, grB, kB = run(B)
    # forward: the B = 100 rows equal the B = 16 pass and the fp64 oracle
    assert rel(xgB, xg16) < 1e-5 and rel(xcB, xc16) < 1e-5
    for a, b in zip(ysB, ys16):
        assert rel(a, b) < 1e-5
    # gradients: each pass against the fp64 oracle evaluated with THAT pass's own activation pattern (the two passes run different
    # kernel variants, so a (Leaky)ReLU pre-activation within rounding of zero may pick different branches: compared with each other
    # directly, the gradients differ by up to 4e-3 from a handful of such elements)
    differing = sum(int((a != b).sum()) for a, b in zip(k16, kB))
    for tag, grads, kinks in (("B=100", grB, kB), ("B=16", gr16, k16)):
        oxg, oxc, oys, ogr, (flips, total, far) = oracle64(kinks)
        assert flips <= max(8, KINK_FRAC * total) and far <= KINK_EPS, (tag, flips, total, far)
        assert rel(xgB if tag == "B=100" else xg16, oxg) < 1e-5 and rel(xcB if tag == "B=100" else xc16, oxc) < 1e-5
        worst = max((rel(grads[key], gref), key) for key, gref in ogr.items())
        if os.environ.get("DCV_REPORT_DIR"):     # every parameter's gradient against the fp64 replay, for comparing precision modes
            os.makedirs(os.environ["DCV_REPORT_DIR"], exist_ok=True)
            with open(os.path.join(os.environ["DCV_REPORT_DIR"], f"b100_split_{name}_{os.environ.get('DCV_PRECISION', 'fp32')}_seed{os.environ.get('DCV_TEST_SEED', '78')}_{tag.replace('=', '')}.txt"), "w") as f:
                f.write("# per recorded activation pattern, in forward order: (elements where the fp64 replay's own sign differs, elements, their largest normalised distance from zero)\n")
                f.write("# " + " ".join("%d/%d/%.1e" % m for m in oracle64.per_activation) + "\n")
                for key, gref in sorted(ogr.items(), key=lambda kv: -rel(grads[kv[0]], kv[1])):
                    f.write("%-8s %-40s %.3e  |g| %.3e\n" % (key[0], key[1], rel(grads[key], gref), float(gref.norm())))
            if os.environ.get("DCV_REPORT_ONLY"):
                continue
        assert worst[0] < 1e-4, (tag, worst, differing)
        assert len(ogr) == 83          # every parameter tensor of the five models
    assert differing <= 64, differing
