"""GPU: the benchmark's own sizes.  bench.py runs isogd-depth at B = 70 (F = 1120 frames), where the kernels pick
variants that smaller batches never reach (no split-K, patch staging with whole-tile rows, thin forms above 65,536
positions, the weight-gradient chunking) and where a bad offset inside a raw-buffer range would read a neighbour
instead of faulting.  So:
  * the heaviest layers and the thin / few-channel layers at exactly the B = 70 shapes: forward, data gradient and
    weight gradient against torch.nn.functional on the host (1e-5; the tolerance of north_star is 1e-3);
  * BatchNorm + activation (training mode) at the largest activation of the step;
  * a batch-split identity over whole models: in eval mode (per-sample arithmetic) rows 0..15 of a B = 70 pass of
    cgen and the three discriminators — outputs, input gradients and parameter gradients for a cotangent that is
    zero outside those rows — equal the B = 16 pass that tests/test_fullwidth_gpu.py verifies against the oracle."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
B = 70
Fr = B * 16
S3, P3 = (1, 2, 2), (0, 1, 1)
# name, transposed, cin, cout, kernel, stride, padding, input shape        
LAYERS = [
    ("cgen.up5 convT 128->64 @32", True, 128, 64, (4, 4), (2, 2), (1, 1), (Fr, 128, 32, 32)),
    ("cgen.up4 convT 256->64 @16", True, 256, 64, (4, 4), (2, 2), (1, 1), (Fr, 256, 16, 16)),
    ("cgen.up3 convT 512->128 @8", True, 512, 128, (4, 4), (2, 2), (1, 1), (Fr, 512, 8, 8)),
    ("cgen.down0 conv 64->64 @64", False, 64, 64, (4, 4), (2, 2), (1, 1), (Fr, 64, 64, 64)),
    ("vdis.1 conv3d 64->128", False, 64, 128, (4, 4, 4), S3, P3, (B, 64, 13, 32, 32)),
    ("vdis.5 conv3d 128->256", False, 128, 256, (4, 4, 4), S3, P3, (B, 128, 10, 16, 16)),
    ("gdis.9 conv3d 64->128", False, 64, 128, (4, 4, 4), S3, P3, (B, 64, 9, 16, 16)),
    ("cgen.up2 convT 512->256 @4", True, 512, 256, (4, 4), (2, 2), (1, 1), (Fr, 512, 4, 4)),
    ("cgen.out convT 128->3 3x3 @64", True, 128, 3, (3, 3), (1, 1), (1, 1), (Fr, 128, 64, 64)),
    ("cgen.in conv 1->64 3x3 @64", False, 1, 64, (3, 3), (1, 1), (1, 1), (Fr, 1, 64, 64)),
    ("ggen.12 convT 64->1 @32", True, 64, 1, (4, 4), (2, 2), (1, 1), (Fr, 64, 32, 32)),
    ("vdis.c conv3d 3->32 stem", False, 3, 32, (4, 4, 4), S3, P3, (B, 3, 16, 64, 64)),
    ("gdis.1 conv3d 1->32 stem", False, 1, 32, (4, 4, 4), S3, P3, (B, 1, 15, 64, 64)),
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


@pytest.mark.parametrize("case", LAYERS, ids=[c[0].split(" ")[0] for c in LAYERS])
def test_layer_at_b70(dev, case):
    from dcvgan_amd import ops
    name, tr, cin, cout, k, s, p, xs = case
    g = torch.Generator(device=dev).manual_seed(11)
    xd, xbig = guarded(torch.randn(xs, device=dev, generator=g))
    xd.requires_grad_(True)
    wd, wbig = guarded(torch.randn(((cin, cout) if tr else (cout, cin)) + k, device=dev, generator=g) * 0.05)
    wd.requires_grad_(True)
    y = ops.conv(xd, wd, ops.conv_geom(wd, s, p, tr))
    cotd, cbig = guarded(torch.randn(y.shape, device=dev, generator=g))
    gx, gw = torch.autograd.grad((y * cotd).sum(), [xd, wd])
    assert bool(torch.isfinite(y).all() and torch.isfinite(gx).all() and torch.isfinite(gw).all()), name        # nothing outside the operands was read
    assert margins_intact(xbig, xd.numel()) and margins_intact(wbig, wd.numel()) and margins_intact(cbig, cotd.numel()), name            # ... or written
    x, w, cot = xd.detach().cpu().requires_grad_(True), wd.detach().cpu().requires_grad_(True), cotd.cpu()
    fn = F.conv_transpose2d if tr else (F.conv3d if len(k) == 3 else F.conv2d)
    y_ref = fn(x, w, None, s, p)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert rel(y, y_ref) < 1e-5, name
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot).sum(), [x, w])
    assert rel(gx, gx_ref) < 1e-5, name
    assert rel(gw, gw_ref) < 2e-5, name     # sums over up to 4.6 M positions: the host's own fp32 sum is the looser side


@pytest.mark.parametrize("shape,slope", [((Fr, 64, 64, 64), 0.0), ((B, 128, 10, 16, 16), 0.2)], ids=["cgen.up5.bn_relu", "vdis.2.bn_lrelu"])
def test_batchnorm_act_at_b70(dev, shape, slope):
    from dcvgan_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    C = shape[1]
    xd = (torch.randn(shape, device=dev, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gam = (torch.rand(C, device=dev, generator=g) + 0.5).requires_grad_(True); bet = (torch.randn(C, device=dev, generator=g) * 0.1).requires_grad_(True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y = ops.bn_act(xd, gam, bet, rm, rv, True, ops.ACT_LEAKY, slope)
    cotd = torch.randn(shape, device=dev, generator=g)
    gx, gg, gb = torch.autograd.grad((y * cotd).sum(), [xd, gam, bet])
    x, ga, be = xd.detach().cpu().requires_grad_(True), gam.detach().cpu().requires_grad_(True), bet.detach().cpu().requires_grad_(True)
    rm_r, rv_r = torch.zeros(C), torch.ones(C)
    y_ref = F.leaky_relu(F.batch_norm(x, rm_r, rv_r, ga, be, True, 0.1, 1e-5), slope)
    gx_ref, gg_ref, gb_ref = torch.autograd.grad((y_ref * cotd.cpu()).sum(), [x, ga, be])
    assert rel(y, y_ref) < 1e-5 and rel(rm, rm_r) < 1e-5 and rel(rv, rv_r) < 1e-5
    assert rel(gx, gx_ref) < 1e-4 and rel(gg, gg_ref) < 1e-4 and rel(gb, gb_ref) < 1e-4


def test_batch_split_identity(dev):
    from dcvgan_amd import trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import InjectedRng
    cfg = CONFIGS["isogd-depth"]
    torch.manual_seed(77)
    models = trainer.build_models(cfg, dev)
    g = torch.Generator(device=dev).manual_seed(3)
    # non-trivial running statistics, then eval mode: every sample is processed on its own
    for m in models.values():
        for mod in m.modules():
            if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                mod.running_mean.copy_(torch.randn(mod.num_features, device=dev, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.num_features, device=dev, generator=g) + 0.5)
        m.eval()
    cgen, idis, vdis, gdis = models["cgen"], models["idis"], models["vdis"], models["gdis"]
    xg70 = torch.rand(B, 16, 1, 64, 64, device=dev, generator=g).mul(2).sub(1).permute(0, 2, 1, 3, 4)   # ggen's output layout
    t = 5

    def draws(n):
        # the same random numbers for sample i whatever the batch: z of cgen, then the Noise layers of idis / vdis
        gg = torch.Generator(device=dev).manual_seed(9)
        full = [("normal", torch.randn((B,) + s, device=dev, generator=gg)) for s in
                ((10,), (1, 64, 64), (3, 64, 64), (64, 32, 32), (128, 16, 16), (256, 8, 8),          # cgen z; idis stems, trunk
                 (64, 13, 32, 32), (128, 10, 16, 16), (256, 7, 8, 8))]                                 # vdis trunk (its stems have no Noise)
        return [(k, v[:n].contiguous()) for k, v in full]

    def run(n):
        for m in models.values():
            m.zero_grad()
        r = InjectedRng(draws(n))
        for m in models.values():
            m._rng = r
        xg = xg70[:n].detach().requires_grad_(True)
        xc = cgen.forward_videos(xg)
        yi, yv, yg = idis(xg[:, :, t], xc[:, :, t]), vdis(xg, xc), gdis(xg, xc)
        assert r.pos == len(r.log)
        tot = 0
        for y in (yi, yv, yg):
            cot = torch.cos(torch.arange(y[:16].numel(), device=dev, dtype=torch.float32) * 0.3).view(y[:16].shape)
            tot = tot + (y[:16] * cot).sum()          # the cotangent is zero for rows >= 16
        tot.backward()
        grads = {(mn, k): p.grad.detach().clone() for mn, m in models.items() if mn != "ggen" for k, p in m.named_parameters()}
        return xc.detach()[:16].clone(), [y.detach()[:16].clone() for y in (yi, yv, yg)], xg.grad[:16].clone(), grads

    xc16, ys16, gx16, gr16 = run(16)
    xc70, ys70, gx70, gr70 = run(B)
    assert rel(xc70, xc16) < 1e-5
    for a, b in zip(ys70, ys16):
        assert rel(a, b) < 1e-5
    # gradients: 1e-3 (north_star).  The two passes run different kernel variants, so a LeakyReLU pre-activation within
    # rounding of zero may pick different branches (measured: 1.2e-4 on one tensor, ~1e-6 on the rest)
    assert rel(gx70, gx16) < 1e-3
    for key, gref in gr16.items():
        assert rel(gr70[key], gref) < 1e-3, key
