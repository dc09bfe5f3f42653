"""GPU: every HIP op against the plain PyTorch fp32 CPU op it replaces (the arithmetic
the reference delegates to torch.nn).  Tolerance: 1e-3 relative (BASELINE.json
north_star), in practice ~1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-3


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    from dcvgan_amd import native
    native.lib()
    return torch.device("cuda:0")


# (name, transposed, dims, Cin, Cout, k, s, p, input spatial, N)
CONVS = [
    ("conv2d_4s2p1", False, 2, 5, 7, 4, 2, 1, (16, 16), 3),
    ("conv2d_4s2p1_wide", False, 2, 40, 72, 4, 2, 1, (8, 8), 5),
    ("conv2d_4s2p1_to1x1", False, 2, 12, 9, 4, 2, 1, (2, 2), 6),
    ("conv2d_3s1p1", False, 2, 2, 6, 3, 1, 1, (12, 12), 2),
    ("conv2d_head", False, 2, 24, 1, 4, 2, 1, (8, 8), 4),
    ("convT2d_4s2p1", True, 2, 6, 5, 4, 2, 1, (8, 8), 3),
    ("convT2d_4s2p1_wide", True, 2, 70, 33, 4, 2, 1, (4, 4), 3),
    ("convT2d_4s2p1_from1x1", True, 2, 10, 8, 4, 2, 1, (1, 1), 5),
    ("convT2d_4s1p0_latent", True, 2, 9, 20, 4, 1, 0, (1, 1), 7),
    ("convT2d_3s1p1", True, 2, 8, 3, 3, 1, 1, (10, 10), 2),
    ("conv3d_4s122", False, 3, 3, 6, 4, (1, 2, 2), (0, 1, 1), (7, 8, 8), 2),
    ("conv3d_4s122_wide", False, 3, 36, 40, 4, (1, 2, 2), (0, 1, 1), (5, 4, 4), 2),
    ("conv3d_head", False, 3, 20, 1, 4, (1, 2, 2), (0, 1, 1), (7, 8, 8), 3),
    # patch-staged operand (16-byte granules of raw rows): all three tile shapes, ragged last tile, tiny planes
    ("conv2d_4s2p1_32_oc40", False, 2, 8, 40, 4, 2, 1, (32, 32), 3),
    ("conv2d_4s2p1_32_oc130", False, 2, 6, 130, 4, 2, 1, (32, 32), 2),
    ("conv2d_4s2p1_16_oc24", False, 2, 12, 24, 4, 2, 1, (16, 16), 5),
    ("conv2d_4s2p1_8_oc72", False, 2, 20, 72, 4, 2, 1, (8, 8), 7),
    ("convT2d_4s2p1_16_oc36", True, 2, 12, 36, 4, 2, 1, (16, 16), 3),
    ("convT2d_4s2p1_32_oc68", True, 2, 8, 68, 4, 2, 1, (32, 32), 2),
    ("convT2d_4s2p1_8_oc132", True, 2, 16, 132, 4, 2, 1, (8, 8), 5),
    ("conv3d_4s122_16_oc70", False, 3, 8, 70, 4, (1, 2, 2), (0, 1, 1), (6, 16, 16), 2),
    ("conv3d_4s122_32_oc36", False, 3, 8, 36, 4, (1, 2, 2), (0, 1, 1), (5, 32, 32), 2),
    # LDS-DMA weight-gradient kernel (128-channel tiles), dense operand staged in 16-byte granules
    ("conv2d_4s2p1_wgrad_dma", False, 2, 16, 128, 4, 2, 1, (16, 16), 9),
    ("convT2d_4s2p1_wgrad_dma", True, 2, 128, 8, 4, 2, 1, (8, 8), 21),
    ("conv3d_4s122_wgrad_dma", False, 3, 8, 128, 4, (1, 2, 2), (0, 1, 1), (6, 16, 16), 3),
    ("conv2d_4s2p1_wgrad_dma64", False, 2, 16, 64, 4, 2, 1, (16, 16), 9),
    ("conv3d_4s122_wgrad_dma64", False, 3, 32, 64, 4, (1, 2, 2), (0, 1, 1), (5, 8, 8), 4),
    # thin patch form (OC <= 4, >= 65536 positions): 3x3 heads, stride-2 stems' data gradients, 3-D depth-step
    ("convT2d_3s1p1_thin3", True, 2, 12, 3, 3, 1, 1, (64, 64), 16),
    ("conv2d_3s1p1_from1", False, 2, 1, 8, 3, 1, 1, (64, 64), 17),
    ("convT2d_4s2p1_to1", True, 2, 8, 1, 4, 2, 1, (32, 32), 65),
    ("conv2d_4s2p1_head_big", False, 2, 8, 2, 4, 2, 1, (64, 64), 64),
    ("conv3d_4s122_stem1", False, 3, 1, 8, 4, (1, 2, 2), (0, 1, 1), (9, 64, 64), 8),
    ("conv3d_4s122_stem3", False, 3, 3, 12, 4, (1, 2, 2), (0, 1, 1), (7, 64, 64), 11),
    # thin_quad_kernel (OC <= 4 scatter-form 4x4 / stride 2 on 32 -> 64 wide rows): 2-D stem data gradient, 2-channel head
    ("conv2d_4s2p1_stem2", False, 2, 2, 8, 4, 2, 1, (64, 64), 4),
    ("convT2d_4s2p1_to2", True, 2, 10, 2, 4, 2, 1, (32, 32), 6),
    # thin_wgrad3_kernel (1 / 2 gathered channels, 3x3, 64-wide rows): the colour generator's stem
    ("conv2d_3s1p1_stem1_wgrad", False, 2, 1, 64, 3, 1, 1, (64, 64), 5),
    ("conv2d_3s1p1_stem2_wgrad", False, 2, 2, 96, 3, 1, 1, (64, 64), 3),
]


@pytest.mark.parametrize("case", CONVS, ids=[c[0] for c in CONVS])
@pytest.mark.parametrize("strided", [False, True])
def test_conv_fwd_bwd(dev, case, strided):
    from dcvgan_amd import ops
    name, tr, nd, cin, cout, k, s, p, sp, n = case
    g = torch.Generator().manual_seed(hash(name) % 10000)
    s_t = (s,) * nd if isinstance(s, int) else s
    p_t = (p,) * nd if isinstance(p, int) else p
    wshape = ((cin, cout) if tr else (cout, cin)) + (k,) * nd
    w = torch.randn(wshape, generator=g) * 0.2
    if strided:  # channel-last memory viewed as NC..: the generators' output layout
        x = torch.randn((n,) + sp + (cin,), generator=g).permute(0, nd + 1, *range(1, nd + 1))
    else:
        x = torch.randn((n, cin) + sp, generator=g)
    x = x.requires_grad_(True); w = w.requires_grad_(True)
    fn = {(False, 2): F.conv2d, (False, 3): F.conv3d, (True, 2): F.conv_transpose2d}[(tr, nd)]
    y_ref = fn(x, w, None, s_t, p_t)
    cot = torch.randn(y_ref.shape, generator=g)
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot).sum(), [x, w])

    xd = x.detach().to(dev)
    if strided:
        xd = x.detach().permute(0, *range(2, nd + 2), 1).contiguous().to(dev).permute(0, nd + 1, *range(1, nd + 1))
        assert (not xd.is_contiguous()) or cin == 1 or all(v == 1 for v in sp)
    xd.requires_grad_(True)
    wd = w.detach().to(dev).requires_grad_(True)
    geom = ops.conv_geom(wd, s_t, p_t, tr)
    y = ops.conv(xd, wd, geom)
    assert y.shape == y_ref.shape
    assert rel(y, y_ref) < TOL
    gx, gw = torch.autograd.grad((y * cot.to(dev)).sum(), [xd, wd])
    assert rel(gx, gx_ref) < TOL
    assert rel(gw, gw_ref) < TOL


# a size at which the ragged split-K of the LDS-DMA kernel triggers (4 classes x 270 position tiles = 1080 workgroups: one round of 1024 + 56): the tiles before the
# split point accumulate / gate in the GEMM epilogue, the split tail in splitk_reduce_kernel — both must honour the slice already holding a gradient
RAGGED_CASE = ("conv2d_4s2p1_ragged_tail", False, 2, 128, 512, 4, 2, 1, (16, 16), 540)


@pytest.mark.parametrize("case", [RAGGED_CASE] + [c for c in CONVS if c[0] in ("conv2d_4s2p1", "conv2d_4s2p1_wide", "conv2d_4s2p1_32_oc40", "conv2d_4s2p1_32_oc130",
                                                                "conv2d_4s2p1_8_oc72", "conv2d_3s1p1", "conv3d_4s122_16_oc70", "convT2d_4s2p1_16_oc36", "conv2d_4s2p1_stem2")],
                         ids=lambda c: c[0])
def test_conv_backward_data_accumulates_into_a_slice(dev, case):
    """dcv_conv_backward_data(accumulate = 1): dx += conv^T(dy, w) where dx is a channel slice of a wider buffer that
    already holds another gradient — how the data gradients of the U-Net's down blocks join the concat's gradient
    slices (ops.GradSlot) without a separate add."""
    import ctypes as C
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    name, tr, nd, cin, cout, k, s, p, sp, n = case
    g = torch.Generator().manual_seed(hash(name) % 1000 + 7)
    s_t = (s,) * nd if isinstance(s, int) else s
    p_t = (p,) * nd if isinstance(p, int) else p
    w = torch.randn(((cin, cout) if tr else (cout, cin)) + (k,) * nd, generator=g) * 0.2
    x = torch.randn((n, cin) + sp, generator=g, requires_grad=True)
    fn = {(False, 2): F.conv2d, (False, 3): F.conv3d, (True, 2): F.conv_transpose2d}[(tr, nd)]
    y = fn(x, w, None, s_t, p_t)
    dy = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad((y * dy).sum(), [x])
    wide = torch.randn((n, cin + 5) + sp, generator=g)            # the buffer: 5 foreign channels, then the slice
    want = wide.clone(); want[:, 5:] += gx
    wide_d, dy_d, w_d = wide.to(dev), dy.to(dev), w.to(dev)
    dx = wide_d[:, 5:]
    geom = ops.conv_geom(w_d, s_t, p_t, tr)
    dxd, dyd = dims5(dx), dims5(dy_d)
    L = N.lib()
    need = L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(dxd), C.byref(dyd), 1)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    N.check(L.dcv_conv_backward_data(C.byref(geom), ptr(dy_d), C.byref(dyd), ptr(w_d), ptr(dx), C.byref(dxd), 1, None, ptr(ws), need, stream_ptr()), "accumulate")
    if name.endswith("ragged_tail"):
        L.dcv_debug_last_kernel.restype = C.c_char_p
        assert "ragged split-K" in L.dcv_debug_last_kernel().decode(), L.dcv_debug_last_kernel().decode()
    assert rel(wide_d, want) < 1e-5
    assert torch.equal(wide_d[:, :5].cpu(), wide[:, :5])          # the neighbouring channels are untouched
    # dcv_conv_backward_data_gated: the same, then times the LeakyReLU derivative read off the conv's own input, which
    # sits in the matching slice of a second buffer of the same layout (Inconv -> DownBlock 0 of the U-Net)
    xin = torch.randn(wide.shape, generator=g)
    want2 = wide.clone(); want2[:, 5:] = (wide[:, 5:] + gx) * torch.where(xin[:, 5:] > 0, 1.0, 0.01)
    wide2_d, xin_d = wide.to(dev), xin.to(dev)
    dx2, x2 = wide2_d[:, 5:], xin_d[:, 5:]
    xd2 = dims5(x2)
    rc = L.dcv_conv_backward_data_gated(C.byref(geom), ptr(dy_d), C.byref(dyd), ptr(w_d), ptr(dx2), C.byref(dxd), 1, ptr(x2), C.byref(xd2),
                                        ops.ACT_LEAKY, 0.01, None, ptr(ws), need, stream_ptr())
    if cin <= 4:
        assert rc == N.DCV_EUNSUPPORTED        # thin kernels: the caller falls back to the two separate steps
    else:
        assert rc == 0, L.dcv_last_error()
        assert rel(wide2_d, want2) < 1e-5 and torch.equal(wide2_d[:, :5].cpu(), wide[:, :5])


def test_packed_weight_cache_follows_the_weights(dev):
    """The K-major packed copy is cached per (weight tensor, autograd version): in-place updates that bump the version
    (optimiser steps, copy_, load_state_dict) are seen; an edit through `.data` is not, until the cache is invalidated."""
    from dcvgan_amd import ops, util
    g = torch.Generator().manual_seed(9)
    conv = torch.nn.Conv2d(16, 64, 4, 2, 1, bias=False).to(dev)
    x = torch.randn(3, 16, 32, 32, generator=g).to(dev)
    geom = ops.conv_geom(conv.weight, (2, 2), (1, 1), False)
    with torch.no_grad():
        y0 = ops.conv(x, conv.weight, geom).clone()
        conv.weight.mul_(2.0)                                   # bumps the version
        assert rel(ops.conv(x, conv.weight, geom), 2 * y0) < 1e-6
        conv.weight.data.mul_(0.5)                              # does not
        ops.invalidate_packed_weights(conv)
        assert rel(ops.conv(x, conv.weight, geom), y0) < 1e-6
        w_before = conv.weight.detach().clone()
        conv.apply(util.init_weights)                           # `.data.normal_` + an explicit version bump
        y2 = ops.conv(x, conv.weight, geom)
        assert not torch.equal(conv.weight, w_before)
        assert rel(y2, F.conv2d(x.cpu(), conv.weight.detach().cpu(), None, 2, 1)) < 1e-5


def test_conv_fused_act(dev):
    from dcvgan_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 8, 8, generator=g, requires_grad=True); w = torch.randn(5, 3, 4, 4, generator=g, requires_grad=True)
    y_ref = F.leaky_relu(F.conv2d(x, w, None, 2, 1), 0.2)
    cot = torch.randn(y_ref.shape, generator=g)
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot).sum(), [x, w])
    xd, wd = x.detach().to(dev).requires_grad_(True), w.detach().to(dev).requires_grad_(True)
    y = ops.conv(xd, wd, ops.conv_geom(wd, (2, 2), (1, 1), False), ops.ACT_LEAKY, 0.2)
    gx, gw = torch.autograd.grad((y * cot.to(dev)).sum(), [xd, wd])
    assert rel(y, y_ref) < TOL and rel(gx, gx_ref) < TOL and rel(gw, gw_ref) < TOL


@pytest.mark.parametrize("shape", [(3, 6, 8, 8), (2, 5, 4, 6, 6), (4, 7, 1, 1), (2, 3, 5, 3, 3)])
@pytest.mark.parametrize("mode", ["train", "eval", "dropout"])
def test_bn_act(dev, shape, mode):
    from dcvgan_amd import ops
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(shape, generator=g) * 1.7 + 0.4).requires_grad_(True)
    Cn = shape[1]
    gamma = (torch.rand(Cn, generator=g) + 0.5).requires_grad_(True); beta = torch.randn(Cn, generator=g).requires_grad_(True)
    rm, rv = torch.randn(Cn, generator=g) * 0.1, torch.rand(Cn, generator=g) + 0.5
    rm_d, rv_d = rm.clone().to(dev), rv.clone().to(dev)
    training = mode != "eval"
    mask = None
    if mode == "dropout":
        mask = (torch.rand(shape[0], Cn, generator=g) > 0.5).float().view(shape[0], Cn, *([1] * (len(shape) - 2))) * 2.0
    h = F.batch_norm(x, rm, rv, gamma, beta, training, 0.1, 1e-5)
    if mask is not None:
        h = h * mask
    y_ref = F.leaky_relu(h, 0.2)
    cot = torch.randn(shape, generator=g)
    g_ref = torch.autograd.grad((y_ref * cot).sum(), [x, gamma, beta])
    xd = x.detach().to(dev).requires_grad_(True)
    gd, bd = gamma.detach().to(dev).requires_grad_(True), beta.detach().to(dev).requires_grad_(True)
    md = None if mask is None else mask.reshape(shape[0], Cn, 1, 1).to(dev)
    y = ops.bn_act(xd, gd, bd, rm_d, rv_d, training, ops.ACT_LEAKY, 0.2, md)
    assert rel(y, y_ref) < TOL
    got = torch.autograd.grad((y * cot.to(dev)).sum(), [xd, gd, bd])
    for a, b in zip(got, g_ref):
        assert rel(a, b) < TOL
    assert rel(rm_d, rm) < 1e-5 and rel(rv_d, rv) < 1e-5


def test_act_cat_diff_noise(dev):
    from dcvgan_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 6, 8, 8, generator=g, requires_grad=True)
    xd = x.detach().to(dev).requires_grad_(True)
    for kind, ref in ((ops.ACT_TANH, torch.tanh), (ops.ACT_LEAKY, lambda t: F.leaky_relu(t, 0.01))):
        y_ref = ref(x); cot = torch.randn(y_ref.shape, generator=g)
        (gr,) = torch.autograd.grad((y_ref * cot).sum(), [x])
        y = ops.act(xd, kind, 0.01)
        (gg,) = torch.autograd.grad((y * cot.to(dev)).sum(), [xd])
        assert rel(y, y_ref) < TOL and rel(gg, gr) < TOL
    # temporal difference
    y_ref = x[:, :, 1:] - x[:, :, :-1]; cot = torch.randn(y_ref.shape, generator=g)
    (gr,) = torch.autograd.grad((y_ref * cot).sum(), [x])
    y = ops.temporal_diff(xd)
    (gg,) = torch.autograd.grad((y * cot.to(dev)).sum(), [xd])
    assert rel(y, y_ref) < 1e-6 and rel(gg, gr) < 1e-6
    # cat + injected noise
    z = torch.randn(2, 4, 6, 8, 8, generator=g, requires_grad=True); zd = z.detach().to(dev).requires_grad_(True)
    nz = torch.randn(2, 7, 6, 8, 8, generator=g)
    y_ref = torch.cat([x, z], 1) + 0.3 * nz; cot = torch.randn(y_ref.shape, generator=g)
    gr = torch.autograd.grad((y_ref * cot).sum(), [x, z])
    y = ops.noise_add(ops.cat_channels(xd, zd), 0.3, nz.to(dev))
    gg = torch.autograd.grad((y * cot.to(dev)).sum(), [xd, zd])
    assert rel(y, y_ref) < 1e-6 and rel(gg[0], gr[0]) < 1e-6 and rel(gg[1], gr[1]) < 1e-6


def test_device_rng_moments(dev):
    from dcvgan_amd import ops
    z = ops.normal((1 << 20,), dev, 1234, 0)
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1) < 5e-3
    z2 = ops.normal((1 << 20,), dev, 1234, 1)
    assert abs((z * z2).mean().item()) < 5e-3  # streams differ
    assert torch.equal(z, ops.normal((1 << 20,), dev, 1234, 0))  # reproducible
    x = torch.zeros(4, 3, 5, 16, 16, device=dev)
    y = ops.noise_add(x, 0.5, None, 7, 3)
    assert abs(y.std().item() - 0.5) < 2e-2
    ystr = ops.noise_add(x.permute(0, 2, 1, 3, 4).contiguous().permute(0, 2, 1, 3, 4), 0.5, None, 7, 3)
    assert torch.equal(y, ystr)  # the draw depends on the logical index, not on strides
    m = ops.dropout2d_mask(512, 256, 0.5, dev, 9, 0)
    assert set(m.unique().tolist()) == {0.0, 2.0} and abs(m.mean().item() - 1.0) < 2e-2


def test_normal_many_is_the_stack_of_single_draws(dev):
    """dcv_normal_fill_many (the generator's per-frame motion noise in one launch): draw j = the single draw at stream position offset + j, bit for bit — odd sizes
    included (a draw's last Philox block is cut, the next draw starts a new one) — and PhiloxRng.normal_many leaves the stream where the single draws would."""
    from dcvgan_amd import ops, rng
    for shape, count in (((100, 10), 16), ((7, 3), 5), ((1,), 3), ((64, 10), 1)):
        many = ops.normal_many(count, shape, dev, 4321, 17)
        assert many.shape == (count,) + shape
        for j in range(count):
            assert torch.equal(many[j], ops.normal(shape, dev, 4321, 17 + j)), (shape, j)
    a, b = rng.PhiloxRng(99), rng.PhiloxRng(99)
    ya = a.normal_many(16, (100, 10), dev)
    yb = torch.stack([b.normal((100, 10), dev) for _ in range(16)], 0)
    assert torch.equal(ya, yb)
    assert torch.equal(a.normal((5,), dev), b.normal((5,), dev))      # same stream position afterwards


@pytest.mark.parametrize("kind", range(5))
def test_gan_loss(dev, kind):
    from dcvgan_amd import ops
    g = torch.Generator().manual_seed(kind)
    y = (torch.randn(6, 4, 4, 4, generator=g) * 3).requires_grad_(True)
    ref = [lambda t: F.binary_cross_entropy_with_logits(t, torch.ones_like(t), reduction="sum") / t.numel(),
           lambda t: F.binary_cross_entropy_with_logits(t, torch.zeros_like(t), reduction="sum") / t.numel(),
           lambda t: F.relu(1 - t).mean(), lambda t: F.relu(1 + t).mean(), lambda t: F.softplus(-t).mean()][kind]
    v_ref = ref(y); (g_ref,) = torch.autograd.grad(v_ref * 1.5, [y])
    yd = y.detach().to(dev).requires_grad_(True)
    v = ops.gan_loss(yd, kind); (gg,) = torch.autograd.grad(v * 1.5, [yd])
    assert abs(v.item() - v_ref.item()) < 1e-5 * max(1, abs(v_ref.item())) and rel(gg, g_ref) < 1e-5


def test_gru_sequence(dev):
    from dcvgan_amd import ops
    g = torch.Generator().manual_seed(2)
    T, B, dm = 16, 5, 10
    cell = torch.nn.GRUCell(dm, dm)
    e = torch.randn(T, B, dm, generator=g); h0 = torch.randn(B, dm, generator=g)
    h = h0; hs = []
    for t in range(T):
        h = cell(e[t], h); hs.append(h)
    out_ref = torch.stack(hs, 1); cot = torch.randn(out_ref.shape, generator=g)
    params = [cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh]
    g_ref = torch.autograd.grad((out_ref * cot).sum(), params)
    pd = [p.detach().to(dev).requires_grad_(True) for p in params]
    out = ops.gru_sequence(e.to(dev), h0.to(dev), *pd)
    got = torch.autograd.grad((out * cot.to(dev)).sum(), pd)
    assert rel(out, out_ref) < 1e-5
    for a, b in zip(got, g_ref):
        assert rel(a, b) < 1e-4


def test_out_slices_and_copy_free_concat(dev):
    """conv(+act) and bn_act writing straight into the two channel slices of a concat buffer, joined
    without a copy: forward and all gradients equal torch's cat path.  (Regression: the fused-activation
    backward once described its dense scratch with the strided output's strides.)"""
    from dcvgan_amd import ops
    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 4, 16, 16, generator=g, requires_grad=True)
    w1 = (torch.randn(6, 4, 4, 4, generator=g) * 0.2).requires_grad_(True)
    w2 = (torch.randn(5, 4, 4, 4, generator=g) * 0.2).requires_grad_(True)
    gam = (torch.rand(5, generator=g) + 0.5).requires_grad_(True); bet = torch.randn(5, generator=g).requires_grad_(True)
    a_ref = F.leaky_relu(F.conv2d(x, w1, None, 2, 1), 0.2)
    b_ref = F.relu(F.batch_norm(F.conv2d(x, w2, None, 2, 1), None, None, gam, bet, True, 0.1, 1e-5))
    y_ref = torch.cat([a_ref, b_ref], 1)
    cot = torch.randn(y_ref.shape, generator=g)
    ref = torch.autograd.grad((y_ref * cot).sum(), [x, w1, w2, gam, bet])
    xd = x.detach().to(dev).requires_grad_(True)
    p = [t.detach().to(dev).requires_grad_(True) for t in (w1, w2, gam, bet)]
    cb = ops.ConcatBuffer(3, 6, 5, (8, 8), dev)
    a = ops.conv(xd, p[0], ops.conv_geom(p[0], (2, 2), (1, 1), False), ops.ACT_LEAKY, 0.2, out=cb.first)
    h = ops.conv(xd, p[1], ops.conv_geom(p[1], (2, 2), (1, 1), False))
    b = ops.bn_act(h, p[2], p[3], None, None, True, ops.ACT_LEAKY, 0.0, out=cb.second)
    y = cb.join(a, b)
    assert y.data_ptr() == cb.buf.data_ptr() and rel(y, y_ref) < TOL
    got = torch.autograd.grad((y * cot.to(dev)).sum(), [xd] + p)
    for u, v in zip(got, ref):
        assert rel(u, v) < TOL


def test_wgrad_long_reduction_accuracy(dev):
    """Weight gradient over 16384 positions per split-free chain (128x128 tile => the LDS-DMA wgrad kernel
    with two-level accumulation): error against an fp64 reference must be in the class of torch's own
    fp32 CPU result (and far inside the 1e-3 bar)."""
    from dcvgan_amd import ops
    g = torch.Generator().manual_seed(77)
    x = torch.randn(64, 128, 32, 32, generator=g)
    w = (torch.randn(128, 128, 4, 4, generator=g) * 0.05)
    dy = torch.randn(64, 128, 16, 16, generator=g)
    w64 = w.double().requires_grad_(True)
    (F.conv2d(x.double(), w64, None, 2, 1) * dy.double()).sum().backward()
    w32 = w.clone().requires_grad_(True)
    (F.conv2d(x, w32, None, 2, 1) * dy).sum().backward()
    xd = x.to(dev); wd = w.to(dev).requires_grad_(True)
    y = ops.conv(xd, wd, ops.conv_geom(wd, (2, 2), (1, 1), False))
    (y * dy.to(dev)).sum().backward()
    e_hip = rel(wd.grad, w64.grad); e_cpu = rel(w32.grad, w64.grad)
    assert e_hip < 1e-5 and e_hip < 5 * max(e_cpu, 2e-7), (e_hip, e_cpu)


def test_conv_bn_fused_statistics(dev):
    """conv -> BatchNorm pair: the batch statistics taken from the GEMM epilogue's per-tile partial sums give the
    same BN output, saved statistics, running statistics and gradients as the separate statistics pass."""
    from dcvgan_amd import ops
    g = torch.Generator().manual_seed(11)
    nfused = 0
    # small problems run split-K (no fused sums: the BN op must then fall back by itself); the 96-sample one does not
    for tr, cin, cout, sp, n in ((False, 16, 40, (64, 64), 96), (True, 24, 130, (8, 8), 7), (False, 8, 36, (5, 16, 16), 2)):
        nd = len(sp)
        k = 4
        s_t = (2, 2) if nd == 2 else (1, 2, 2)
        p_t = (1, 1) if nd == 2 else (0, 1, 1)
        wshape = ((cin, cout) if tr else (cout, cin)) + (k,) * nd
        w0 = (torch.randn(wshape, generator=g) * 0.1).to(dev)
        x0 = torch.randn((n, cin) + sp, generator=g).to(dev)
        gamma0 = (torch.rand(cout, generator=g) + 0.5).to(dev); beta0 = torch.randn(cout, generator=g).to(dev)
        res = []
        for fused in (True, False):
            x = x0.clone().requires_grad_(True); w = w0.clone().requires_grad_(True)
            gamma = gamma0.clone().requires_grad_(True); beta = beta0.clone().requires_grad_(True)
            rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
            geom = ops.conv_geom(w, s_t, p_t, tr)
            box = [] if fused else None
            y = ops.conv(x, w, geom, bn_stats=box)
            nfused += bool(box)
            z = ops.bn_act(y, gamma, beta, rm, rv, True, ops.ACT_LEAKY, 0.2, partials=box[0] if box else None)
            cot = torch.cos(torch.arange(z.numel(), device=dev, dtype=torch.float32)).view(z.shape)
            gr = torch.autograd.grad((z * cot).sum(), [x, w, gamma, beta])
            res.append([z.detach(), rm, rv] + [t for t in gr])
        for a, b in zip(*res):
            assert rel(a.cpu(), b.cpu()) < 1e-5
    assert nfused >= 1


def test_conv_bn_when_the_conv_splits_its_tail(dev):
    """A conv -> BatchNorm pair whose convolution runs the ragged split-K (550 position tiles x 2 channel tiles = 1100 workgroups: one round + 76): the
    tail's outputs come out of splitk_reduce_kernel, the epilogue's fused BatchNorm sums are not produced (nparts = 0) and the BN op takes its own statistics —
    against torch on the host: output, running statistics and all four gradients."""
    import ctypes as C
    from dcvgan_amd import native as N, ops
    g = torch.Generator().manual_seed(21)
    n, cin, cout = 1100, 128, 256
    w0 = torch.randn(cout, cin, 4, 4, generator=g) * 0.05
    x0 = torch.randn(n, cin, 16, 16, generator=g)
    gamma0 = torch.rand(cout, generator=g) + 0.5; beta0 = torch.randn(cout, generator=g) * 0.1
    x, w, gamma, beta = (t.to(dev).requires_grad_(True) for t in (x0, w0, gamma0, beta0))
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    box = []
    y = ops.conv(x, w, ops.conv_geom(w, (2, 2), (1, 1), False), bn_stats=box)
    L = N.lib(); L.dcv_debug_last_kernel.restype = C.c_char_p
    assert "ragged split-K" in L.dcv_debug_last_kernel().decode(), L.dcv_debug_last_kernel().decode()
    assert box == []                                             # no fused sums from a split convolution
    z = ops.bn_act(y, gamma, beta, rm, rv, True, ops.ACT_NONE, 0.0)    # no activation: with 18 M elements behind a 2048-term sum a (Leaky)ReLU adds ~5e-4 of branch lottery to dx
    cot = torch.cos(torch.arange(z.numel(), dtype=torch.float32) * 0.37).view(z.shape)
    gr = torch.autograd.grad((z * cot.to(dev)).sum(), [x, w, gamma, beta])
    xr, wr, gar, ber = (t.clone().requires_grad_(True) for t in (x0, w0, gamma0, beta0))
    rm_r, rv_r = torch.zeros(cout), torch.ones(cout)
    z_ref = F.batch_norm(F.conv2d(xr, wr, None, 2, 1), rm_r, rv_r, gar, ber, True, 0.1, 1e-5)
    gr_ref = torch.autograd.grad((z_ref * cot).sum(), [xr, wr, gar, ber])
    assert rel(z, z_ref) < 1e-5 and rel(rm, rm_r) < 1e-5 and rel(rv, rv_r) < 1e-5
    for a, b in zip(gr, gr_ref):
        assert rel(a, b) < 1e-4


def test_thin_wgrad_ragged_slabs(dev):
    """thin_wgrad3_kernel walks `pps` images per slab; 1025 images in slabs of 2 leave a last slab of one image."""
    import ctypes as C
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    g = torch.Generator().manual_seed(5)
    n = 1025
    x = torch.randn(n, 1, 64, 64, generator=g)
    w = (torch.randn(64, 1, 3, 3, generator=g) * 0.2).requires_grad_(True)
    y_ref = F.conv2d(x, w, None, 1, 1)
    cot = torch.randn(y_ref.shape, generator=g)
    (gw_ref,) = torch.autograd.grad((y_ref * cot).sum(), [w])
    x_d, dy_d = x.to(dev), cot.to(dev)
    dw = torch.full(w.shape, float("nan"), device=dev)
    geom = ops.conv_geom(dw, (1, 1), (1, 1), False)
    xd, dyd = dims5(x_d), dims5(dy_d)
    L = N.lib()
    need = L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(xd), C.byref(dyd), 2)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    N.check(L.dcv_conv_backward_weight(C.byref(geom), ptr(x_d), C.byref(xd), ptr(dy_d), C.byref(dyd), ptr(dw), ptr(ws), need, stream_ptr()), "wgrad")
    assert "thin_wgrad3_kernel<1, 8> (513 slabs)" in L.dcv_debug_last_kernel().decode()
    assert rel(dw, gw_ref) < 1e-4
