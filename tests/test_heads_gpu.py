"""GPU: the discriminators' heads — Conv2d / Conv3d(C -> 1, 4x4(x4), stride (1,)2,2, padding (0,)1,1) on 8 x 8 planes (/root/reference/src/discriminator.py:
the last layer of ImageDiscriminator, VideoDiscriminator, GradientDiscriminator) — forward, data gradient and weight gradient on the head_* kernels, through the
C ABI, against torch's fp64 convolution on the host, inside NaN guard bands.  Plane counts that are not multiples of four (the kernels take four planes per
workgroup), channel counts that do not divide by the channel splits, one sample, the 2-D and the 3-D form, a fused LeakyReLU, accumulation into existing gradients,
an input that is a channel slice of a wider buffer; repeatability; and a neighbouring geometry (16 x 16 planes) that must take the generic path."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GUARD = 4096


def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    from dcvgan_amd import native
    native.lib()
    return torch.device("cuda:0")


def last_kernel():
    from dcvgan_amd import native as N
    L = N.lib()
    L.dcv_debug_last_kernel.restype = C.c_char_p
    return L.dcv_debug_last_kernel().decode()


def guarded(shape, dev, fill=None):
    n = 1
    for s_ in shape:
        n *= s_
    big = torch.full((n + 2 * GUARD,), float("nan"), device=dev)
    t = big[GUARD:GUARD + n].view(shape)
    if fill is not None:
        t.copy_(fill)
    return t, big, n


def intact(big, n):
    return bool(torch.isnan(big[:GUARD]).all() and torch.isnan(big[GUARD + n:]).all())


def conv_ref(x, w, nd):
    if nd == 1:
        return F.conv2d(x[:, :, 0], w[:, :, 0], None, 2, 1).unsqueeze(2)
    return F.conv3d(x, w, None, (1, 2, 2), (0, 1, 1))


# samples, channels, input planes per sample, depth taps
CASES = [(5, 256, 7, 4), (3, 128, 6, 4), (70, 256, 1, 1), (1, 48, 4, 4), (2, 256, 9, 4), (7, 100, 1, 1)]


@pytest.mark.parametrize("n,c,d,nd", CASES, ids=lambda v: str(v))
def test_head_three_passes(dev, n, c, d, nd):
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    L = N.lib()
    g_ = torch.Generator().manual_seed(3 + n + c + d + nd)
    x = torch.randn(n, c, d, 8, 8, generator=g_, dtype=torch.float64)
    w = torch.randn(1, c, nd, 4, 4, generator=g_, dtype=torch.float64) * 0.05
    od = d - nd + 1
    dy = torch.randn(n, 1, od, 4, 4, generator=g_, dtype=torch.float64)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = conv_ref(xr, wr, nd)
    gx, gw = torch.autograd.grad((y_ref * dy).sum(), [xr, wr])
    xd_, wd_, dyd_ = x.float().to(dev), w.float().to(dev), dy.float().to(dev)
    geom = ops.conv_geom(wd_, (1, 2, 2), (0, 1, 1), False)
    y, ybig, yn = guarded((n, 1, od, 4, 4), dev)
    xm, ym = dims5(xd_), dims5(y)
    need = max(L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(xm), C.byref(ym), wh) for wh in (0, 1, 2))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    # forward, with a fused LeakyReLU on the 3-D cases
    act, slope = (ops.ACT_LEAKY, 0.2) if nd == 4 else (ops.ACT_NONE, 0.0)
    N.check(L.dcv_conv_forward(C.byref(geom), ptr(xd_), C.byref(xm), ptr(wd_), ptr(y), C.byref(ym), act, slope, None, ptr(ws), need, stream_ptr()), "fwd")
    assert ("head_fwd_kernel" in last_kernel()) == (nd == 4), last_kernel()      # (the 2-D head's forward stays on the tile kernel: measured equal)
    want = F.leaky_relu(y_ref.detach(), 0.2) if nd == 4 else y_ref.detach()
    assert intact(ybig, yn) and rel(y, want) < 5e-6
    # data gradient, plain and accumulated
    dx, xbig, xn = guarded(tuple(x.shape), dev)
    N.check(L.dcv_conv_backward_data(C.byref(geom), ptr(dyd_), C.byref(ym), ptr(wd_), ptr(dx), C.byref(xm), 0, None, ptr(ws), need, stream_ptr()), "dgrad")
    assert ("head_dgrad_kernel" in last_kernel()) == (nd == 1), last_kernel()    # (the 3-D heads' data gradient stays on the tile kernel: measured faster there)
    assert intact(xbig, xn) and rel(dx, gx) < 5e-6
    old = torch.randn(x.shape, generator=g_)
    dx.copy_(old)
    N.check(L.dcv_conv_backward_data(C.byref(geom), ptr(dyd_), C.byref(ym), ptr(wd_), ptr(dx), C.byref(xm), 1, None, ptr(ws), need, stream_ptr()), "dgrad acc")
    assert intact(xbig, xn) and rel(dx, old.double() + gx) < 5e-6
    # weight gradient, plain and accumulated
    dw, wbig, wn = guarded(tuple(w.shape), dev)
    N.check(L.dcv_conv_backward_weight(C.byref(geom), ptr(xd_), C.byref(xm), ptr(dyd_), C.byref(ym), ptr(dw), ptr(ws), need, stream_ptr()), "wgrad")
    assert "head_wgrad_kernel" in last_kernel(), last_kernel()
    assert intact(wbig, wn) and rel(dw, gw) < 2e-6
    first = dw.clone()
    N.check(L.dcv_conv_backward_weight_acc(C.byref(geom), ptr(xd_), C.byref(xm), ptr(dyd_), C.byref(ym), ptr(dw), 1, ptr(ws), need, stream_ptr()), "wgrad acc")
    assert intact(wbig, wn) and rel(dw, 2 * gw) < 2e-6
    # the same call again gives the same bits
    dw2 = torch.empty_like(first)
    N.check(L.dcv_conv_backward_weight(C.byref(geom), ptr(xd_), C.byref(xm), ptr(dyd_), C.byref(ym), ptr(dw2), ptr(ws), need, stream_ptr()), "wgrad")
    assert torch.equal(dw2, first)


def test_head_on_a_channel_slice_and_the_neighbouring_geometry(dev):
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    L = N.lib()
    g_ = torch.Generator().manual_seed(41)
    wide = torch.randn(4, 160, 5, 8, 8, generator=g_).to(dev)
    x = wide[:, 32:160]                                         # 128 channels of a wider buffer: sample stride != C * plane volume
    w = (torch.randn(1, 128, 4, 4, 4, generator=g_) * 0.05).to(dev)
    assert not x.is_contiguous()
    geom = ops.conv_geom(w, (1, 2, 2), (0, 1, 1), False)
    y = torch.empty(4, 1, 2, 4, 4, device=dev)
    xm, ym = dims5(x), dims5(y)
    need = L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(xm), C.byref(ym), 0)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    N.check(L.dcv_conv_forward(C.byref(geom), ptr(x), C.byref(xm), ptr(w), ptr(y), C.byref(ym), 0, 0.0, None, ptr(ws), need, stream_ptr()), "fwd")
    assert "head_fwd_kernel" in last_kernel()
    assert rel(y, F.conv3d(x.double().cpu(), w.double().cpu(), None, (1, 2, 2), (0, 1, 1))) < 2e-6
    # 16 x 16 planes: not the head's geometry
    x16 = torch.randn(2, 64, 4, 16, 16, generator=g_).to(dev)
    w16 = (torch.randn(1, 64, 4, 4, 4, generator=g_) * 0.05).to(dev)
    g16 = ops.conv_geom(w16, (1, 2, 2), (0, 1, 1), False)
    y16 = torch.empty(2, 1, 1, 8, 8, device=dev)
    xm16, ym16 = dims5(x16), dims5(y16)
    need16 = L.dcv_conv_workspace_bytes(C.byref(g16), C.byref(xm16), C.byref(ym16), 0)
    ws16 = torch.empty(need16, dtype=torch.uint8, device=dev)
    N.check(L.dcv_conv_forward(C.byref(g16), ptr(x16), C.byref(xm16), ptr(w16), ptr(y16), C.byref(ym16), 0, 0.0, None, ptr(ws16), need16, stream_ptr()), "fwd")
    assert "head_" not in last_kernel()
    assert rel(y16, F.conv3d(x16.double().cpu(), w16.double().cpu(), None, (1, 2, 2), (0, 1, 1))) < 1e-5
