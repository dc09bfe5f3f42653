"""GPU: the colour generator's RGB head (`Outconv`, /root/reference/src/generator.py:256-282: ConvTranspose2d(2 ngf, 3, 3, 1, 1) + Tanh on 64 x 64 frames)
on its round-6 kernels — the data gradient as a K = 27 GEMM on the matrix pipe (widen_mfma_kernel) and the weight gradient as one MFMA column of taps
with the 128-channel operand streamed once (thinj_wgrad_kernel) — against torch on the host, through the C ABI, inside NaN guard bands, on the shapes
that exercise every index path: image heights that give 4 / 2 / 1 row groups per workgroup, the first and last rows and columns (halo taps), image
counts that leave XCD slots and the last workgroup ragged, a destination that is a channel slice of a wider buffer (accumulate), 64 and 128 channels,
and 256 channels for the weight gradient's second channel tile.  The B = 70 shape itself is in tests/test_b70_gpu.py (`cgen.out`)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GUARD = 8192


def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def guarded(t):
    big = torch.full((t.numel() + 2 * GUARD,), float("nan"), device=t.device, dtype=t.dtype)
    big[GUARD:GUARD + t.numel()] = t.reshape(-1)
    return torch.as_strided(big, t.shape, t.stride(), GUARD), big


def margins_intact(big, n):
    return bool(torch.isnan(big[:GUARD]).all() and torch.isnan(big[GUARD + n:]).all())


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


# frames, channels into the head, image height (the width is 64: the kernels' row form)
HEAD_CASES = [(5, 128, 64), (9, 128, 32), (3, 128, 24), (2, 128, 20), (17, 64, 16), (1, 128, 4), (2, 256, 8)]


@pytest.mark.parametrize("n,cin,h", HEAD_CASES, ids=lambda v: str(v))
def test_rgb_head_forward_and_gradients(dev, n, cin, h):
    from dcvgan_amd import ops
    g = torch.Generator().manual_seed(100 + n + cin + h)
    x = torch.randn(n, cin, h, 64, generator=g, requires_grad=True)
    w = (torch.randn(cin, 3, 3, 3, generator=g) * 0.05).requires_grad_(True)
    y_ref = torch.tanh(F.conv_transpose2d(x, w, None, 1, 1))
    cot = torch.randn(y_ref.shape, generator=g)
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot).sum(), [x, w])
    xd, xbig = guarded(x.detach().to(dev)); xd.requires_grad_(True)
    wd, wbig = guarded(w.detach().to(dev)); wd.requires_grad_(True)
    y = ops.conv(xd, wd, ops.conv_geom(wd, (1, 1), (1, 1), True), ops.ACT_TANH)
    cotd, cbig = guarded(cot.to(dev))
    # (which kernels ran is asserted in the direct C-ABI tests below: the note dcv_debug_last_kernel reads is per thread, and backward runs on autograd's)
    gx, gw = torch.autograd.grad((y * cotd).sum(), [xd, wd])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(gx).all() and torch.isfinite(gw).all())
    assert margins_intact(xbig, x.numel()) and margins_intact(wbig, w.numel()) and margins_intact(cbig, cot.numel())
    assert rel(y, y_ref) < 1e-5
    assert rel(gx, gx_ref) < 1e-5
    assert rel(gw, gw_ref) < 2e-5
    # the first / last rows and columns separately (halo taps): a wrong zero there moves the norm by little
    for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), h - 1), (slice(None), slice(None), slice(None), 0), (slice(None), slice(None), slice(None), 63)):
        assert rel(gx[sl], gx_ref[sl]) < 1e-5


@pytest.mark.parametrize("n,cin,h", [(3, 128, 32), (5, 64, 8)], ids=lambda v: str(v))
def test_rgb_head_data_gradient_accumulates_into_a_slice(dev, n, cin, h):
    """dcv_conv_backward_data(accumulate = 1) into the last `cin` channels of a wider buffer that already holds a gradient (how the U-Net's skip
    connections receive theirs, ops.GradSlot): the neighbouring channels stay untouched, the slice gets old + new."""
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    g = torch.Generator().manual_seed(7 + n + cin)
    w = torch.randn(cin, 3, 3, 3, generator=g) * 0.1
    x = torch.randn(n, cin, h, 64, generator=g, requires_grad=True)
    y = F.conv_transpose2d(x, w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad((y * dy).sum(), [x])
    wide = torch.randn(n, cin + 5, h, 64, generator=g)
    want = wide.clone(); want[:, 5:] += gx
    wide_d, dy_d, w_d = wide.to(dev), dy.to(dev), w.to(dev)
    dx = wide_d[:, 5:]
    geom = ops.conv_geom(w_d, (1, 1), (1, 1), True)
    dxd, dyd = dims5(dx), dims5(dy_d)
    L = N.lib()
    need = L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(dxd), C.byref(dyd), 1)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    N.check(L.dcv_conv_backward_data(C.byref(geom), ptr(dy_d), C.byref(dyd), ptr(w_d), ptr(dx), C.byref(dxd), 1, None, ptr(ws), need, stream_ptr()), "accumulate")
    assert "widen_mfma_kernel" in last_kernel(), last_kernel()
    assert rel(wide_d, want) < 1e-5
    assert torch.equal(wide_d[:, :5].cpu(), wide[:, :5])


def test_rgb_head_weight_gradient_accumulates(dev):
    """dcv_conv_backward_weight_acc: the slab reduce adds into what dw holds (second use of a weight in one backward)."""
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    g = torch.Generator().manual_seed(21)
    n, cin, h = 6, 128, 16
    w = (torch.randn(cin, 3, 3, 3, generator=g) * 0.1).requires_grad_(True)
    x = torch.randn(n, cin, h, 64, generator=g)
    dy = torch.randn(n, 3, h, 64, generator=g)
    (gw,) = torch.autograd.grad((F.conv_transpose2d(x, w, None, 1, 1) * dy).sum(), [w])
    old = torch.randn(w.shape, generator=g)
    xd, dyd_, dwd = x.to(dev), dy.to(dev), old.to(dev)
    geom = ops.conv_geom(w.detach().to(dev), (1, 1), (1, 1), True)
    xdm, dym = dims5(xd), dims5(dyd_)
    L = N.lib()
    need = L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(xdm), C.byref(dym), 2)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    N.check(L.dcv_conv_backward_weight_acc(C.byref(geom), ptr(xd), C.byref(xdm), ptr(dyd_), C.byref(dym), ptr(dwd), 1, ptr(ws), need, stream_ptr()), "wgrad acc")
    assert "thinj_wgrad_kernel" in last_kernel(), last_kernel()
    assert rel(dwd, old + gw) < 2e-5


@pytest.mark.parametrize("n,h,cbn,act", [(5, 64, 64, "relu"), (3, 32, 64, "relu"), (2, 16, 32, "leaky"), (9, 8, 96, "none")], ids=lambda v: str(v))
def test_head_data_gradient_fused_with_the_batchnorm_backward(dev, n, h, cbn, act):
    """dcv_conv_backward_data_bn against the two calls it replaces (dcv_conv_backward_data, then dcv_bn_act_backward on the first cbn channels of its result):
    the head's other channels bit for bit (same kernel body), the BatchNorm input's gradient and dgamma / dbeta to fp32 summation-order accuracy; the first cbn
    channels of dx are left unwritten (NaN-filled here: nothing may depend on them)."""
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    g_ = torch.Generator().manual_seed(31 + n + h + cbn)
    C_ = 128
    w = (torch.randn(C_, 3, 3, 3, generator=g_) * 0.1).to(dev)
    dy = torch.randn(n, 3, h, 64, generator=g_).to(dev)
    bx = (torch.randn(n, cbn, h, 64, generator=g_) * 1.3 + 0.2).to(dev)
    gamma = (torch.rand(cbn, generator=g_) + 0.5).to(dev); beta = (torch.randn(cbn, generator=g_) * 0.2).to(dev)
    code, slope = {"relu": (ops.ACT_LEAKY, 0.0), "leaky": (ops.ACT_LEAKY, 0.2), "none": (ops.ACT_NONE, 0.0)}[act]
    mean = bx.mean((0, 2, 3)); invstd = 1.0 / torch.sqrt(bx.var((0, 2, 3), unbiased=False) + 1e-5)
    geom = ops.conv_geom(w, (1, 1), (1, 1), True)
    L = N.lib()
    # the two separate calls
    dx_ref = torch.empty(n, C_, h, 64, device=dev)
    dxd, dyd = dims5(dx_ref), dims5(dy)
    need = L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(dxd), C.byref(dyd), 1)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    N.check(L.dcv_conv_backward_data(C.byref(geom), ptr(dy), C.byref(dyd), ptr(w), ptr(dx_ref), C.byref(dxd), 0, None, ptr(ws), need, stream_ptr()), "plain")
    first = dx_ref[:, :cbn]
    bdx_ref = torch.empty_like(bx); dgb_ref = torch.empty(2, cbn, device=dev)
    nb = L.dcv_bn_workspace_bytes(cbn)
    wsb = torch.empty(nb, dtype=torch.uint8, device=dev)
    fd, bxd, bdd = dims5(first), dims5(bx), dims5(bdx_ref)
    N.check(L.dcv_bn_act_backward(ptr(first), C.byref(fd), ptr(bx), C.byref(bxd), ptr(bdx_ref), C.byref(bdd), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), None,
                                  1, code, slope, ptr(dgb_ref[0]), ptr(dgb_ref[1]), ptr(wsb), nb, stream_ptr()), "bn backward")
    # the fused call
    dx = torch.full((n, C_, h, 64), float("nan"), device=dev)
    bdx = torch.full_like(bx, float("nan")); dgb = torch.full((2, cbn), float("nan"), device=dev)
    need2 = L.dcv_conv_backward_data_bn_workspace_bytes(C.byref(dxd), cbn)
    ws2 = torch.empty(need2, dtype=torch.uint8, device=dev)
    fused = C.c_int(0)
    N.check(L.dcv_conv_backward_data_bn(C.byref(geom), ptr(dy), C.byref(dyd), ptr(w), ptr(dx), C.byref(dxd), None, ptr(ws), need, cbn, ptr(bx), C.byref(bxd),
                                        ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), code, slope, ptr(bdx), C.byref(bdd), ptr(dgb[0]), ptr(dgb[1]),
                                        ptr(ws2), need2, C.byref(fused), stream_ptr()), "fused")
    torch.cuda.synchronize()
    assert fused.value == 1 and "head_bn_kernel<2>" in last_kernel(), last_kernel()
    assert bool(torch.isnan(dx[:, :cbn]).all())                                    # never written
    assert torch.equal(dx[:, cbn:], dx_ref[:, cbn:])                                # the same MFMA chain, the same stores
    assert rel(dgb, dgb_ref) < 2e-6 and rel(bdx, bdx_ref) < 2e-6, (rel(dgb, dgb_ref), rel(bdx, bdx_ref))


@pytest.mark.parametrize("n,h,cbn,act", [(5, 64, 64, "relu"), (3, 32, 64, "leaky"), (2, 16, 32, "none")], ids=lambda v: str(v))
def test_head_forward_and_weight_gradient_normalise_on_load(dev, n, h, cbn, act):
    """dcv_conv_forward_bn / dcv_conv_backward_weight_bn: the operand's first cbn channels are NaN (never written); the kernels read the BatchNorm input instead and
    apply act(x * sc + sh) on load — against the plain entries on the materialised operand (dcv_bn_apply), to fp32 summation-order accuracy; padding rows stay zero
    (first / last output rows and columns compared separately)."""
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    g_ = torch.Generator().manual_seed(77 + n + h + cbn)
    C_ = 128
    w = (torch.randn(C_, 3, 3, 3, generator=g_) * 0.1).to(dev)
    bx = (torch.randn(n, cbn, h, 64, generator=g_) * 1.3 + 0.2).to(dev)
    gamma = (torch.rand(cbn, generator=g_) + 0.5).to(dev); beta = (torch.randn(cbn, generator=g_) * 0.2).to(dev)
    code, slope = {"relu": (ops.ACT_LEAKY, 0.0), "leaky": (ops.ACT_LEAKY, 0.2), "none": (ops.ACT_NONE, 0.0)}[act]
    mean = bx.mean((0, 2, 3)); invstd = 1.0 / torch.sqrt(bx.var((0, 2, 3), unbiased=False) + 1e-5)
    rest = torch.randn(n, C_ - cbn, h, 64, generator=g_).to(dev)
    dy = torch.randn(n, 3, h, 64, generator=g_).to(dev)
    geom = ops.conv_geom(w, (1, 1), (1, 1), True)
    L = N.lib()
    full = torch.empty(n, C_, h, 64, device=dev); full[:, cbn:] = rest
    first = full[:, :cbn]
    bxd, fd = dims5(bx), dims5(first)
    N.check(L.dcv_bn_apply(ptr(bx), C.byref(bxd), ptr(first), C.byref(fd), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), None, code, slope, stream_ptr()), "bn_apply")
    z = (bx * (gamma * invstd)[None, :, None, None] + (beta - mean * gamma * invstd)[None, :, None, None])
    want_first = torch.where(z > 0, z, z * slope) if code == ops.ACT_LEAKY else z
    assert rel(first, want_first) < 1e-6
    holey = torch.full((n, C_, h, 64), float("nan"), device=dev); holey[:, cbn:] = rest
    y_ref = torch.empty(n, 3, h, 64, device=dev); y = torch.empty_like(y_ref)
    xd, yd = dims5(full), dims5(y)
    need = max(L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(xd), C.byref(yd), 0), L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(xd), C.byref(yd), 2))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    N.check(L.dcv_conv_forward(C.byref(geom), ptr(full), C.byref(xd), ptr(w), ptr(y_ref), C.byref(yd), ops.ACT_TANH, 0.0, None, ptr(ws), need, stream_ptr()), "fwd")
    tail = (cbn, ptr(bx), C.byref(bxd), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), code, slope)
    N.check(L.dcv_conv_forward_bn(C.byref(geom), ptr(holey), C.byref(xd), ptr(w), ptr(y), C.byref(yd), ops.ACT_TANH, 0.0, None, ptr(ws), need, *tail, stream_ptr()), "fwd bn")
    assert "on load" in last_kernel(), last_kernel()
    assert bool(torch.isfinite(y).all()) and rel(y, y_ref) < 1e-6      # (small batches take the split-K gather on the plain entry: another summation order)
    for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), h - 1), (slice(None), slice(None), slice(None), 0), (slice(None), slice(None), slice(None), 63)):
        assert rel(y[sl], y_ref[sl]) < 1e-6                              # the padding rows / columns stayed zero under the transform
    dw_ref = torch.empty_like(w); dw = torch.empty_like(w)
    N.check(L.dcv_conv_backward_weight(C.byref(geom), ptr(full), C.byref(xd), ptr(dy), C.byref(yd), ptr(dw_ref), ptr(ws), need, stream_ptr()), "wgrad")
    N.check(L.dcv_conv_backward_weight_bn(C.byref(geom), ptr(holey), C.byref(xd), ptr(dy), C.byref(yd), ptr(dw), 0, ptr(ws), need, *tail, stream_ptr()), "wgrad bn")
    assert "on load" in last_kernel(), last_kernel()
    assert bool(torch.isfinite(dw).all()) and rel(dw, dw_ref) < 1e-6
    # a geometry the head's kernels do not take is refused before anything runs
    w4 = torch.randn(C_, 4, 3, 3, device=dev); g4 = ops.conv_geom(w4, (1, 1), (1, 1), True)
    y4 = torch.empty(n, 4, h, 64, device=dev); y4d = dims5(y4)
    assert L.dcv_conv_forward_bn(C.byref(g4), ptr(holey), C.byref(xd), ptr(w4), ptr(y4), C.byref(y4d), 0, 0.0, None, ptr(ws), need, *tail, stream_ptr()) == N.DCV_EUNSUPPORTED


def test_round6_entries_refuse_bad_arguments(dev):
    """Error behaviour of the round-6 entry points, checked before anything is launched (the launch counter does not move): null operands -> DCV_EINVAL, a BatchNorm
    activation the fused backward cannot differentiate -> DCV_EUNSUPPORTED, a second workspace below dcv_conv_backward_data_bn_workspace_bytes -> DCV_EWORKSPACE."""
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    L = N.lib()
    n, h, cbn, C_ = 2, 8, 64, 128
    w = torch.randn(C_, 3, 3, 3, device=dev) * 0.1
    geom = ops.conv_geom(w, (1, 1), (1, 1), True)
    dy = torch.randn(n, 3, h, 64, device=dev); dx = torch.empty(n, C_, h, 64, device=dev)
    bx = torch.randn(n, cbn, h, 64, device=dev); bdx = torch.empty_like(bx)
    v = torch.ones(cbn, device=dev); dg = torch.empty(cbn, device=dev); db = torch.empty(cbn, device=dev)
    dyd, dxd, bxd = dims5(dy), dims5(dx), dims5(bx)
    need = L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(dxd), C.byref(dyd), 1)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    L.dcv_conv_backward_data_bn_workspace_bytes.restype = C.c_size_t
    need2 = L.dcv_conv_backward_data_bn_workspace_bytes(C.byref(dxd), cbn)
    assert need2 > 0
    ws2 = torch.empty(need2, dtype=torch.uint8, device=dev)
    fused = C.c_int(-1)

    def bwd(dy_p, act, ws2_bytes):
        return L.dcv_conv_backward_data_bn(C.byref(geom), dy_p, C.byref(dyd), ptr(w), ptr(dx), C.byref(dxd), None, ptr(ws), ws.numel(), cbn, ptr(bx), C.byref(bxd),
                                           ptr(v), ptr(v), ptr(v), ptr(v), act, 0.0, ptr(bdx), C.byref(bxd), ptr(dg), ptr(db), ptr(ws2), ws2_bytes, C.byref(fused), stream_ptr())
    before = L.dcv_launch_count()
    assert bwd(None, ops.ACT_LEAKY, need2) == N.DCV_EINVAL
    assert bwd(ptr(dy), ops.ACT_TANH, need2) == N.DCV_EUNSUPPORTED
    assert bwd(ptr(dy), ops.ACT_LEAKY, need2 - 1) == N.DCV_EWORKSPACE
    s1 = torch.ones((), device=dev)
    assert L.dcv_scale_dev(None, 4, ptr(s1), ptr(dg), stream_ptr()) == N.DCV_EINVAL
    assert L.dcv_scale_dev(ptr(dg), -1, ptr(s1), ptr(dg), stream_ptr()) == N.DCV_EINVAL
    assert L.dcv_bn_apply(None, C.byref(bxd), ptr(bdx), C.byref(bxd), ptr(v), ptr(v), ptr(v), ptr(v), None, ops.ACT_NONE, 0.0, stream_ptr()) == N.DCV_EINVAL
    assert L.dcv_bn_forward_stats_only(ptr(bx), C.byref(bxd), None, None, None, ptr(v), ptr(v), 0.1, 1e-5, None, 1, cbn, stream_ptr()) == N.DCV_EINVAL
    assert L.dcv_launch_count() == before
    assert b"" != L.dcv_last_error()
    # ... and the same call with good arguments runs and reports the fused path
    N.check(bwd(ptr(dy), ops.ACT_LEAKY, need2), "dcv_conv_backward_data_bn")
    assert fused.value == 1 and bool(torch.isfinite(bdx).all()) and bool(torch.isfinite(dx[:, cbn:]).all())
