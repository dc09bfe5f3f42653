"""GPU: the 3-D discriminators' stems — Conv3d(cin -> 32, 4x4x4, stride (1,2,2), padding (0,1,1)) on 64-wide frames (/root/reference/src/discriminator.py:181-206,
288-305: the depth / flow and colour branches of VideoDiscriminator, GradientDiscriminator's first layer) — weight gradient on stem3d_wgrad_kernel, through the C ABI,
against torch's fp64 convolution on the host, inside NaN guard bands.  Shapes exercise every index path: 1 / 2 / 3 input channels (2 / 4 / 6 column tiles), frame
heights that give 1 ... 32 output rows (the first and last rows read the zero rows above / below the frame), depths with 1 ... 10 output planes, sample counts
that leave the last wave's run of rows ragged and put sample / plane boundaries inside a run; strided operands as the modules hand them over (x a (B,C,T,H,W) view of
(B,T,C,H,W) memory, dy one 32-channel half of a 64-channel concat buffer); accumulation into an existing gradient; and a width the kernel does not take (fallback)."""
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


def reference(x, dy, cout):
    """dw of conv3d(x, w, stride (1,2,2), padding (0,1,1)) for cotangent dy, in fp64 on the host."""
    w = torch.zeros(cout, x.shape[1], 4, 4, 4, dtype=torch.float64, requires_grad=True)
    y = F.conv3d(x.double().cpu(), w, None, (1, 2, 2), (0, 1, 1))
    (gw,) = torch.autograd.grad((y * dy.double().cpu()).sum(), [w])
    return gw


def run(dev, x, dy, cout=32, accumulate=None, expect="stem3d_wgrad_kernel"):
    from dcvgan_amd import native as N, ops
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    L = N.lib()
    w = torch.empty(cout, x.shape[1], 4, 4, 4, device=dev)
    geom = ops.conv_geom(w, (1, 2, 2), (0, 1, 1), False)
    xd, dyd = dims5(x), dims5(dy)
    need = L.dcv_conv_workspace_bytes(C.byref(geom), C.byref(xd), C.byref(dyd), 2)
    assert need > 0, L.dcv_last_error()
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    big = torch.full((w.numel() + 2 * GUARD,), float("nan"), device=dev)
    dw = big[GUARD:GUARD + w.numel()].view(w.shape)
    if accumulate is not None:
        dw.copy_(accumulate)
    N.check(L.dcv_conv_backward_weight_acc(C.byref(geom), ptr(x), C.byref(xd), ptr(dy), C.byref(dyd), ptr(dw), int(accumulate is not None), ptr(ws), need, stream_ptr()), "wgrad")
    if expect is not None:
        assert expect in last_kernel(), last_kernel()
    torch.cuda.synchronize()
    assert bool(torch.isnan(big[:GUARD]).all() and torch.isnan(big[GUARD + w.numel():]).all())
    return dw.clone()


# samples, input channels, input depth, frame height
CASES = [(3, 1, 5, 64), (2, 3, 4, 64), (5, 2, 7, 8), (1, 3, 4, 2), (7, 1, 13, 16), (4, 3, 16, 64), (9, 2, 6, 32), (33, 1, 4, 4)]


@pytest.mark.parametrize("n,cin,d,h", CASES, ids=lambda v: str(v))
def test_stem_weight_gradient(dev, n, cin, d, h):
    g = torch.Generator().manual_seed(5 + n + 10 * cin + d + h)
    x = torch.randn(n, cin, d, h, 64, generator=g).to(dev)
    dy = torch.randn(n, 32, d - 3, h // 2, 32, generator=g).to(dev)
    dw = run(dev, x, dy)
    assert bool(torch.isfinite(dw).all())
    assert rel(dw, reference(x, dy, 32)) < 2e-6


def test_stem_weight_gradient_on_the_modules_strided_operands_and_accumulate(dev):
    """x as VideoDiscriminator receives a generated clip — a (B,C,T,H,W) view of (B,T,C,H,W) memory (generator.py:90-97) — dy as the stem's half of the 64-channel
    concat buffer's gradient, and the sum into an existing gradient (the second use of the weight in one backward, trainer.py:299-309)."""
    g = torch.Generator().manual_seed(77)
    n, cin, d, h = 4, 3, 8, 64
    x = torch.randn(n, d, cin, h, 64, generator=g).to(dev).permute(0, 2, 1, 3, 4)
    both = torch.randn(n, 64, d - 3, h // 2, 32, generator=g).to(dev)
    dy = both[:, 32:]
    old = torch.randn(32, cin, 4, 4, 4, generator=g).to(dev)
    assert not x.is_contiguous() and not dy.is_contiguous()
    want = reference(x, dy, 32)
    assert rel(run(dev, x, dy), want) < 2e-6
    assert rel(run(dev, x, dy, accumulate=old), old.double().cpu() + want) < 2e-6


def test_runs_are_bitwise_repeatable_and_other_widths_fall_back(dev):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(6, 2, 9, 32, 64, generator=g).to(dev)
    dy = torch.randn(6, 32, 6, 16, 32, generator=g).to(dev)
    a, b = run(dev, x, dy), run(dev, x, dy)
    assert torch.equal(a, b)
    dy16 = torch.randn(6, 16, 6, 16, 32, generator=g).to(dev)          # 16 output channels (a width_div = 2 model): the generic kernel
    dw = run(dev, x, dy16, cout=16, expect=None)
    assert "stem3d" not in last_kernel()
    assert rel(dw, reference(x, dy16, 16)) < 2e-6
