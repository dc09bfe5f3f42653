"""GPU: the bf16 channels-last path at the BENCH sizes — every distinct convolution shape of surreal-depth1 / isogd-flow at B = 100 and the
isogd-depth ones that differ at B = 70: forward, data gradient, weight gradient against torch.nn.functional on the host with the SAME bf16-valued operands, every
operand inside NaN guard bands, and the kernel variant asserted (dcv_debug_last_kernel): the position-split weight gradients (up to 768 splits, both slab-reduce
forms), the fused thin-source / thin-destination kernels, the XCD-range workgroup map, the 64 x 256 and 128 x 128 tiles with their tail tiles are exactly what
B <= 16 never selects.  Bars as in tests/test_cl16_gpu.py: activations / data gradients 5e-3 (one bf16 rounding of the result: ~2.3e-3), weight gradients (fp32
out, fp32 accumulation over up to 6.5 M positions in position-split slabs) 5e-5."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_cl16_gpu import guarded_cl, margins_intact, r16, rel

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
S3, P3 = (1, 2, 2), (0, 1, 1)
# name, transposed, cin, cout, kernel, stride, padding, input shape, expected kernel family (forward, data gradient)
LAYERS = [
    ("ggen96.0_convT_50_768", True, 50, 768, (4, 4), (1, 1), (0, 0), (1600, 50, 1, 1), ("cl_gather", "cl_gather")),
    ("ggen96.3_convT_768_384", True, 768, 384, (4, 4), (2, 2), (1, 1), (1600, 768, 4, 4), ("cl_patch_convt", "cl_gather")),
    ("ggen96.6_convT_384_192", True, 384, 192, (4, 4), (2, 2), (1, 1), (1600, 384, 8, 8), ("cl_patch_convt", "cl_gather")),
    ("ggen96.9_convT_192_96", True, 192, 96, (4, 4), (2, 2), (1, 1), (1600, 192, 16, 16), ("cl_patch_convt", "cl_gather")),
    ("ggen96.12_convT_96_1", True, 96, 1, (4, 4), (2, 2), (1, 1), (1600, 96, 32, 32), ("cl_col2im", "thin")),
    ("ggen.12_convT_64_2_flow", True, 64, 2, (4, 4), (2, 2), (1, 1), (1600, 64, 32, 32), ("cl_col2im", "thin")),
    ("cgen.in_conv3_1_64", False, 1, 64, (3, 3), (1, 1), (1, 1), (1600, 1, 64, 64), ("cl_widen3x3", "cl_thin3x3")),
    ("cgen.in_conv3_2_64_flow", False, 2, 64, (3, 3), (1, 1), (1, 1), (1600, 2, 64, 64), ("cl_widen3x3", "cl_thin3x3")),
    ("cgen.down0_conv_64_64", False, 64, 64, (4, 4), (2, 2), (1, 1), (1600, 64, 64, 64), ("64 x 128", "cl_patch_convt")),
    ("cgen.down1_conv_64_128", False, 64, 128, (4, 4), (2, 2), (1, 1), (1600, 64, 32, 32), ("128 x 128", "cl_patch_convt")),
    ("cgen.down3_conv_256_256", False, 256, 256, (4, 4), (2, 2), (1, 1), (1600, 256, 8, 8), ("128 x 128", "cl_patch_convt")),
    ("cgen.up0_convT_266_256", True, 266, 256, (4, 4), (2, 2), (1, 1), (1600, 266, 1, 1), ("128 x 128", "128 x 128")),
    ("cgen.up2_convT_512_256", True, 512, 256, (4, 4), (2, 2), (1, 1), (1600, 512, 4, 4), ("cl_patch_convt", "128 x 128")),
    ("cgen.up4_convT_256_64", True, 256, 64, (4, 4), (2, 2), (1, 1), (1600, 256, 16, 16), ("cl_patch_convt", "128 x 128")),
    ("cgen.up5_convT_128_64", True, 128, 64, (4, 4), (2, 2), (1, 1), (1600, 128, 32, 32), ("cl_patch_convt", "128 x 128")),
    ("cgen.out_convT3_128_3", True, 128, 3, (3, 3), (1, 1), (1, 1), (1600, 128, 64, 64), ("cl_thin3x3", "cl_widen3x3")),
    ("idis.c_conv_3_32", False, 3, 32, (4, 4), (2, 2), (1, 1), (100, 3, 64, 64), ("thin", "cl_col2im")),
    ("idis.5_conv_128_256", False, 128, 256, (4, 4), (2, 2), (1, 1), (100, 128, 16, 16), ("128 x 128", "cl_patch_convt")),
    ("idis.9_conv_256_1", False, 256, 1, (4, 4), (2, 2), (1, 1), (100, 256, 8, 8), ("cl_col2im", "thin")),
    ("vdis.g_conv3d_1_32", False, 1, 32, (4, 4, 4), S3, P3, (100, 1, 16, 64, 64), ("thin", "cl_col2im")),
    ("vdis.g_conv3d_2_32_flow", False, 2, 32, (4, 4, 4), S3, P3, (100, 2, 16, 64, 64), ("thin", "cl_gather")),
    ("vdis.c_conv3d_3_32", False, 3, 32, (4, 4, 4), S3, P3, (100, 3, 16, 64, 64), ("thin", "cl_gather")),
    ("vdis.1_conv3d_64_128", False, 64, 128, (4, 4, 4), S3, P3, (100, 64, 13, 32, 32), ("128 x 128", "64 x 128")),
    ("vdis.5_conv3d_128_256", False, 128, 256, (4, 4, 4), S3, P3, (100, 128, 10, 16, 16), ("128 x 128", "128 x 128")),
    ("vdis.9_conv3d_256_1", False, 256, 1, (4, 4, 4), S3, P3, (100, 256, 7, 8, 8), ("cl_col2im", "thin")),
    ("gdis.1_conv3d_1_32", False, 1, 32, (4, 4, 4), S3, P3, (100, 1, 15, 64, 64), ("thin", "cl_col2im")),
    ("gdis.5_conv3d_32_64", False, 32, 64, (4, 4, 4), S3, P3, (100, 32, 12, 32, 32), ("64 x 128", "32 x 256")),
    ("gdis.9_conv3d_64_128", False, 64, 128, (4, 4, 4), S3, P3, (100, 64, 9, 16, 16), ("128 x 128", "64 x 128")),
    ("isogd70_cgen.up5_convT_128_64", True, 128, 64, (4, 4), (2, 2), (1, 1), (1120, 128, 32, 32), ("cl_patch_convt", "128 x 128")),
    ("isogd70_vdis.1_conv3d_64_128", False, 64, 128, (4, 4, 4), S3, P3, (70, 64, 13, 32, 32), ("128 x 128", "64 x 128")),
]


@pytest.mark.parametrize("case", LAYERS, ids=[c[0] for c in LAYERS])
def test_cl16_layer_at_bench_size(case):
    from dcvgan_amd import native, ops, ops_cl
    L = native.lib()
    name, tr, cin, cout, k, s, p, xs, want = case
    nd = len(k)
    g = torch.Generator().manual_seed(abs(hash(name)) % 10007)
    w = r16(torch.randn(((cin, cout) if tr else (cout, cin)) + k, generator=g) * 0.05).requires_grad_(True)
    x = r16(torch.randn(xs, generator=g)).requires_grad_(True)
    fn = {(False, 2): F.conv2d, (False, 3): F.conv3d, (True, 2): F.conv_transpose2d}[(tr, nd)]
    y_ref = fn(x, w, None, s, p)
    cot = r16(torch.randn(y_ref.shape, generator=g))
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot).sum(), [x, w])
    xc, xstore, G = guarded_cl(x.shape, x.detach().to(DEV))
    xc.requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    y = ops_cl.conv(xc, wd, ops.conv_geom(wd, s, p, tr))
    k_fwd = L.dcv_debug_last_kernel().decode()
    cc, cstore, G2 = guarded_cl(cot.shape, cot.to(DEV))
    # data gradient and weight gradient separately, so that each kernel's name can be read
    (gx,) = torch.autograd.grad(y, [xc], cc, retain_graph=True)
    # (autograd runs both; the last launch of the first call is the weight gradient's reduce: ask for the data gradient alone through the op)
    dx = ops_cl.cl_empty(xc.shape, DEV)
    from dcvgan_amd.native import dims5, ptr, stream_ptr
    import ctypes as C
    geo = ops.conv_geom(wd, s, p, tr)
    dxd, dyd = dims5(dx), dims5(cc)
    pk = ops_cl._packed(wd, 1, geo, dxd, dyd, tuple(xc.shape))
    wsp, wsn = ops._ws("clconv", L.dcv_cl_conv_workspace_bytes(C.byref(geo), C.byref(dxd), C.byref(dyd), 1), DEV)
    native.check(L.dcv_cl_conv_backward_data(C.byref(geo), ptr(cc), C.byref(dyd), ptr(pk), ptr(dx), C.byref(dxd), 0, wsp, wsn, stream_ptr()), "dgrad")
    k_dgrad = L.dcv_debug_last_kernel().decode()
    (gw,) = torch.autograd.grad(y, [wd], cc)
    # (the kernel-name diagnostic is per host thread and autograd's backward runs on its own: the weight gradient once more, called from this thread)
    dw = torch.empty(wd.shape, dtype=torch.float32, device=DEV)
    xd5 = dims5(xc)
    need = L.dcv_cl_wgrad_workspace_bytes(C.byref(geo), C.byref(xd5), C.byref(dyd)); wsp, wsn = ops._ws("clconv", need, DEV)
    native.check(L.dcv_cl_conv_backward_weight(C.byref(geo), ptr(xc), C.byref(xd5), ptr(cc), C.byref(dyd), ptr(dw), wsp, wsn, stream_ptr()), "wgrad")
    k_wgrad = L.dcv_debug_last_kernel().decode()
    torch.cuda.synchronize()
    assert torch.equal(dw, gw), name
    assert want[0] in k_fwd, (name, k_fwd)
    assert want[1] in k_dgrad, (name, k_dgrad)
    assert "cl_wgrad_kernel" in k_wgrad, (name, k_wgrad)
    assert torch.equal(dx[:, :cin], gx[:, :cin]), name               # the op called directly = the op autograd called
    assert bool(torch.isfinite(y.float()).all() and torch.isfinite(gx.float()).all() and torch.isfinite(gw).all()), name      # nothing outside the operands was read
    assert margins_intact(xstore, G) and margins_intact(cstore, G2), name                                                       # ... or written
    errs = [rel(y.float(), y_ref), rel(gx.float(), gx_ref), rel(gw, gw_ref)]
    assert errs[0] < 5e-3 and errs[1] < 5e-3 and errs[2] < 5e-5, (name, errs, k_fwd, k_dgrad, k_wgrad)
