"""Stand-alone attempt at DESIGN §8(d): tools/guard/pk_vs_mfma.hip's packed-FP32 kernel checks itself (integer-valued data: any mismatch is a wrong instruction result) while
a second stream keeps the matrix pipes busy with bf16 MFMAs, fp32 MFMAs, or nothing.  Build first:  hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/guard/pk_vs_mfma.hip -o tools/guard/pk_vs_mfma.so
Usage: python tools/pk_vs_mfma_probe.py [rounds]"""
import ctypes as C
import os
import sys
import torch

R = int(sys.argv[1]) if len(sys.argv) > 1 else 30
G = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "guard", "pk_vs_mfma.so"))
G.pk_check_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
G.mfma_spin_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randint(-4, 5, (64, 64), generator=g).float().to(dev)
w = torch.randint(-3, 4, (64, 16), generator=g).float().to(dev)
bad = torch.zeros(1, dtype=torch.int64, device=dev); first = torch.zeros(4, dtype=torch.int32, device=dev); out = torch.zeros(4, device=dev)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
sys.path.insert(0, '.')
from dcvgan_amd import native as N, ops      # the library's own convolutions as the other stream's load: the kernels beside which the defect showed
N.lib()
xc_ = torch.randn(16, 128, 10, 16, 16, device=dev).requires_grad_(True); wc_ = (torch.randn(256, 128, 4, 4, 4, device=dev) * 0.05).requires_grad_(True)


def lib_conv(mode):
    N.set_precision(mode)
    with torch.cuda.stream(sb):
        for _ in range(3):
            y = ops.conv(xc_, wc_, ops.conv_geom(wc_, (1, 2, 2), (0, 1, 1), False))
            torch.autograd.grad(y, [xc_], torch.ones_like(y))
    N.set_precision("fp32")


for what, code in (("nothing else on the card", -1), ("fp32 MFMA waves (a spin kernel) on another stream", 0), ("bf16 MFMA waves (a spin kernel) on another stream", 1),
                   ("the library's conv3d 128 -> 256 forward + data gradient, fp32 MFMA, on another stream", "fp32"),
                   ("the library's conv3d 128 -> 256 forward + data gradient, bf16 products (LDS-DMA patch kernels), on another stream", "bf16"),
                   ("the same at f32x6", "f32x6")):
    bad.zero_(); torch.cuda.synchronize()
    for _ in range(R):
        if isinstance(code, str):
            lib_conv(code)
        elif code >= 0:
            assert G.mfma_spin_launch(code, out.data_ptr(), 4000, 1024, sb.cuda_stream) == 0
        for _ in range(20):
            assert G.pk_check_launch(x.data_ptr(), w.data_ptr(), 64, 8, bad.data_ptr(), first.data_ptr(), 1024, sa.cuda_stream) == 0
    torch.cuda.synchronize()
    f = [int(v) & 0xffffffff for v in first.tolist()]
    print(f"packed-FP32 self-check beside {what}: {int(bad.item())} mismatching (thread, pass) pairs in {R * 20} launches of 262 144 threads x 8 passes"
          + (f"; first at block {f[0]} thread {f[1]}: packed {f[2]:#010x} single {f[3]:#010x}" if bad.item() else ""))
