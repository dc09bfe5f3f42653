"""The dominant kernel alone, for rocprofv3 passes: cgen.up_blocks.5 forward (ConvTranspose2d 128 -> 64, 4x4 s2 p1 on
(F,128,32,32), F = 16 B) — 3 warm-up + N timed launches, exactly bench.py's probe.  Usage: python3 tools/probe_dominant.py [B] [N]"""
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, ops

native.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(B * 16, 128, 32, 32, device=dev)
w = torch.randn(128, 64, 4, 4, device=dev) * 0.02
g = ops.conv_geom(w, (2, 2), (1, 1), True)
with torch.no_grad():
    for _ in range(3):
        ops.conv(x, w, g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        ops.conv(x, w, g)
    e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / N
fl = 2.0 * B * 16 * 64 * 64 * 64 * 128 * 4
print(f"{native.lib().dcv_debug_last_kernel().decode()}: {ms:.4f} ms, {fl / ms / 1e9:.1f} TFLOP/s, batch {B}")
