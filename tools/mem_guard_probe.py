"""Does a kernel always see what its predecessor on the same stream wrote, while another stream runs a discriminator's forward + backward?
tools/guard/lds_guard.hip: fill (pattern k) then check (pattern k) in consecutive launches on stream A, K rounds over a buffer of N words that is reused every round.
Usage: python tools/mem_guard_probe.py [config] [B] [precision] [module] [rounds] [words]"""
import ctypes as C
import os
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, trainer
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.rng import PhiloxRng

name = sys.argv[1] if len(sys.argv) > 1 else "surreal-depth1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
which = sys.argv[4] if len(sys.argv) > 4 else "gdis"
K = int(sys.argv[5]) if len(sys.argv) > 5 else 2000
NW = int(sys.argv[6]) if len(sys.argv) > 6 else (1 << 20)
native.lib()
native.set_precision(mode)
G = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "guard", "lds_guard.so"))
G.mem_guard_round.argtypes = [C.c_void_p, C.c_longlong, C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
G.mem_guard_check.argtypes = [C.c_void_p, C.c_longlong, C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
STATIC = "static" in sys.argv       # fill once, then only read: a word that reads back wrong was never rewritten, so it is the load that went wrong
PASSES = 20
for a_ in sys.argv:
    if a_.startswith("passes="):
        PASSES = int(a_.split("=")[1])
dev = torch.device("cuda:0")
cfg = CONFIGS[name].scaled(batchsize=B)
g = torch.Generator().manual_seed(3)
xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)
torch.manual_seed(11)
models = trainer.build_models(cfg, dev)
r = PhiloxRng(5)
for m in models.values():
    m._rng = r
    m.train()
d = models[which]
bad = torch.zeros(1, dtype=torch.int64, device=dev); first = torch.zeros(4, dtype=torch.int32, device=dev)
buf = torch.zeros(NW, dtype=torch.int32, device=dev)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
xgr = xg.clone().requires_grad_(True); xcr = xc.clone().requires_grad_(True)


def work():
    y = d(xgr[:, :, 2], xcr[:, :, 2]) if which == "idis" else d(xgr, xcr)
    y.float().sum().backward()


work(); torch.cuda.synchronize()
for trial in range(3):
    bad.zero_(); first.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sb):
        e0.record()
        for _ in range(PASSES):
            work()
        e1.record()
    if STATIC:
        with torch.cuda.stream(sa):
            pass
        rc = G.mem_guard_round(buf.data_ptr(), NW, 7, bad.data_ptr(), first.data_ptr(), 256, 256, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    for k in range(K):
        if STATIC:
            rc = G.mem_guard_check(buf.data_ptr(), NW, 7, bad.data_ptr(), first.data_ptr(), 1024, sa.cuda_stream)
        else:
            rc = G.mem_guard_round(buf.data_ptr(), NW, k + 1 + trial * K, bad.data_ptr(), first.data_ptr(), 256, 256, sa.cuda_stream)
        assert rc == 0, rc
    torch.cuda.synchronize()
    f = [int(v) & 0xffffffff for v in first.tolist()]
    print(f"{mode} {which}: trial {trial}: {K} {'check-only' if STATIC else 'fill/check'} rounds over {NW * 4 >> 20} MiB beside {PASSES} forward+backward passes ({e0.elapsed_time(e1):.0f} ms): {int(bad.item())} words read back wrong"
          + (f"; first: word {f[0]} got {f[1]:#010x} want {f[2]:#010x} (round {f[3]}; the previous round's value would be {(f[0] * 2654435761 ^ ((f[3] - 1) * 0x9e3779b9)) & 0xffffffff:#010x})" if bad.item() else ""))
