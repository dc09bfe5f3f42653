"""Does any kernel of a module write LDS it does not own?  Guard workgroups (tools/guard/lds_guard.hip: 32 KB of pattern each, re-checked every ~30 us)
sit on the card on one stream while a discriminator's forward + backward runs beside them on another; words that changed under a guard are counted.
Usage: python tools/lds_guard_probe.py [config] [B] [precision] [module] [trials]"""
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
T = int(sys.argv[5]) if len(sys.argv) > 5 else 10
SHORT = int(sys.argv[6]) if len(sys.argv) > 6 else 0      # > 0: that many launches of short-lived guards instead of one long-lived one
native.lib()
native.set_precision(mode)
G = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "guard", "lds_guard.so"))
G.lds_guard_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
G.xwave_guard_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_void_p]
XWAVE = "xwave" in sys.argv
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
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
xgr = xg.clone().requires_grad_(True); xcr = xc.clone().requires_grad_(True)


def work():
    y = d(xgr[:, :, 2], xcr[:, :, 2]) if which == "idis" else d(xgr, xcr)
    y.float().sum().backward()


work(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); work(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
rounds = max(10, int(ms * T * 1000 / 35))
for trial in range(3):
    bad.zero_(); first.zero_(); torch.cuda.synchronize()
    if SHORT:       # short-lived guards: each launch takes LDS that other workgroups have just left (a write that lands after its workgroup ended would show here)
        with torch.cuda.stream(sb):
            for _ in range(T):
                work()
        for i_ in range(SHORT):
            rc = G.xwave_guard_launch(bad.data_ptr(), first.data_ptr(), 2048, i_ * 2048 + 1, sa.cuda_stream) if XWAVE else G.lds_guard_launch(bad.data_ptr(), first.data_ptr(), 1024, 2, 2, sa.cuda_stream)
            assert rc == 0, rc
    else:
        rc = G.lds_guard_launch(bad.data_ptr(), first.data_ptr(), 512, rounds, 10, sa.cuda_stream)
        assert rc == 0, rc
        with torch.cuda.stream(sb):
            for _ in range(T):
                work()
    torch.cuda.synchronize()
    what = ("%d launches of 1024 short-lived guards" % SHORT) if SHORT else ("512 guard workgroups x %d checks" % rounds)
    print(f"{mode} {which}: trial {trial}: {T} forward+backward passes ({ms:.2f} ms each) beside {what}: "
          f"{int(bad.item())} LDS words changed under a guard" + (f"; first: block {first[0].item()}, word {first[1].item()}, got {first[2].item() & 0xffffffff:#010x}, pattern {first[3].item() & 0xffffffff:#010x}" if bad.item() else ""))
