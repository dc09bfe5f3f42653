"""Producer / consumer launches on ONE stream whose workgroups map data to different XCDs (tools/guard/lds_guard.hip, mem_guard_round_xcd), run while two other streams
carry a discriminator each (one of them at another MFMA precision, the mix the step's side streams produce).  A word that reads back as the PREVIOUS round's value means the
consumer launch saw a cache line its predecessor on the same stream had already overwritten.
Usage: python tools/xcd_guard_probe.py [precision of vdis] [rounds] [MiB]"""
import ctypes as C
import os
import sys
sys.path.insert(0, '.')
import torch
from dcvgan_amd import native, trainer, util
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.rng import PhiloxRng

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
MIB = int(sys.argv[3]) if len(sys.argv) > 3 else 16
native.lib()
native.set_precision("fp32")
G = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "guard", "lds_guard.so"))
G.mem_guard_round_xcd.argtypes = [C.c_void_p, C.c_longlong, C.c_uint, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
B = 16
cfg = CONFIGS["surreal-depth1"].scaled(batchsize=B)
g = torch.Generator().manual_seed(3)
xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev).requires_grad_(True); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev).requires_grad_(True)
torch.manual_seed(11)
models = trainer.build_models(cfg, dev)
r = PhiloxRng(5)
for m in models.values():
    m._rng = r
    m.train()
if mode != "none":
    util.set_precision(models["vdis"], mode)
bad = torch.zeros(1, dtype=torch.int64, device=dev); first = torch.zeros(4, dtype=torch.int32, device=dev)
nchunks = MIB * 256
buf = torch.zeros(nchunks * 1024, dtype=torch.int32, device=dev)
sa, s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
main = torch.cuda.current_stream()


def work():
    ys = []
    for s, k in ((s1, "vdis"), (s2, "gdis")):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            ys.append(models[k](xg, xc))
    for s, y in zip((s1, s2), ys):
        main.wait_stream(s); y.record_stream(main)
    sum(y.float().sum() for y in ys).backward()
    xg.grad = None; xc.grad = None


if mode != "none":
    work(); torch.cuda.synchronize()
for trial in range(3):
    bad.zero_(); first.zero_(); torch.cuda.synchronize()
    done = 0
    while done < K:
        if mode != "none":
            work()
        for k in range(100):
            rc = G.mem_guard_round_xcd(buf.data_ptr(), nchunks, trial * K + done + k + 1, 1 + (k % 7), bad.data_ptr(), first.data_ptr(), 512, sa.cuda_stream)
            assert rc == 0, rc
        done += 100
    torch.cuda.synchronize()
    f = [int(v) & 0xffffffff for v in first.tolist()]
    print(f"vdis {mode}, gdis fp32 on two streams; guard stream: {K} fill/check rounds over {MIB} MiB with shifted block->chunk maps: {int(bad.item())} words read back wrong"
          + (f"; first: word {f[0]} got {f[1]:#010x} want {f[2]:#010x} (round {f[3]}; the previous round's value there: {(f[0] * 2654435761 ^ ((f[3] - 1) * 0x9e3779b9)) & 0xffffffff:#010x})" if bad.item() else ""))
