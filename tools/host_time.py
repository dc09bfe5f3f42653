"""Host enqueue time of one G+D iteration against its GPU time (is the host ever the bottleneck?).
usage: python3 tools/host_time.py [config [precision [profile]]]   -> per iteration: host ms (step() returns), wall ms (after synchronize);
precision bf16cl = the 16-bit channels-last path; a third argument adds a cProfile of three iterations (top functions by own time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcvgan_amd import trainer
from dcvgan_amd.configs import CONFIGS

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "isogd-depth"]
dev = torch.device("cuda:0")
if len(sys.argv) > 2 and sys.argv[2] == "bf16cl":
    from dcvgan_amd import ops_cl
    ops_cl.enable(True)
torch.manual_seed(0)
models = trainer.build_models(cfg, dev)
opts = trainer.build_optimizers(cfg, models)
run = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg))
B = cfg.batchsize
xc = torch.rand(B, 3, cfg.video_length, 64, 64, device=dev) * 2 - 1
xg = torch.rand(B, cfg.channel, cfg.video_length, 64, 64, device=dev) * 2 - 1
for i in range(3):
    run.step(xc, xg, i)
torch.cuda.synchronize()
for i in range(5):
    t0 = time.perf_counter()
    run.step(xc, xg, i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"iteration {i}: host {1e3 * (t1 - t0):.1f} ms, wall {1e3 * (t2 - t0):.1f} ms")

if len(sys.argv) > 3:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(3):
        run.step(xc, xg, i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(40)
