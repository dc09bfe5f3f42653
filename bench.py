#!/usr/bin/env python3
"""videos/s of the DCVGAN G+D training step (config/isogd-depth.yml) on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = the reference trainer's iteration (trainer.py:279-363): D phase (3 D's on real and
fake batches, backward, 3 Adam steps) + G phase (fresh fakes, backward, ggen/cgen/ggen Adam
steps), fp32, synthetic U(-1,1) clips of shape (B,3,16,64,64)+(B,1,16,64,64) resident in HBM,
per-GPU batch 70.  N > 1 is weak-scaling data parallel: every rank runs the step on its own
batch and RNG stream; gradients are all-reduced (RCCL) inside the optimiser wrapper.

Rank 0 prints ONE JSON line.  `roofline` = conv/convT/GRU FLOPs of the as-written step
 x videos/s against the dense fp32 MFMA peak, plus the dominant kernel
timed alone with HIP events; `cpu_baseline` = the CPU oracle's step on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense f32 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="isogd-depth")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing the DP path on one GPU)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-minimal", action="store_true", help="skip the secondary run with the dead D-phase generator backward elided")
    return ap.parse_args()


def dominant_kernel_probe(models, cfg, dev):
    """Time the step's largest single launch family alone: cgen.up_blocks[5] forward
    (ConvTranspose2d 128->64, 4x4 s2 p1 on (F,128,32,32); 4.29 GFLOP/video)
    = 4 stride-parity launches of gather_gemm_kernel<2,2,1,4> + 4 weight packs."""
    from dcvgan_amd import layers, ops
    conv = models["cgen"].up_blocks[5].main[0]
    F_ = cfg.batchsize * cfg.video_length
    x = torch.randn(F_, conv.in_channels, 32, 32, device=dev)
    g = layers.geom_of(conv)
    with torch.no_grad():
        for _ in range(2):
            ops.conv(x, conv.weight, g)
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record(s)
        for _ in range(reps):
            ops.conv(x, conv.weight, g)
        e1.record(s)
        e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * F_ * 64 * 64 * conv.out_channels * conv.in_channels * 4  # 2x2 taps per output
    return {"name": "gather_gemm_kernel<2,2,1,4> x4 (cgen.up_blocks.5 fwd)", "ms": ms, "gflop": flops / 1e9,
            "achieved": flops / ms / 1e9, "unit": "TFLOP/s", "frac": flops / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS}


def host_threads():
    """CPUs this process may really use: affinity mask, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, batch, steps):
    from dcvgan_amd import trainer
    torch.set_num_threads(host_threads())
    from oracle import dcvgan_oracle as O
    c = cfg.scaled(batchsize=batch)
    torch.manual_seed(c.seed)
    models = trainer.build_models(c, torch.device("cpu"))
    states = {n: {k: v.detach().clone() for k, v in m.state_dict().items()} for n, m in models.items()}
    g = torch.Generator().manual_seed(c.seed)
    xc = torch.rand(batch, 3, 16, 64, 64, generator=g) * 2 - 1
    xg = torch.rand(batch, c.channel, 16, 64, 64, generator=g) * 2 - 1
    so = O.StepOracle(c, states, O.TorchRng(record=False))
    so.step(xc, xg, 3)  # warm-up
    t0 = time.perf_counter()
    for i in range(steps):
        so.step(xc, xg, i)
    dt = (time.perf_counter() - t0) / steps
    return {"value": batch / dt, "unit": "videos/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} steps of the CPU oracle (pure-torch restatement of trainer.py:279-363), batch {batch}, "
                      f"{cfg.name} widths, fp32, {dt:.2f} s/step"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
    if a.all_ranks_on_device0:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    assert world == a.gpus or world == 1 and a.gpus == 1, f"--gpus {a.gpus} but WORLD_SIZE {world}"

    from dcvgan_amd import native, optim, trainer
    from dcvgan_amd.configs import CONFIGS, FLOPS_PER_VIDEO_STEP
    native.lib()
    cfg = CONFIGS[a.config]
    if a.batch:
        cfg = cfg.scaled(batchsize=a.batch)
    B = cfg.batchsize

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(CONFIGS[a.config], a.cpu_batch, a.cpu_steps)

    torch.manual_seed(cfg.seed)  # identical init on every rank, then made exact by a broadcast
    models = trainer.build_models(cfg, dev)
    for m in models.values():
        optim.broadcast_module(m)
    torch.manual_seed(cfg.seed + rank)  # per-rank Philox stream and data
    opts = trainer.build_optimizers(cfg, models, data_parallel=world > 1)
    runner = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=False)
    g = torch.Generator().manual_seed(cfg.seed + rank)
    lo, hi = (-0.5, 0.5) if cfg.channel == 2 else (-1.0, 1.0)
    xc = (torch.rand(B, 3, cfg.video_length, 64, 64, generator=g) * 2 - 1).to(dev)
    xg = (torch.rand(B, cfg.channel, cfg.video_length, 64, 64, generator=g) * (hi - lo) + lo).to(dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n0 = native.launch_count()
    for i in range(a.warmup):
        runner.step(xc, xg, i % cfg.video_length)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = runner.step(xc, xg, (a.warmup + i) % cfg.video_length)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    launches = native.launch_count() - n0
    losses = {k: float(v) for k, v in out.items()}

    # secondary, clearly labelled: same parameters/updates, dead D-phase generator backward elided
    minimal = None
    if not a.no_minimal:
        runner2 = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=False, elide_dead_backward=True)
        runner2.step(xc, xg, 0)
        sync()
        t1 = time.perf_counter()
        for i in range(a.steps):
            runner2.step(xc, xg, i % cfg.video_length)
        sync()
        dt2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = t.item()
        f_min = FLOPS_PER_VIDEO_STEP[a.config][1]
        minimal = {"note": "NOT the headline: D-phase fakes built without a tape (StepRunner(elide_dead_backward=True)); identical "
                           "parameter updates, FLOPs = BASELINE.md 'minimal' column",
                   "value": B * world / (dt2 / a.steps), "unit": "videos/s", "ms_per_step": dt2 / a.steps * 1e3,
                   "flops_per_video_step": f_min, "frac": f_min * (B / (dt2 / a.steps)) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
    assert all(x == x and abs(x) < 1e4 for x in losses.values()), losses

    if rank == 0:
        ms = dt / a.steps * 1e3
        vps = B * world / (dt / a.steps)
        f_step = FLOPS_PER_VIDEO_STEP[a.config][0]
        tfl = f_step * (B / (dt / a.steps)) / 1e12  # per GPU
        probe = dominant_kernel_probe(models, cfg, dev)
        traffic = None
        try:  # HBM bytes per step from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE are in KB)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            kb = sum(v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0) for v in pm.values())
            traffic = {"hbm_gb_per_step": kb * 1024 / 2 / 1e9, "algorithmic_gb_per_step": 1.06 * B, "source": "profiles/r01_pmc_summary.json "
                       "(2 steps; FETCH_SIZE raw — dword gathers are uncalibrated on gfx950, wide streams read 1/2)"}
        except Exception:
            pass
        line = {
            "metric": "videos/sec per G+D step, 16x64x64 RGB+depth", "value": vps, "unit": "videos/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config/{a.config}.yml G+D step (trainer.py:279-363), as-written schedule, fp32",
                       "per_gpu_batch": B, "global_batch": B * world, "clip": "16x64x64 RGB + geometry",
                       "parallelism": f"dp{world}", "hip_launches_per_step": launches // max(1, a.steps + a.warmup)},
            "roofline": {"bound": "mfma", "achieved": tfl, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": tfl / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                         "flops_per_video_step": f_step, "dominant_kernel": probe},
            "cpu_baseline": cpu,
            "minimal_schedule": minimal,
            "losses_last_step": losses,
            "peak_mem_gb": torch.cuda.max_memory_allocated(dev) / 1e9,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
