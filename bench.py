#!/usr/bin/env python3
"""videos/s of the DCVGAN G+D training iteration on MI355X (default: config/isogd-depth.yml, B = 70 per GPU, fp32).

    python bench.py --gpus N --steps K --warmup W [--config isogd-depth|surreal-depth1|isogd-flow]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: under a launcher (WORLD_SIZE set) this process IS a rank; without one the parent — before any
HIP call — starts N fresh child processes of this script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set),
relays rank 0's JSON line and exits non-zero if any rank fails or BENCH_TIMEOUT (default 1500 s) passes.

One step = the reference trainer's iteration (trainer.py:279-363): D phase (3 D's on the real and the fake batch,
backward, 3 Adam steps — gated by num_gen_update) + G phase (fresh fakes, backward, ggen / cgen / ggen Adam steps),
fp32, synthetic clips of shape (B,3,16,64,64) + (B,Cg,16,64,64) resident in HBM.  N > 1 is weak-scaling data parallel:
every rank runs the step on its own batch and RNG stream; the gradients of a phase are all-reduced (RCCL) as ONE
bucket inside the optimiser wrapper (dcvgan_amd/optim.py).

Rank 0 prints ONE JSON line.
  roofline      bound "mfma".  achieved / frac: the WHOLE ITERATION — conv / convT / GRU FLOPs of the as-written schedule
                 x videos/s against the dense MFMA peak of the precision in use; `mfma_floor_ms` /
                `hbm_floor_ms` are the iteration's floors under that peak and under 8 TB/s (algorithmic bytes of the
                precision's storage type).  `dominant_kernel`: cgen.up_blocks.5 forward, the step's largest single GEMM
                (2 * M * OC * K FLOP per launch), timed alone with HIP events on the stream it is launched on.  `traffic`
                (HBM bytes per iteration / per launch of the dominant kernel from rocprofv3 PMC passes) cannot be
                collected from inside the process: it is read from the committed profile of the SAME kernel sources
                (sha256), config and batch size, else null.
  cpu_baseline  the CPU oracle's step (pure-torch restatement of the reference trainer, pinned to the reference by
                tests/golden) on this host's cores — rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, dense f32 MFMA
PEAK_BF16_MFMA_TFLOPS = 16 * 157.3   # same guide: the bf16 MFMA rate is 16x the f32 one (~2.5 PFLOP/s dense)
PEAK_HBM_GBPS = 8000.0          # same guide, HBM3E


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="isogd-depth")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dp-overlap", action="store_true", help="N > 1: skip the secondary `data_parallel.dp_overlap` leg (GradBucket(overlap=True) timed after the headline)")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing the DP path on one GPU)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "f32x6", "bf16cl"],
                    help="bf16: MFMA products in bf16 with fp32 accumulation in the large GEMM kernels (tensors, weights, statistics and "
                         "optimiser state stay fp32) — a secondary throughput line for BASELINE configs[2]/[4], not the headline.  "
                         "f32x6 (experimental): fp32 operands split exactly into three bf16 pieces, six bf16 MFMA products, fp32 accumulation.  "
                         "bf16cl: the bf16 DATA path of BASELINE configs[2]/[4] — activations and gradients bf16 channels-last in HBM, fp32 master "
                         "weights / statistics / optimiser (dcvgan_amd/ops_cl.py)")
    ap.add_argument("--no-as-trainer", action="store_true", help="skip the secondary `as_trainer` key: the same iteration driven the way the reference's "
                                                                 "trainer.py drives it (4 host syncs on the losses, :326-328,363; a fresh pinned host batch "
                                                                 ".to(device) per iteration, :293-297), timed AFTER the headline region")
    ap.add_argument("--minimal", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-minimal", action="store_true", help="skip the secondary `minimal_schedule` key (the schedule with the dead D-phase generator backward elided, "
                                                              "timed AFTER the headline region) — profile runs use this so that the trace holds the as-written schedule only")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary_precisions` key: short runs of the 16-bit paths (bf16 channels-last data path on "
                                                                "surreal-depth1 — the config BASELINE names for bf16 — and on the headline config; fp32-on-bf16 on the headline config), timed AFTER everything else")
    ap.add_argument("--cpus", type=int, default=0, help="pin this rank to K host CPUs (host-headroom probe: 8 ranks on a 16-CPU share have 2 each)")
    return ap.parse_args()


def gpu_numa_cpus(local_rank, local_world):
    """CPUs of the NUMA node GPU `local_rank` hangs off, this rank's share of them.  The node comes from the device's own PCI address
    (torch's device properties -> /sys/bus/pci/devices/<address>/numa_node), so no enumeration order is assumed; ranks whose GPUs share a
    node split its CPUs.  None when anything is missing (containers often hide sysfs; older torch has no PCI fields): the rank then keeps
    the affinity it was started with.  Needs the device runtime, so it runs after torch.cuda.set_device."""
    def node_of(i):
        p = torch.cuda.get_device_properties(i)
        addr = "%04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id")), int(getattr(p, "pci_bus_id")), int(getattr(p, "pci_device_id")))
        return int(open(f"/sys/bus/pci/devices/{addr}/numa_node").read())
    try:
        n_dev = torch.cuda.device_count()
        if local_rank >= n_dev:
            return None
        node = node_of(local_rank)
        if node < 0:
            return None
        cpus = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus += list(range(int(lo), int(hi or lo) + 1))
        cpus = sorted(set(cpus) & os.sched_getaffinity(0))
        peers = [r for r in range(min(local_world, n_dev)) if r == local_rank or node_of(r) == node]
        if not cpus or local_rank not in peers:
            return None
        share = max(1, len(cpus) // len(peers))
        i = peers.index(local_rank)
        return cpus[i * share:(i + 1) * share] or None
    except Exception:
        return None


def spawn_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher.  The parent never touches the GPU (torch.cuda.device_count() does not
    initialise HIP on this image): the ranks are fresh children, never a re-exec of a process that holds the device."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if a.gpus > have and not a.all_ranks_on_device0:
        print(f"bench.py: --gpus {a.gpus} but only {have} visible (use --all-ranks-on-device0 --backend gloo to rehearse on one card)", file=sys.stderr)
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + float(os.environ.get("BENCH_TIMEOUT", "1500"))
    rc, out0 = 0, b""
    try:
        import threading
        buf = []
        rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)   # drain rank 0's pipe while we poll
        rd.start()
        live = set(range(a.gpus))
        while live and rc == 0:
            for r in sorted(live):
                c = procs[r].poll()
                if c is not None:
                    live.discard(r)
                    if c != 0:
                        print(f"bench.py: rank {r} exited with code {c}", file=sys.stderr)
                        rc = c if c > 0 else 1
            if time.time() > deadline:
                print("bench.py: BENCH_TIMEOUT passed", file=sys.stderr)
                rc = 124
            if live and rc == 0:
                time.sleep(0.2)
        if rc == 0:
            rd.join(10)
            out0 = buf[0] if buf else b""
    finally:
        for p in procs:            # exactly the PIDs started above
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(10)
            except Exception:
                pass
    lines = [l for l in out0.decode(errors="replace").splitlines() if l.startswith("{")]
    if rc == 0 and not lines:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    if rc == 0:
        print(lines[-1], flush=True)
    return rc


def dominant_kernel_probe(models, cfg, dev):
    """The step's largest single launch: cgen.up_blocks[5] forward — ConvTranspose2d 128 -> 64, 4x4 s2 p1 on (F,128,32,32),
    4.29 GFLOP/video: the 4 stride-parity classes run as ONE gather-GEMM launch.  Timed with HIP
    events recorded on the stream the library launches on (the current stream)."""
    from dcvgan_amd import layers, native, ops
    conv = models["cgen"].up_blocks[5].main[0]
    F_ = cfg.batchsize * cfg.video_length
    x = torch.randn(F_, conv.in_channels, 32, 32, device=dev)
    g = layers.geom_of(conv)
    from dcvgan_amd import ops_cl
    if ops_cl.active():       # the same layer on the bf16 channels-last path
        with torch.no_grad():
            x = ops_cl.from_f32(x)
        ops = ops_cl
    with torch.no_grad():
        for _ in range(3):
            ops.conv(x, conv.weight, g)     # the first call also packs the weights (cached afterwards, as in the step)
        name = native.lib().dcv_debug_last_kernel().decode()
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record(s)
        for _ in range(reps):
            ops.conv(x, conv.weight, g)
        e1.record(s)
        e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * F_ * 64 * 64 * conv.out_channels * conv.in_channels * 4  # 2x2 taps per output position
    return {"layer": "cgen.up_blocks.5 forward", "kernel": name, "ms": ms, "gflop_per_launch": flops / 1e9, "tflops": flops / ms / 1e9}


def committed_traffic(batch, kernel):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes (tools/round_profile.sh ->
    tools/dominant_pmc.py; FETCH_SIZE doubled for the 16-byte-per-lane operand streams as the MI355X guide prescribes).  PMC counters
    cannot be collected from inside this process, so the figure is only reported when the profile was taken from THIS SOURCE — the sha256
    of dcvgan_amd/csrc (native.csrc_digest) stored in the profile equals the working tree's — at THIS batch size from the kernel instance
    this run just launched (`dcv_debug_last_kernel`); otherwise null — a stale number is worse than none."""
    import glob
    from dcvgan_amd import native
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dominant_kernel_pmc.json")), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        inst = lambda k: k.split(" (")[0].replace(" ", "")      # template instance without the library's "(tile, classes)" annotation
        if t.get("csrc_sha256") != native.csrc_digest():
            return None, {"note": f"profiles/{os.path.basename(f)} was taken from other kernel sources (csrc_sha256 {str(t.get('csrc_sha256'))[:12]} at git head "
                                  f"{t.get('git_head')}, this build {native.csrc_digest()[:12]}): not reported"}
        if int(t.get("batch", -1)) == int(batch) and inst(t.get("kernel", "")) == inst(kernel):
            return t["hbm_bytes_per_launch"], {"algorithmic_bytes_per_launch": t["algorithmic_bytes_per_launch"], "unit": "bytes per launch",
                                               "traffic_over_algorithmic": t.get("traffic_over_algorithmic"), "profiled_at_git_head": t.get("git_head"),
                                               "source": f"profiles/{os.path.basename(f)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2)"}
        return None, {"note": f"profiles/{os.path.basename(f)} was taken from kernel {t.get('kernel')!r} at batch {t.get('batch')}; this run launched {kernel!r} at batch {batch}"}
    return None, None


def committed_step_traffic(config, batch, cl=False):
    """HBM bytes per ITERATION from the newest committed whole-step PMC passes (tools/pmc_step.sh), under the same rule: same source digest,
    same config and batch, else null.  FETCH_SIZE doubled as the guide prescribes for 16-byte-per-lane streams (every large read of the step
    is one: LDS-DMA granules, 16-byte elementwise groups); the raw sum is given beside it."""
    import glob
    from dcvgan_amd import native
    pattern = "r*_pmc_step_bf16cl_summary.json" if cl else "r*_pmc_step_summary.json"      # cl: the bf16 channels-last path's passes (digest over its sources too)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        meta = t.get("__meta__") or {}
        if meta.get("csrc_sha256") != native.csrc_digest(cl=cl) or meta.get("config") != config or int(meta.get("batch", -1)) != int(batch):
            return None, {"note": f"profiles/{os.path.basename(f)}: taken from other sources / config / batch ({meta or 'no metadata'}): not reported"}
        steps = max(1, int(meta.get("steps", 1)))
        fe = sum(v.get("FETCH_SIZE", 0.0) for k, v in t.items() if k != "__meta__") * 1024 / steps
        wr = sum(v.get("WRITE_SIZE", 0.0) for k, v in t.items() if k != "__meta__") * 1024 / steps
        return 2 * fe + wr, {"unit": "bytes per iteration", "fetch_size_bytes_raw": fe, "write_size_bytes": wr, "profiled_at_git_head": meta.get("git_head"),
                             "source": f"profiles/{os.path.basename(f)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2)"}
    return None, None


def stress_leg(dev, half, B=100, steps=5):
    """BASELINE configs[4]'s discriminator shape: vdis + gdis (config/isogd-flow.yml widths: ndf 64 / 32, Noise sigma 0.2 on vdis) forward + backward on 32 x 128 x 128 flow
    clips, on the 16-bit channels-last data path with `half` = "bf16" or "fp16" elements (3 x (55.1 + 13.6) GFLOP per clip).  Besides the rate: the largest
    pre-BatchNorm magnitude (fp16's largest finite value is 65504) and how many parameter-gradient elements came out exactly zero (fp16's smallest normal number is 6.1e-5;
    the cotangent of a mean over the logits starts near 1e-6 at this batch)."""
    from dcvgan_amd import discriminator as D, layers, ops_cl
    ops_cl.enable(True, half=half)
    try:
        torch.manual_seed(0)
        vdis = D.VideoDiscriminator(2, 3, True, 0.2, 64).to(dev)
        gdis = D.GradientDiscriminator(2, 3, False, 0.2, 32).to(dev)
        xc = (torch.rand(B, 3, 32, 128, 128, device=dev) * 2 - 1).requires_grad_(True)
        xg = (torch.rand(B, 2, 32, 128, 128, device=dev) - 0.5).requires_grad_(True)

        def step():
            for m in (vdis, gdis):
                m.zero_grad()
            yv, yg = vdis(xg, xc), gdis(xg, xc)
            (yv.mean() + yg.mean()).backward()
            return yv, yg
        layers.PREBN_TAP = []
        yv, yg = step()
        torch.cuda.synchronize()
        peak = max(float(t) for t in layers.PREBN_TAP)
        layers.PREBN_TAP = None
        gs = [p.grad for m in (vdis, gdis) for p in m.parameters() if p.grad is not None]
        zero, tot = sum(int((g == 0).sum()) for g in gs), sum(g.numel() for g in gs)
        finite = bool(torch.isfinite(yv).all() and torch.isfinite(yg).all() and all(bool(torch.isfinite(g).all()) for g in gs))
        for _ in range(4):      # untimed: the caching allocator reaches its steady state (tensors the weight gradients' companion stream still reads are returned late: the
            step()              # first steps of a new shape allocate — a device synchronisation each; with one warm-up step the bf16 leg, which runs first, read 68 ms for 52)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / steps
        gf = 3 * (55.1 + 13.6) * B
        return {"value": B / ms * 1e3, "unit": "clips/s (vdis + gdis forward + backward)", "ms_per_step": ms, "per_gpu_batch": B, "tflops": gf / ms, "frac_of_its_mfma_peak": gf / ms / PEAK_BF16_MFMA_TFLOPS,
                "all_finite": finite, "largest_pre_batchnorm_magnitude": peak, "fp16_largest_finite": 65504.0,
                "parameter_gradient_elements_exactly_zero": zero, "parameter_gradient_elements": tot, "peak_mem_gb": torch.cuda.max_memory_allocated(dev) / 1e9}
    finally:
        layers.PREBN_TAP = None
        ops_cl.enable(False, half="bf16")


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


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
    return {"value": batch / dt, "unit": "videos/s", "cores": torch.get_num_threads(), "host_cpus_visible": os.cpu_count(),
            "cores_note": "cores = the threads the oracle ran on = this process's CPU quota (cgroup cpu.max / affinity); host_cpus_visible = what the box shows",
            "kind": "port", "cpu": cpu_model(),
            "sample": f"{steps} steps of the CPU oracle (pure-torch restatement of trainer.py:279-363, pinned to the reference by tests/golden), "
                      f"batch {batch}, {cfg.name} widths, fp32, {dt:.2f} s/step"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    numa_cpus = None
    if a.cpus:
        mine = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, set(mine[(rank * a.cpus) % len(mine):][:a.cpus]) or set(mine[:a.cpus]))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
    if a.all_ranks_on_device0:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if not a.cpus and not a.all_ranks_on_device0:
        # every rank (spawned by us or by torch.distributed.run; also the single rank of an N = 1 run) enqueues ~900 launches per iteration from one Python
        # thread: keep that thread and its allocations on the NUMA node its GPU hangs off
        numa_cpus = gpu_numa_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        if numa_cpus:
            os.sched_setaffinity(0, set(numa_cpus))
    # N ranks share the host's CPU quota (a rank pinned to its own CPU set uses that set)
    torch.set_num_threads(max(1, len(numa_cpus)) if numa_cpus else max(1, host_threads() // max(1, world)))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    assert world == a.gpus or world == 1 and a.gpus == 1, f"--gpus {a.gpus} but WORLD_SIZE {world}"

    from dcvgan_amd import native, optim, trainer
    from dcvgan_amd.configs import ALGORITHMIC_HBM_GB_PER_VIDEO_ITERATION, CONFIGS, flops_per_video_iteration
    native.lib()
    cfg = CONFIGS[a.config]
    if a.batch:
        cfg = cfg.scaled(batchsize=a.batch)
    B = cfg.batchsize

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(CONFIGS[a.config], a.cpu_batch, a.cpu_steps)

    if a.precision == "bf16cl":
        from dcvgan_amd import ops_cl
        ops_cl.enable(True)
    else:
        native.set_precision(a.precision)
    # f32x6 runs on the bf16 pipe with six products per fp32 product: its roofline is the bf16 peak / 6 (the weight gradients stay on the fp32 pipe)
    peak = {"fp32": PEAK_FP32_MFMA_TFLOPS, "f32x6": PEAK_BF16_MFMA_TFLOPS / 6.0}.get(a.precision, PEAK_BF16_MFMA_TFLOPS)
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

    def timed(run, steps, first):
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            out = run.step(xc, xg, (first + i) % cfg.video_length)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, out

    n0 = native.launch_count()
    for i in range(a.warmup):
        runner.step(xc, xg, i % cfg.video_length)
    dt, out = timed(runner, a.steps, a.warmup)
    launches = native.launch_count() - n0
    peak_mem_headline = torch.cuda.max_memory_allocated(dev)      # the headline iteration's own peak: read before any secondary leg allocates
    losses = {k: float(v) for k, v in out.items()}
    assert all(x == x and abs(x) < 1e4 for x in losses.values()), losses

    # data parallel: what the communicator really is, the collectives' own time, and the overlapped reduction timed against the headline's — all AFTER the headline region
    dp_info = None
    if world > 1:
        dp_info = {"world": dist.get_world_size(), "backend": dist.get_backend(), "rccl_world": dist.get_world_size() if dist.get_backend() == "nccl" else None,
                   "rank0_device": str(dev), "devices_visible": torch.cuda.device_count()}
        try:
            bD, bG = opts["idis"].bucket, opts["ggen"].bucket
            bD.timed, bG.timed = [], []
            n_probe = 3
            for i in range(n_probe):
                runner.step(xc, xg, i % cfg.video_length)
            sync()
            for name, b in (("D", bD), ("G", bG)):
                ms = [e0.elapsed_time(e1) for e0, e1, _ in b.timed]
                nbytes = sum(n for _, _, n in b.timed) // max(1, n_probe)
                per = sum(ms) / max(1, n_probe)
                dp_info[f"collective_{name}_phase"] = {"ms_per_step": per, "bytes_per_step": nbytes, "collectives_per_step": len(ms) // max(1, n_probe),
                                                         "bus_gbps": (2.0 * (world - 1) / world) * nbytes / max(per, 1e-9) / 1e6}
            bD.timed = bG.timed = None
            dp_info["note"] = ("collective_*: HIP events on the compute stream around the all-reduce of that phase's gradient bucket (issue ... reduced), mean of 3 "
                               "iterations after the headline region; bus_gbps = 2 (N - 1) / N x bytes / time (ring all-reduce convention)")
        except Exception as e:
            dp_info["collective_error"] = f"{type(e).__name__}: {e}"[:300]

    # secondary, clearly labelled, timed AFTER the headline region (--no-minimal skips it): identical parameter updates, the dead D-phase generator backward elided
    minimal = None
    if not a.no_minimal:
        runner2 = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=False, elide_dead_backward=True)
        runner2.iteration = runner.iteration
        runner2.step(xc, xg, 0)
        dt2, _ = timed(runner2, a.steps, 1)
        f_min = flops_per_video_iteration(cfg, True)
        minimal = {"note": "NOT the headline: D-phase fakes built without a tape (StepRunner(elide_dead_backward=True)); identical parameter "
                           "updates, FLOPs = BASELINE.md 'minimal' column",
                   "value": B * world / (dt2 / a.steps), "unit": "videos/s", "ms_per_step": dt2 / a.steps * 1e3,
                   "flops_per_video_step": f_min, "frac": f_min * (B / (dt2 / a.steps)) / 1e12 / peak}

    # secondary: the iteration as the reference's trainer.py drives it — the three D losses and the G loss are read on the host with
    # .cpu().item() where trainer.py:326-328,363 read them, and every iteration starts from a fresh pinned host batch .to(device)
    # (trainer.py:293-297; the DataLoader there pins memory, train.py:101-109).  The headline above stays the HBM-resident, sync-free figure.
    as_trainer = None
    if not a.no_as_trainer and world == 1:      # (N = 1 only: a secondary key must never be able to stall or fail a multi-rank headline run)
        import itertools
        from dcvgan_amd.dataprep import DevicePrefetcher
        pin = [(xc.cpu().pin_memory(), xg.cpu().pin_memory()) for _ in range(2)]      # a loader's double buffer (DataLoader(pin_memory=True), train.py:101-109)

        def leg(prefetch):
            r3 = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=True)
            r3.iteration = runner.iteration
            feed = DevicePrefetcher(itertools.cycle(pin), dev) if prefetch else None

            class _AsTrainer:
                def step(self, _xc, _xg, t):
                    if feed is not None:
                        dc, dg = next(feed)                       # already on the device: the trainer's .to(device) is a no-op
                    else:
                        hc, hg = pin[r3.iteration & 1]
                        dc, dg = hc.to(dev, non_blocking=True), hg.to(dev, non_blocking=True)
                    return r3.step(dc, dg, t)
            at = _AsTrainer()
            at.step(None, None, 0)
            d, _ = timed(at, a.steps, 1)
            return d
        try:
            dt3 = leg(True)
            dt4 = leg(False)
            as_trainer = {"note": "NOT the headline: sync_losses=True (four .cpu().item() host reads per iteration where trainer.py:326-328,363 has them; the loss objects "
                                  "carry an event-guarded pinned host copy, dcvgan_amd.loss.HostMirroredLoss, so a read waits for the loss kernels, not for the "
                                  "backward + Adam enqueued behind them) + a fresh pinned host batch copied to the device every iteration (trainer.py:293-297) "
                                  "through dcvgan_amd.dataprep.DevicePrefetcher (side stream, one batch ahead)",
                          "value": B * world / (dt3 / a.steps), "unit": "videos/s", "ms_per_step": dt3 / a.steps * 1e3,
                          "slower_than_headline_pct": (dt3 / dt - 1.0) * 100.0,
                          "h2d_mb_per_step": (xc.numel() + xg.numel()) * 4 / 1e6,
                          "without_prefetcher": {"note": "the batch copied with .to(device, non_blocking=True) on the compute stream at the top of the iteration instead",
                                                 "ms_per_step": dt4 / a.steps * 1e3, "slower_than_headline_pct": (dt4 / dt - 1.0) * 100.0}}
        except Exception as e:          # a secondary key: the headline line is still printed
            as_trainer = {"error": f"{type(e).__name__}: {e}"[:300]}

    # secondary, clearly labelled: the 16-bit paths, each on freshly built models of its config, timed after everything above
    secondary = None
    if not a.no_secondary and world == 1 and a.precision == "fp32" and a.config == "isogd-depth" and not a.batch:
        from dcvgan_amd import ops_cl
        secondary = {}
        legs = (("bf16cl", "surreal-depth1"), ("bf16cl", "isogd-depth"), ("f32x6", "isogd-depth"))
        for prec, cname in legs:
            c2 = CONFIGS[cname]
            torch.cuda.empty_cache()       # the headline's cached blocks (fp32 tensors of another config) go back to the driver before a leg allocates its own
            torch.cuda.reset_peak_memory_stats(dev)
            try:
                if prec == "bf16cl":
                    ops_cl.enable(True)
                else:
                    native.set_precision(prec)
                torch.manual_seed(c2.seed)
                m2 = trainer.build_models(c2, dev)
                for m in m2.values():
                    optim.broadcast_module(m)
                torch.manual_seed(c2.seed + rank)
                o2 = trainer.build_optimizers(c2, m2, data_parallel=world > 1)
                r2 = trainer.StepRunner(c2, m2, o2, trainer.build_loss(c2), sync_losses=False)
                g2 = torch.Generator().manual_seed(c2.seed + rank)
                lo2, hi2 = (-0.5, 0.5) if c2.channel == 2 else (-1.0, 1.0)
                xc2 = (torch.rand(c2.batchsize, 3, c2.video_length, 64, 64, generator=g2) * 2 - 1).to(dev)
                xg2 = (torch.rand(c2.batchsize, c2.channel, c2.video_length, 64, 64, generator=g2) * (hi2 - lo2) + lo2).to(dev)
                for i in range(6):
                    out2 = r2.step(xc2, xg2, i)
                ns = max(4, min(a.steps, 12))
                sync()
                t0 = time.perf_counter()
                for i in range(ns):
                    out2 = r2.step(xc2, xg2, (3 + i) % c2.video_length)
                sync()
                d2 = time.perf_counter() - t0
                if world > 1:
                    t = torch.tensor([d2], device=dev, dtype=torch.float64)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    d2 = t.item()
                l2 = {k: float(v) for k, v in out2.items()}
                assert all(x == x and abs(x) < 1e4 for x in l2.values()), l2
                pk2 = PEAK_BF16_MFMA_TFLOPS if prec == "bf16cl" else PEAK_BF16_MFMA_TFLOPS / 6.0
                f2 = flops_per_video_iteration(c2)
                secondary[f"{prec}:{cname}"] = {"value": c2.batchsize * world / (d2 / ns), "unit": "videos/s", "ms_per_step": d2 / ns * 1e3, "steps": ns, "per_gpu_batch": c2.batchsize,
                                                "frac_of_its_mfma_peak": f2 * (c2.batchsize / (d2 / ns)) / 1e12 / pk2, "peak_tflops": pk2,
                                                "mfma_floor_ms": f2 * c2.batchsize / (pk2 * 1e12) * 1e3,
                                                "hbm_floor_ms": ALGORITHMIC_HBM_GB_PER_VIDEO_ITERATION * (0.5 if prec == "bf16cl" else 1.0) * c2.batchsize / PEAK_HBM_GBPS * 1e3,
                                                "peak_mem_gb": torch.cuda.max_memory_allocated(dev) / 1e9, "losses_last_step": l2}
                if prec == "bf16cl":
                    tr2, trd2 = committed_step_traffic(cname, c2.batchsize, cl=True)
                    secondary[f"{prec}:{cname}"].update({"traffic": tr2, "traffic_detail": trd2})
                del m2, o2, r2, xc2, xg2
            except Exception as e:      # a secondary leg may fail; the headline line is still printed
                secondary[f"{prec}:{cname}"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                ops_cl.enable(False)
                native.set_precision("fp32")
            torch.cuda.empty_cache()
        # BASELINE configs[4]: the 32 x 128 x 128 discriminator shape in fp16 (as worded) and in bf16, one after the other in this call
        for half in ("bf16", "fp16"):
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(dev)
            try:
                secondary[f"{half}cl:stress_d_32x128x128"] = stress_leg(dev, half)
            except Exception as e:
                secondary[f"{half}cl:stress_d_32x128x128"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        secondary["note"] = ("NOT the headline: bf16cl = bf16 channels-last data path (activations / gradients bf16 in HBM, fp32 masters, statistics, accumulation, optimiser; "
                             "tolerance tests tests/test_cl16_gpu.py); stress_d_32x128x128 = BASELINE configs[4]'s discriminator shape (vdis + gdis forward + backward on 32 x 128 x 128 flow clips, B = 100) on the same path with bf16 and with fp16 elements (dcv_clf16_*; tests/test_fp16_gpu.py); f32x6 = fp32 emulated on the bf16 matrix pipe in the forward / data-gradient GEMMs (3-way bf16 split, six products, sign-alternating accumulation; the fp32 parity suites pass with it as the process default; experimental, not the default); DESIGN §8")

    if rank == 0:
        per_step = dt / a.steps
        vps = B * world / per_step
        f_step = flops_per_video_iteration(cfg)
        step_tflops = f_step * (B / per_step) / 1e12        # per GPU
        probe = dominant_kernel_probe(models, cfg, dev)
        gb_step = ALGORITHMIC_HBM_GB_PER_VIDEO_ITERATION * B
        traffic, traffic_detail = committed_traffic(B, probe["kernel"]) if a.precision == "fp32" else (None, None)
        step_traffic, step_traffic_detail = committed_step_traffic(a.config, B) if a.precision == "fp32" else (None, None)
        gating = "" if cfg.num_gen_update == 1 else f", D update every {cfg.num_gen_update} iterations (FLOPs averaged over the cycle)"
        line = {
            "metric": "videos/sec per G+D step, 16x64x64 RGB+depth" if cfg.channel == 1 else "videos/sec per G+D step, 16x64x64 RGB+flow",
            "value": vps, "unit": "videos/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": per_step * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "f32x6": "f32 (emulated: 3 x bf16 split, 6 products)"}.get(a.precision, "bf16"), "data": "synthetic",
            "config": {"workload": f"config/{a.config}.yml G+D iteration (trainer.py:279-363), as-written schedule, "
                                   + {"fp32": "fp32", "bf16": "bf16 MFMA products / fp32 accumulation, storage, statistics and optimiser (throughput mode)",
                                      "f32x6": "fp32 emulated on the bf16 matrix pipe in the forward / data-gradient GEMMs (3-way bf16 split, six products, fp32 accumulation; experimental)",
                                      "bf16cl": "bf16 channels-last DATA path: bf16 activations and gradients in HBM, fp32 master weights / statistics / accumulation / optimiser (throughput mode)"}[a.precision] + gating,
                       "per_gpu_batch": B, "global_batch": B * world, "clip": f"16x64x64 RGB + {cfg.channel}-channel {cfg.geometric_info}",
                       "parallelism": f"dp{world}", "hip_launches_per_step": launches // max(1, a.steps + a.warmup),
                       "rank0_cpus": sorted(os.sched_getaffinity(0)) if len(os.sched_getaffinity(0)) <= 64 else len(os.sched_getaffinity(0)),
                       "rank0_cpus_from_gpu_numa_node": bool(numa_cpus)},
            "roofline": {"bound": "mfma", "achieved": step_tflops, "peak": peak, "unit": "TFLOP/s",
                         "frac": step_tflops / peak, "traffic": step_traffic, "traffic_detail": step_traffic_detail,
                         "definition": "round 4 on: achieved / frac = the WHOLE ITERATION (conv / convT / GRU FLOPs of the as-written schedule x videos/s against the "
                                       "dense MFMA peak of the precision in use) — the same quantity as round 1's roofline.frac and rounds 2-3's roofline.step.frac; "
                                       "the dominant kernel timed alone (rounds 2-3's roofline.frac) is roofline.dominant_kernel; traffic = HBM bytes per ITERATION "
                                       "from the committed PMC passes of this very source (else null), dominant_kernel.traffic = per launch of that kernel",
                         "flops_per_video_step": f_step,
                         "mfma_floor_ms": f_step * B / (peak * 1e12) * 1e3,
                         "hbm_floor_ms": gb_step * (0.5 if a.precision == "bf16cl" else 1.0) / PEAK_HBM_GBPS * 1e3,
                         "dominant_kernel": dict(probe, achieved=probe["tflops"], frac=probe["tflops"] / peak, traffic=traffic, traffic_detail=traffic_detail),
                         "step": {"achieved": step_tflops, "frac": step_tflops / peak, "flops_per_video_step": f_step},
                         "hbm": {"algorithmic_gb_per_step": gb_step, "achieved_gbps": gb_step / per_step, "peak_gbps": PEAK_HBM_GBPS,
                                 "frac": gb_step / per_step / PEAK_HBM_GBPS}},
            "env_toggles": {k: v for k, v in sorted(os.environ.items()) if k.startswith("DCV_")} or None,      # every A/B switch that was set for this run (none in the driver's)
            "cpu_baseline": cpu,
            "data_parallel": dp_info,
            "minimal_schedule": minimal,
            "as_trainer": as_trainer,
            "secondary_precisions": secondary,
            "losses_last_step": losses,
            "peak_mem_gb": peak_mem_headline / 1e9,
        }
    else:
        line = None
    if world > 1 and not a.no_dp_overlap:
        # Secondary `data_parallel.dp_overlap`: the same iteration with GradBucket(overlap=True) — per-model chunks reduced from the hook of their last gradient, on a
        # communication stream, under the rest of the backward — timed after everything else so that ONE multi-GPU run yields the A/B.  It has never met a world > 1 on
        # RCCL before the driver's run, so a watchdog guards it: if the leg is not done in time, rank 0 prints the headline line as it stands and every rank leaves.
        import threading
        budget = float(os.environ.get("BENCH_DP_OVERLAP_TIMEOUT", "90"))

        def bail():
            if line is not None:
                line["data_parallel"]["dp_overlap"] = {"error": f"not finished within {budget:.0f} s: abandoned by the watchdog (the headline above is unaffected)"}
                print(json.dumps(line), flush=True)
            os._exit(0)
        dog = threading.Timer(budget, bail)
        dog.daemon = True
        dog.start()
        res = None
        try:
            for b_ in {id(o.bucket): o.bucket for o in opts.values()}.values():      # the headline's buckets let go of the parameters (their hooks would fire beside the new ones)
                for h_ in b_._hooks:
                    h_.remove()
                b_._hooks = []
            o3 = trainer.build_optimizers(cfg, models, data_parallel=True, overlap=True)
            r3 = trainer.StepRunner(cfg, models, o3, trainer.build_loss(cfg), sync_losses=False)
            r3.iteration = runner.iteration
            for i in range(3):      # the arrival order of a chunk's gradients is learned in the first backward; overlapped from the second
                r3.step(xc, xg, i)
            ns = max(3, min(a.steps, 10))
            d3, _ = timed(r3, ns, 3)
            bG3 = o3["ggen"].bucket
            res = {"ms_per_step": d3 / ns * 1e3, "value": B * world / (d3 / ns), "unit": "videos/s", "steps": ns,
                   "vs_headline_pct": (d3 / ns / (dt / a.steps) - 1.0) * 100.0, "early_collectives_G_bucket": bG3.early, "collectives_G_bucket": bG3.collectives,
                   "note": "NOT the headline: GradBucket(overlap=True) (dcvgan_amd/optim.py), fresh Adam state on the same models; negative vs_headline_pct = faster"}
        except Exception as e:
            res = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog.cancel()
        if line is not None:
            line["data_parallel"]["dp_overlap"] = res
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()     # rank 0 runs the dominant-kernel probe and prints before anyone tears the communicator down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
