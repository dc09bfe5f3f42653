"""GPU, two ranks on one card (gloo): trainer.StepRunner + optim.DataParallelAdam over the real DCVGAN modules.
The ranks are fresh child processes (never a re-exec of this one)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(mode, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outs = [str(tmp_path / f"{mode}{r}.json") for r in range(2)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), "2", str(port), mode, outs[r]], env=env) for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [json.load(open(o)) for o in outs]


def test_two_ranks_distinct_data(tmp_path):
    for r in _run("distinct", tmp_path):
        assert r["collectives_per_iteration"] == [2, 2], r      # one per phase (D bucket, G bucket)
        assert r["reductions"] == 4 and r["grad_sum_relerr"] <= 1e-6, r
        assert all(abs(s - 0.5) < 1e-12 for s in r["grad_scale"]), r
        assert r["replicas_identical"], r


def test_two_ranks_same_data_equal_one_process(tmp_path):
    for r in _run("same", tmp_path):
        assert r["replicas_identical"] and r["equals_single_process"] and r["moved"], r


def test_two_ranks_same_data_equal_one_process_cl16(tmp_path):
    """The same exactness statement on the bf16 channels-last data path: its weight gradients are fp32 and every reduction in it has a fixed order, so two ranks
    on the same data reproduce the single-process run bit for bit (data parallelism needs nothing from that path but fp32 gradients in the buckets)."""
    for r in _run("same-cl16", tmp_path):
        assert r["replicas_identical"] and r["equals_single_process"] and r["moved"], r


def test_two_ranks_overlapped_reduction_same_data(tmp_path):
    """GradBucket(overlap=True) with the real modules and the lanes on: chunks reduced from their last gradient's hook, under the rest of the backward — the
    replicas stay identical and equal the single-process run bit for bit, and collectives really were launched early."""
    for r in _run("same-overlap", tmp_path):
        assert r["replicas_identical"] and r["equals_single_process"] and r["moved"], r
        assert r["early_collectives"] > 0, r


def test_two_ranks_overlapped_reduction_distinct_data(tmp_path):
    """... and on different data per rank the overlapped reduction gives the parameters of the plain one, bit for bit (same collectives on the same ranges)."""
    for r in _run("distinct-overlap", tmp_path):
        assert r["replicas_identical"] and r["equals_plain_reduction"] and r["moved"], r
        assert r["early_collectives"] > 0, r


@pytest.mark.parametrize("precision", ["fp32", "bf16cl"])
def test_bench_spawns_its_own_ranks(precision):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the ranks itself (fresh children, before any HIP call)
    and relays rank 0's single JSON line.  Two ranks share the card here, hence gloo (RCCL refuses two ranks on one device)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--all-ranks-on-device0", "--backend", "gloo",
                        "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-minimal", "--precision", precision], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 8 and r["config"]["parallelism"] == "dp2" and r["scaling"] == "weak"
    assert r["value"] > 0 and r["cpu_baseline"] is None
    # what a multi-GPU run reports besides the headline: the communicator, the collectives' own time, and the overlapped reduction's A/B
    dp = r["data_parallel"]
    assert dp["world"] == 2 and dp["backend"] == "gloo" and dp["rccl_world"] is None
    for ph in ("D", "G"):
        assert dp[f"collective_{ph}_phase"]["ms_per_step"] > 0 and dp[f"collective_{ph}_phase"]["bytes_per_step"] > 0, dp
    assert dp["dp_overlap"].get("ms_per_step", 0) > 0 and dp["dp_overlap"]["early_collectives_G_bucket"] > 0, dp["dp_overlap"]


def test_bench_under_the_drivers_launcher():
    """The command line the driver uses for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...`): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher, bench.py must not spawn ranks of its own, and exactly one JSON line comes out
    (rank 0's).  Two ranks on this one card, hence gloo and --all-ranks-on-device0; the launcher itself never touches the GPU."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--all-ranks-on-device0", "--backend", "gloo", "--batch", "4", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-minimal", "--no-as-trainer", "--no-secondary"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["config"]["global_batch"] == 8 and r["config"]["parallelism"] == "dp2"
    assert r["value"] > 0 and r["data_parallel"]["world"] == 2 and r["data_parallel"]["backend"] == "gloo"
    assert r["data_parallel"]["dp_overlap"].get("ms_per_step", 0) > 0, r["data_parallel"]["dp_overlap"]      # the leg the driver's own N > 1 run will execute too
