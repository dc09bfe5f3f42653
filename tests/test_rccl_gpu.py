"""GPU: RCCL itself.  The data-parallel tests rehearse on gloo (two ranks cannot share a card under RCCL); here the `nccl` backend is
opened for real with world_size 1 on the card and the headline config's G-phase bucket (55 MB) goes through GradBucket.reduce ->
dist.all_reduce -> Adam on the re-pointed slices (trainer.py:356-359 under SURVEY §8(e)'s wrapper).  Fresh child process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world1_bucket_allreduce():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_probe.py"), "5"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["backend"] == "nccl" and r["world_size"] == 1
    assert 50e6 < r["bucket_bytes"] < 60e6 and r["tensors"] == 55         # ggen + cgen of isogd-depth: 55.1 MB
    assert r["collectives"] == r["reductions"] == 7                       # one collective per reduction: the bucket is ONE message
    assert r["reduced_equals_local"] and r["storages_after_reduce"] == 1   # sum over one rank; every .grad is a slice of the flat buffer
    assert r["adam_moved_fraction"] > 0.99
    assert r["ms_per_reduction"] < 50.0
    d = os.environ.get("DCV_REPORT_DIR")
    if d:
        os.makedirs(d, exist_ok=True)
        json.dump(r, open(os.path.join(d, "rccl_world1_bucket.json"), "w"), indent=1)


def test_rccl_world1_overlapped_chunks():
    """optim.GradBucket(overlap=True) on the card: chunk collectives launched during the backward on the communication stream, Adam and the next backward ordered behind
    them; three StepRunner iterations equal the step-time reduction and the plain run bit for bit (tools/rccl_overlap_probe.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_overlap_probe.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["backend"] == "nccl" and r["world_size"] == 1
    assert r["sync_equals_plain"] and r["overlap_equals_sync"], r
    assert r["sync"]["early"] == 0 and r["overlap"]["early"] >= 2, r          # from the second iteration on, chunks go out during the backward
    assert sum(r["overlap"]["chunks"]) >= 3, r                                    # G bucket: ggen and cgen apart; D bucket: the small models merged
    d = os.environ.get("DCV_REPORT_DIR")
    if d:
        os.makedirs(d, exist_ok=True)
        json.dump(r, open(os.path.join(d, "rccl_world1_overlap.json"), "w"), indent=1)
