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
