"""Calibration of oracle/stepcheck.py: the REFERENCE's own arithmetic (the pinned fp32 torch-CPU oracle) as the checked implementation, at
full width, against the teacher-forced fp64 oracle with its own activation pattern.  Shows what any fp32 implementation scores under the
bars the HIP path is held to (profiles/r03_step_parity/reference_fp32_cpu.txt).  CPU only:
    python3 tools/stepcheck_reference.py step_fullwidth_isogd_depth.npz step_fullwidth_surreal_depth1.npz step_fullwidth_isogd_flow.npz"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_stepcheck_cpu import _Fp32Runner
from tests import fullwidth as FW, goldenio as G
from oracle import stepcheck as SC
torch.set_num_threads(8)
for fixture in sys.argv[1:]:
    fx = G.load(fixture)
    cfg, models = FW.same_seed_models(fx)
    cfg.num_gen_update = int(fx["meta/num_gen_update"]); cfg.lr = {m: float(fx[f"meta/lr/{m}"]) for m in G.MODELS}
    B = cfg.batchsize
    gd = torch.Generator().manual_seed(int(fx["meta/seed_data"]))
    lo, hi = (-0.5, 0.5) if cfg.channel == 2 else (-1.0, 1.0)
    xc = torch.rand(B, 3, 16, 64, 64, generator=gd) * 2 - 1
    xg = torch.rand(B, cfg.channel, 16, 64, 64, generator=gd) * (hi - lo) + lo
    torch.manual_seed(int(fx["meta/seed_run"]))
    run = _Fp32Runner(cfg, FW.states_of(models))
    forced = SC.ForcedStepOracle(cfg, run.so.rng.log)
    for it in range(int(fx["meta/iters"])):
        res = SC.checked_iteration(run, run.models, run.opts, forced, run, xc, xg, xc, xg, int(fx["meta/t_rands"][it]), cfg.lr)
        rows = res["rows"]
        print(fixture, it+1, "loss_rel %.2e buf %.2e" % (res["loss_rel"], res["buffers_rel"]), "rel_l2", SC.worst(rows, "rel_l2"),
              "sens", sum(r["n_sensitive"] for r in rows), "off", sum(r["n_sensitive_off"] for r in rows), "kinks", res["kink_flips"], res["kink_total"], res["kink_worst_call"], flush=True)
