"""The REFERENCE's own fp32 arithmetic (the pinned torch-CPU oracle) under oracle/stepcheck.py over exactly the cells of tools/step_seed_sweep.py:
same (config, seed, batch, iteration) -> same initial models, real batch, frame indices and random draws (the HIP sweep replays the draws of this very run).
One line per cell in the HIP sweep's format, plus where the furthest (Leaky)ReLU pattern mismatch sits (BatchNorm layer, channel, |mean| / std of its input).
CPU only:  python3 tools/step_seed_sweep_reference.py --seeds 6 --iters 2 --batch 2 [--seed0 0] [--configs ...] [--threads 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--configs", nargs="*", default=["isogd-depth", "surreal-depth1", "isogd-flow"])
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from dcvgan_amd import trainer
    from dcvgan_amd.configs import CONFIGS
    from oracle import stepcheck as SC
    from tests.test_stepcheck_cpu import _Fp32Runner
    worst = {"update_rel_l2": 0.0, "loss_rel": 0.0, "buffers_rel": 0.0, "kink_far": 0.0, "kink_frac": 0.0, "sens_off_frac": 0.0, "worst_over_lr": 0.0,
             "moment_rel": 0.0, "moment2_rel": 0.0}
    for name in a.configs:
        for seed in range(a.seed0, a.seed0 + a.seeds):
            cfg = CONFIGS[name].scaled(batchsize=a.batch)
            torch.manual_seed(1000 + seed)
            models = trainer.build_models(cfg, torch.device("cpu"))
            g = torch.Generator().manual_seed(2000 + seed)
            lo, hi = (-0.5, 0.5) if cfg.channel == 2 else (-1.0, 1.0)
            xc = torch.rand(a.batch, 3, 16, 64, 64, generator=g) * 2 - 1
            xg = torch.rand(a.batch, cfg.channel, 16, 64, 64, generator=g) * (hi - lo) + lo
            states = {n: {k: v.detach().clone() for k, v in m.state_dict().items()} for n, m in models.items()}
            torch.manual_seed(3000 + seed)
            run = _Fp32Runner(cfg, states)
            forced = SC.ForcedStepOracle(cfg, run.so.rng.log)
            ts = [(3 + 5 * i + seed) % 16 for i in range(a.iters)]
            for it, t in enumerate(ts, 1):
                res = SC.checked_iteration(run, run.models, run.opts, forced, run, xc, xg, xc, xg, t, cfg.lr)
                rows = res["rows"]
                w, wk = SC.worst(rows, "rel_l2")
                ns, no = sum(x["n_sensitive"] for x in rows), sum(x["n_sensitive_off"] for x in rows)
                per_row = max((x["n_sensitive_off"] / x["n_sensitive"]) for x in rows if x["n_sensitive"] >= 80) if any(x["n_sensitive"] >= 80 for x in rows) else 0.0
                wl = max(x["worst_over_lr"] / max(1, x["calls"]) for x in rows)
                m1 = max(x.get("moment_rel", 0.0) for x in rows); m2 = max(x.get("moment2_rel", 0.0) for x in rows)
                kw = res.get("kink_worst_call")
                print(f"{name:15s} seed {seed} it {it}: update {w:.2e} loss {res['loss_rel']:.1e} buffers {res['buffers_rel']:.1e} kinks {res['kink_flips']}/{res['kink_total']} far {res['kink_far']:.1e} "
                      f"sensitive {ns} off {no} (worst tensor {per_row:.3f}) worst/lr/call {wl:.2f} moments {m1:.1e} {m2:.1e} | worst update {wk[0]}/{wk[1]} | furthest kink: call {kw[0] if kw else -1} "
                      f"shape {kw[4] if kw else ()} at {kw[5] if kw else None}", flush=True)
                worst["update_rel_l2"] = max(worst["update_rel_l2"], w); worst["loss_rel"] = max(worst["loss_rel"], res["loss_rel"])
                worst["buffers_rel"] = max(worst["buffers_rel"], res["buffers_rel"]); worst["kink_far"] = max(worst["kink_far"], res["kink_far"])
                worst["kink_frac"] = max(worst["kink_frac"], res["kink_flips"] / max(1, res["kink_total"])); worst["sens_off_frac"] = max(worst["sens_off_frac"], per_row)
                worst["worst_over_lr"] = max(worst["worst_over_lr"], wl); worst["moment_rel"] = max(worst["moment_rel"], m1); worst["moment2_rel"] = max(worst["moment2_rel"], m2)
    print("maxima over %d runs: %s" % (len(a.configs) * a.seeds * a.iters, {k: float("%.3g" % v) for k, v in worst.items()}))


if __name__ == "__main__":
    main()
