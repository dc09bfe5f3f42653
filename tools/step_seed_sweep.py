"""How much margin do the composed-iteration bars have?  The step tests run fixed seeds; this sweeps seeds: for each (config, seed) it builds the five
models at full width, draws a real batch, runs ITERS iterations of trainer.StepRunner on the HIP path and checks each with oracle/stepcheck.py (teacher-forced
fp64 oracle, the run's own activation pattern).  Prints one line per (config, seed, iteration) and the maxima beside the bars.
    python3 tools/step_seed_sweep.py [--seeds 8] [--iters 2] [--batch 2] [--configs isogd-depth surreal-depth1 isogd-flow]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--seed0", type=int, default=0, help="first seed (sweeps with different --seed0 do not repeat one another)")
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--configs", nargs="*", default=["isogd-depth", "surreal-depth1", "isogd-flow"])
    a = ap.parse_args()
    from dcvgan_amd import layers, native, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import InjectedRng
    from oracle import dcvgan_oracle as O
    from oracle import stepcheck as SC
    native.lib()
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    worst = {"update_rel_l2": 0.0, "loss_rel": 0.0, "buffers_rel": 0.0, "kink_far": 0.0, "kink_frac": 0.0, "sens_off_frac": 0.0, "worst_over_lr": 0.0}
    fails = 0
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
            so = O.StepOracle(cfg, states)                      # the reference's arithmetic: supplies the draws
            ts = [(3 + 5 * i + seed) % 16 for i in range(a.iters)]
            for t in ts:
                so.step(xc, xg, t)
            for m in models.values():
                m.to(dev)
                for sub in m.modules():
                    if hasattr(sub, "device"):
                        sub.device = dev
            r = InjectedRng(so.rng.log)
            for m in models.values():
                m._rng = r
            opts = trainer.build_optimizers(cfg, models)
            runner = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=True)
            forced = SC.ForcedStepOracle(cfg, so.rng.log)
            xc_d, xg_d = xc.to(dev), xg.to(dev)
            for it, t in enumerate(ts, 1):
                res = SC.checked_iteration(runner, models, opts, forced, layers, xc_d, xg_d, xc, xg, t, cfg.lr)
                rows = res["rows"]
                w = max(x["rel_l2"] for x in rows)
                ns, no = sum(x["n_sensitive"] for x in rows), sum(x["n_sensitive_off"] for x in rows)
                per_row = max((x["n_sensitive_off"] / x["n_sensitive"]) for x in rows if x["n_sensitive"] >= 80) if any(x["n_sensitive"] >= 80 for x in rows) else 0.0
                wl = max(x["worst_over_lr"] / max(1, x["calls"]) for x in rows)
                ok = True
                try:
                    SC.assert_iteration(res, cfg.lr, f"{name} seed {seed} it {it}")
                except AssertionError as e:
                    ok = False; fails += 1
                    print("  FAIL", str(e)[:300])
                m1 = max(x.get("moment_rel", 0.0) for x in rows); m2 = max(x.get("moment2_rel", 0.0) for x in rows)
                wk = SC.worst(rows, "rel_l2")[1]
                kw = res.get("kink_worst_call")
                print(f"{name:15s} seed {seed} it {it}: update {w:.2e} loss {res['loss_rel']:.1e} buffers {res['buffers_rel']:.1e} kinks {res['kink_flips']}/{res['kink_total']} far {res['kink_far']:.1e} "
                      f"sensitive {ns} off {no} (worst tensor {per_row:.3f}) worst/lr/call {wl:.2f} moments {m1:.1e} {m2:.1e} {'ok' if ok else 'FAIL'} | worst update {wk[0]}/{wk[1]} | "
                      f"furthest kink: call {kw[0] if kw else -1} shape {kw[4] if kw else ()} at {kw[5] if kw else None}", flush=True)
                worst["moment_rel"] = max(worst.get("moment_rel", 0.0), m1); worst["moment2_rel"] = max(worst.get("moment2_rel", 0.0), m2)
                worst["update_rel_l2"] = max(worst["update_rel_l2"], w); worst["loss_rel"] = max(worst["loss_rel"], res["loss_rel"])
                worst["buffers_rel"] = max(worst["buffers_rel"], res["buffers_rel"]); worst["kink_far"] = max(worst["kink_far"], res["kink_far"])
                worst["kink_frac"] = max(worst["kink_frac"], res["kink_flips"] / max(1, res["kink_total"])); worst["sens_off_frac"] = max(worst["sens_off_frac"], per_row)
                worst["worst_over_lr"] = max(worst["worst_over_lr"], wl)
            assert r.pos == len(so.rng.log)
            del models, opts, runner, forced
            torch.cuda.empty_cache()
    print("maxima over %d runs: %s" % (len(a.configs) * a.seeds * a.iters, {k: float("%.3g" % v) for k, v in worst.items()}))
    print("bars: update %.0e  loss %.0e  buffers %.0e  kink_far %.0e  kink_frac %.0e  sensitive-off per tensor 0.25 (or 8)  worst/lr/call 2.1   failures: %d" %
          (SC.UPDATE_TOL, SC.LOSS_TOL, SC.BUFFER_TOL, SC.KINK_EPS, SC.KINK_FRAC, fails))


if __name__ == "__main__":
    main()
