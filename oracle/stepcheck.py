"""Checker for ONE composed training iteration (trainer.py:279-363 + train.py:171-176).  TEST INFRASTRUCTURE ONLY — imported by
``tests/`` and ``__graft_entry__.smoke()``, never by ``dcvgan_amd/``.

Comparing the parameters after an iteration with another fp32 run of the same iteration is a lottery, twice over:
(a) a (Leaky)ReLU pre-activation within rounding of zero takes the other branch in one arithmetic and moves every upstream gradient
by ~1e-3; (b) Adam's first moves are ``-lr * g' / (|g'| + eps)`` ~ ``-lr * sign(g')`` (g' = g + weight_decay * theta), so an element
whose g' is within the gradient's rounding error of zero moves by +lr in one run and -lr in the other.  ``ForcedStepOracle`` removes
both so that the update itself can be held tight:

  * it evaluates the pinned oracle's iteration in **fp64**, on the draws the checked run consumed (``ReplayRng64``);
  * every (Leaky)ReLU differentiates with the branch pattern the checked run took (``KinkTape``; the pattern itself is bounded
    separately: it may differ from fp64's own signs only within ``KINK_EPS`` rms of zero, for a handful of elements);
  * before every iteration it is **teacher-forced**: parameters, BatchNorm buffers and Adam moments / step counts are overwritten
    with the checked run's current values, so iteration k compares one iteration, not the accumulated lottery of k of them;
  * the fp32 rounding of the stored parameter (<= ulp(theta) / 2 per optimiser call — 3e-4 of a move of lr for a BatchNorm gamma ~ 1;
    the reference's fp32 parameters carry the same) is taken off the difference first;
  * ``compare`` splits each tensor's elements into *insensitive* ones — sqrt(v_hat) >= TAU * rms(sqrt(v_hat)); at Adam's first
    step sqrt(v_hat) = |g'| — whose update is a smooth function of the gradient and must agree in relative L2, and *sensitive*
    ones (|g'| within TAU of zero relative to the tensor), which are counted, may each be off by at most the largest move Adam can
    make (2 * lr per optimiser call), and of which only a bounded number may disagree at all.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import dcvgan_oracle as O

MODELS = ("ggen", "cgen", "idis", "vdis", "gdis")
TAU = 1e-2            # sensitive: sqrt(v_hat) < TAU * rms over the tensor
KINK_EPS = 1e-2       # |pre-activation| / rms(layer) of an element whose recorded branch differs from fp64's own sign.  Wider than the 5e-5 of the single-pass gradient tests: after an Adam step some BatchNorm channels are ill-conditioned (|mean| / std ~ 1e3: fp32 rounding of x is 1e-4 of the normalised value); measured 1.2e-4 on HIP (gdis.main.6, surreal-depth1 iteration 2) and 9.6e-5 for the reference's own fp32 CPU arithmetic (gdis.main.10, isogd-depth iteration 1; tools/stepcheck_reference.py)
KINK_FRAC = 5e-6      # such elements / all activation elements


class ReplayRng64(O.ReplayRng):
    """The recorded fp32 draws, widened: the fp64 graph sees exactly the random numbers the checked run consumed."""

    def _next(self, kind, shape):
        return super()._next(kind, shape).double()


def snapshot(models, opts) -> Dict[str, dict]:
    """CPU copy of everything an iteration reads and writes, from objects with the reference's duck types (``state_dict()``,
    ``parameters()``; the optimisers' ``.params`` / ``.state`` as in torch.optim / dcvgan_amd.optim)."""
    out = {}
    for n in MODELS:
        sd = {k: v.detach().cpu().clone() for k, v in models[n].state_dict().items()}
        o = getattr(opts[n], "inner", opts[n])
        adam = []
        for p in o.params:
            s = o.state.get(p)
            adam.append(None if not s else {"step": int(s["step"]), "exp_avg": s["exp_avg"].detach().cpu().clone(),
                                           "exp_avg_sq": s["exp_avg_sq"].detach().cpu().clone()})
        out[n] = {"state": sd, "adam": adam, "training": bool(models[n].training)}
    return out


class ForcedStepOracle:
    def __init__(self, cfg, draw_log, betas=(0.5, 0.999)):
        self.cfg, self.betas = cfg, betas
        self.rng = ReplayRng64(draw_log)
        self.so = None

    def _build(self, snap):
        st = {n: {k: (v.double().clone() if v.dtype.is_floating_point else v.clone()) for k, v in snap[n]["state"].items()} for n in MODELS}
        self.so = O.StepOracle(self.cfg, st, self.rng)

    def force(self, snap, iteration_done: int):
        """Overwrite the oracle's parameters, buffers, Adam state, iteration counter and generator train/eval mode with `snap`."""
        if self.so is None:
            self._build(snap)
        so = self.so
        so.iteration = iteration_done
        so.gen_training = snap["ggen"]["training"]
        with torch.no_grad():
            for n in MODELS:
                for k, v in so.st[n].items():
                    v.copy_(snap[n]["state"][k].to(v.dtype))
                opt = so.opt[n]
                for p, s in zip(O.trainable(so.st[n]), snap[n]["adam"]):
                    if s is None:
                        opt.state.pop(p, None)
                    else:
                        opt.state[p] = {"step": torch.tensor(float(s["step"])), "exp_avg": s["exp_avg"].double().clone(),
                                        "exp_avg_sq": s["exp_avg_sq"].double().clone()}
                    p.grad = None

    def step(self, xc_real, xg_real, t_rand: int, kinks) -> dict:
        """One iteration from the forced state with the recorded branch patterns.  -> losses + kink statistics."""
        with O.KinkTape(kinks) as tape:
            losses = self.so.step(xc_real.double(), xg_real.double(), t_rand)
        assert tape.pos == len(kinks), (tape.pos, len(kinks))
        flips = sum(m[0] for m in tape.mismatch); total = sum(m[1] for m in tape.mismatch)
        far = max([m[2] for m in tape.mismatch] or [0.0])
        worst_call = max(range(len(tape.mismatch)), key=lambda i: tape.mismatch[i][2]) if tape.mismatch else -1
        return {"losses": losses, "kink_flips": flips, "kink_total": total, "kink_far": far,
                "kink_worst_call": (worst_call,) + tuple(tape.mismatch[worst_call]) if worst_call >= 0 else None}

    def compare(self, before, after, lrs) -> List[dict]:
        """Per parameter tensor: the checked run's update (after - before, both snapshots) against this oracle's from the same
        `before`.  Rows: model, key, numel, calls (optimiser steps this iteration), rel_l2 over the insensitive elements,
        n_sensitive, n_sensitive_off (sensitive elements whose updates differ by more than 1 % of lr), worst_over_lr."""
        rows = []
        b1, b2 = self.betas
        for n in MODELS:
            opt = self.so.opt[n]
            keys = [k for k in self.so.st[n] if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
            for k, p, s0, s1 in zip(keys, O.trainable(self.so.st[n]), before[n]["adam"], after[n]["adam"]):
                t0, t1 = before[n]["state"][k].double(), after[n]["state"][k].double()
                d_hip, d_ref = t1 - t0, p.detach() - t0
                calls = (s1["step"] if s1 else 0) - (s0["step"] if s0 else 0)
                ref_state = opt.state.get(p)
                ref_calls = (int(ref_state["step"]) if ref_state else 0) - (s0["step"] if s0 else 0)
                row = {"model": n, "key": k, "numel": t0.numel(), "calls": calls, "ref_calls": ref_calls}
                if ref_calls == 0 or calls == 0:
                    row.update(rel_l2=0.0 if float(d_hip.abs().max()) == 0.0 and float(d_ref.abs().max()) == 0.0 else float("inf"),
                               n_sensitive=0, n_sensitive_off=0, worst_over_lr=float(d_hip.abs().max()) / lrs[n])
                    rows.append(row)
                    continue
                vhat = ref_state["exp_avg_sq"] / (1.0 - b2 ** float(ref_state["step"]))
                r = vhat.sqrt()
                sens = r < TAU * r.pow(2).mean().sqrt()
                # the checked run stores theta in fp32: every optimiser call rounds theta_new to the fp32 grid, an absolute error of up
                # to ulp(theta)/2 per call that has nothing to do with the update's arithmetic (BatchNorm gammas ~1: ulp 1.2e-7 against
                # moves of lr = 2e-4, i.e. 3e-4 of the update — the reference's fp32 parameters carry exactly the same).  It is taken
                # off before the comparison: what remains is the error of the update itself.
                a32 = after[n]["state"][k].float().abs()
                ulp = (torch.nextafter(a32, torch.full_like(a32, float("inf"))) - a32).double()
                raw = d_hip - d_ref
                diff = (raw.abs() - calls * ulp).clamp_min(0.0)
                ins = ~sens
                den = d_ref[ins].norm().clamp_min(1e-300)
                row.update(rel_l2=float(diff[ins].norm() / den) if int(ins.sum()) else 0.0,
                           n_sensitive=int(sens.sum()),
                           n_sensitive_off=int((diff[sens] > 1e-2 * lrs[n]).sum()),
                           worst_over_lr=float(raw.abs().max()) / lrs[n], rel_l2_raw=float(raw[ins].norm() / den) if int(ins.sum()) else 0.0,
                           # Adam's moments after the iteration over the same insensitive elements: at the first steps the UPDATE is ~lr * sign(g'),
                           # so only these hold the gradient's MAGNITUDE to the oracle's (round 4: asserted, MOMENT_TOL)
                           moment_rel=float((s1["exp_avg"].double() - ref_state["exp_avg"])[ins].norm() / ref_state["exp_avg"][ins].norm().clamp_min(1e-300)) if int(ins.sum()) else 0.0,
                           moment2_rel=float((s1["exp_avg_sq"].double() - ref_state["exp_avg_sq"])[ins].norm() / ref_state["exp_avg_sq"][ins].norm().clamp_min(1e-300)) if int(ins.sum()) else 0.0)
                rows.append(row)
        return rows

    def buffers_rel(self, after) -> float:
        """Largest relative error of a BatchNorm running statistic after the iteration (and num_batches_tracked must be equal)."""
        worst = 0.0
        for n in MODELS:
            for k, v in self.so.st[n].items():
                a = after[n]["state"][k]
                if k.endswith("num_batches_tracked"):
                    assert int(a) == int(v), (n, k, int(a), int(v))
                elif k.endswith(("running_mean", "running_var")):
                    worst = max(worst, float((a.double() - v).norm() / v.norm().clamp_min(1e-30)))
        return worst


def worst(rows, field):
    r = max(rows, key=lambda x: x[field])
    return r[field], (r["model"], r["key"])


def checked_iteration(runner, models, opts, forced: ForcedStepOracle, layers_mod, xc_dev, xg_dev, xc_cpu, xg_cpu, t_rand: int, lrs) -> dict:
    """Run ONE iteration of `runner` (the checked implementation; `layers_mod.KINK_TAP` is its branch-pattern tap) and the same
    iteration of the teacher-forced fp64 oracle; return everything the callers assert on."""
    before = snapshot(models, opts)
    forced.force(before, runner.iteration)
    layers_mod.KINK_TAP = kinks = []
    try:
        got = runner.step(xc_dev, xg_dev, t_rand)
    finally:
        layers_mod.KINK_TAP = None
    got = {k: float(v) for k, v in got.items()}
    after = snapshot(models, opts)
    ref = forced.step(xc_cpu, xg_cpu, t_rand, kinks)
    rows = forced.compare(before, after, lrs)
    loss_rel = max(abs(got[k] - ref["losses"][k]) / max(abs(ref["losses"][k]), 1e-30) for k in ref["losses"])
    return {"losses": got, "ref_losses": ref["losses"], "loss_rel": loss_rel, "rows": rows, "buffers_rel": forced.buffers_rel(after),
            "kink_flips": ref["kink_flips"], "kink_total": ref["kink_total"], "kink_far": ref["kink_far"], "kink_worst_call": ref["kink_worst_call"]}


# The bars every caller uses.  Measured: the fixed-seed tests' reports (update <= 1.6e-5, pattern <= 1.2e-4 rms) and sweeps over 20 seeds x 3 full-width
# configs x 2 iterations at B = 2 plus B = 4 / B = 8 runs (tools/step_seed_sweep.py -> profiles/r03_step_parity/seed_sweep_*.txt, 150 checked iterations,
# 0 failures): update <= 1.2e-4, loss <= 2.5e-6, buffers <= 1.6e-5, pattern <= 6e-7 of the elements and <= 9.1e-4 rms from zero, sensitive elements off
# <= 2.8 % of a tensor's.  The two lottery quantities (update, pattern distance) are heavy-tailed over seeds — an ill-conditioned BatchNorm channel —
# so their bars sit ~10x above the sweeps' maxima rather than the tests' values; a 1 % learning-rate error moves the update by 1e-2 (tests/test_stepcheck_cpu.py).
UPDATE_TOL = 1e-3          # relative L2 of a tensor's update over its insensitive elements
LOSS_TOL = 1e-4            # every loss of the iteration, relative
BUFFER_TOL = 2e-4          # BatchNorm running statistics after the iteration, relative L2 per buffer
MOMENT_TOL = 1e-3          # Adam's exp_avg / exp_avg_sq after the iteration, relative L2 over the insensitive elements (round 4; measured: profiles/r04_step_parity/)


def assert_iteration(res, lrs, tag=""):
    """The tight statements about one checked iteration.  -> (worst rel_l2, sensitive elements, of which off)"""
    assert res["loss_rel"] <= LOSS_TOL, (tag, res["losses"], res["ref_losses"])
    assert res["buffers_rel"] <= BUFFER_TOL, (tag, res["buffers_rel"])
    assert res["kink_flips"] <= max(8, KINK_FRAC * res["kink_total"]) and res["kink_far"] <= KINK_EPS, (tag, res["kink_flips"], res["kink_total"], res["kink_far"], res.get("kink_worst_call"))
    n_sens = n_off = 0
    w = 0.0
    for r in res["rows"]:
        assert r["calls"] == r["ref_calls"], (tag, r)                       # same optimiser schedule (gating, the double ggen step)
        assert r["rel_l2"] <= UPDATE_TOL, (tag, r)
        assert r.get("moment_rel", 0.0) <= MOMENT_TOL and r.get("moment2_rel", 0.0) <= MOMENT_TOL, (tag, r)
        # a sensitive element is at most the largest move Adam can make away, per optimiser call: |m_hat| / sqrt(v_hat) <= 1 / sqrt(1 - beta2)
        # in general, and ~1 (i.e. 2 * lr between +lr and -lr) on the first steps that these tests take
        assert r["worst_over_lr"] <= 2.1 * max(1, r["calls"]), (tag, r)
        # ... and only a few of them disagree at all: an element is within the gradient's own rounding error of zero with probability
        # ~1e-6 / TAU of being sensitive at all; a quarter of the sensitive ones (at least 8) is ~10x the largest share measured (2.8 %) and far below "all of them"
        assert r["n_sensitive_off"] <= max(8, 0.25 * r["n_sensitive"]), (tag, r)
        n_sens += r["n_sensitive"]; n_off += r["n_sensitive_off"]; w = max(w, r["rel_l2"])
    return w, n_sens, n_off


def report_lines(res, it):
    out = ["# iteration %d: losses hip %s" % (it, res["losses"]), "#              fp64 %s" % (res["ref_losses"],),
           "# loss_rel %.3e  buffers_rel %.3e  kinks: %d of %d on the other branch than fp64, furthest %.2e rms from zero" %
           (res["loss_rel"], res["buffers_rel"], res["kink_flips"], res["kink_total"], res["kink_far"]),
           "# furthest: activation call %s" % (res.get("kink_worst_call"),),
           "# %-44s %9s %5s %-11s %-9s %-6s %-9s %-10s %-10s" % ("tensor", "numel", "calls", "rel_l2(ins)", "sensitive", "off", "worst/lr", "moment_rel", "moment2_rel")]
    for r in res["rows"]:
        out.append("%-46s %9d %5d %.3e   %9d %6d %.3e %.3e %.3e" % (r["model"] + "/" + r["key"], r["numel"], r["calls"], r["rel_l2"], r["n_sensitive"],
                                                                  r["n_sensitive_off"], r["worst_over_lr"], r.get("moment_rel", 0.0), r.get("moment2_rel", 0.0)))
    return out
