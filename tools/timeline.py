"""Where an iteration's wall time goes that is NOT under an MFMA-bound kernel: from a rocprofv3 --kernel-trace CSV, over the last `--iters` iterations
(an iteration = one `adam_multi_kernel` burst of the G phase ... the next), the union of the intervals in which at least one matrix-pipe kernel
(gather_gemm*, wgrad_dma*, wgrad_gemm*) is running, its complement, and the kernels that occupy the complement (the exposed HBM-bound / small work).

    python tools/timeline.py <trace dir> [--iters 3]"""
import argparse
import collections
import csv
import glob

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
f = glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
MFMA = ("gather_gemm", "wgrad_dma", "wgrad_gemm")


def short(k):
    k = k.replace("void ", "").replace("dcv::", "")
    return k.split("(")[0][:56]


# iteration boundaries: starts of adam bursts separated by > 20 ms
adam = [s for s, e, k, q in ev if "adam_multi" in k]
marks = []
for s in adam:
    if not marks or s - marks[-1] > 15_000_000:
        marks.append(s)
# a G-phase Adam burst follows a D-phase one within the iteration: keep every second mark by spacing
# (robust form: use the periodicity — take marks whose distance to the previous kept one exceeds 60 % of the median period of alternate marks)
per = sorted(marks[i + 2] - marks[i] for i in range(len(marks) - 2))
period = per[len(per) // 2] if per else 0
kept = []
for m in marks:
    if not kept or m - kept[-1] > 0.6 * period:
        kept.append(m)
kept = kept[-(a.iters + 1):]
t0, t1 = kept[0], kept[-1]
n_it = len(kept) - 1
sel = [(max(s, t0), min(e, t1), k, q) for s, e, k, q in ev if e > t0 and s < t1]
print(f"{n_it} iterations, {(t1 - t0) / 1e6 / n_it:.2f} ms each (under the tracer)")


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


mf = union([(s, e) for s, e, k, q in sel if any(m in k for m in MFMA)])
anyk = union([(s, e) for s, e, k, q in sel])
tm = sum(e - s for s, e in mf); ta = sum(e - s for s, e in anyk)
print(f"per iteration: under an MFMA-bound kernel {tm / 1e6 / n_it:.2f} ms, under other kernels only {(ta - tm) / 1e6 / n_it:.2f} ms, idle {((t1 - t0) - ta) / 1e6 / n_it:.2f} ms")
# complement of mf inside [t0, t1]
comp = []
cur = t0
for s, e in mf:
    if s > cur:
        comp.append((cur, s))
    cur = max(cur, e)
if cur < t1:
    comp.append((cur, t1))
# attribute: for every non-MFMA kernel, its overlap with the complement; several concurrent ones share (each counted in full, then normalised)
occ = collections.defaultdict(float)
ci = 0
others = sorted((s, e, k) for s, e, k, q in sel if not any(m in k for m in MFMA))
import bisect
starts = [c[0] for c in comp]
for s, e, k in others:
    i = max(0, bisect.bisect_right(starts, s) - 1)
    while i < len(comp) and comp[i][0] < e:
        o = min(e, comp[i][1]) - max(s, comp[i][0])
        if o > 0:
            occ[short(k)] += o
        i += 1
tot = sum(occ.values())
print("kernels occupying the time with no MFMA-bound kernel running (ms per iteration, overlaps among themselves counted in full):")
for k, v in sorted(occ.items(), key=lambda kv: -kv[1])[:24]:
    print(f"  {v / 1e6 / n_it:7.3f}  {k}")
# MFMA kernel time summed vs union: how much they overlap each other
sm = sum(e - s for s, e, k, q in sel if any(m in k for m in MFMA))
print(f"sum of MFMA-bound kernel durations {sm / 1e6 / n_it:.2f} ms per iteration (union {tm / 1e6 / n_it:.2f}: concurrency {sm / tm:.2f}x)")
