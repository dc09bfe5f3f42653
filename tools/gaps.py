"""Idle-gap analysis of a rocprofv3 kernel trace: per step, sum of kernel durations vs span."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# last 40% of the trace = timed steps
n = len(ev); ev = ev[int(n * 0.5):]
span = ev[-1][1] - ev[0][0]
busy = 0; cur_end = ev[0][0]; gaps = []
for s, e, k in ev:
    if s > cur_end: gaps.append((s - cur_end, k))
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
print("kernels", len(ev), "span ms", span / 1e6, "busy ms", busy / 1e6, "idle ms", (span - busy) / 1e6, "idle frac", 1 - busy / span)
import collections
g = sorted(gaps, reverse=True)
print("largest gaps (us):", [(round(x / 1e3, 1), k[:40]) for x, k in g[:12]])
h = collections.Counter()
for x, _ in gaps: h[min(int(x / 1e3), 50)] += 1
print("gap histogram (us: count):", sorted(h.items()))
by = collections.defaultdict(lambda: [0, 0])
for x, k in gaps:
    by[k[:60]][0] += x; by[k[:60]][1] += 1
print("idle before kernel (top):")
for k, (x, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]: print("  %8.2f ms  n=%5d avg %6.1f us  %s" % (x / 1e6, c, x / c / 1e3, k))
# where in time the idle sits: 2 ms windows over the analysed span, idle ms and the kernels that followed the gaps
W = 2_000_000
win = collections.defaultdict(float); wk = collections.defaultdict(collections.Counter)
cur_end = ev[0][0]
for s, e, k in ev:
    if s > cur_end:
        w = (s - ev[0][0]) // W
        win[w] += s - cur_end; wk[w][k[:28]] += s - cur_end
    cur_end = max(cur_end, e)
print("idle per 2 ms window (ms; windows with > 0.15 ms idle):")
for w in sorted(win):
    if win[w] > 150_000: print("  t=%6.1f ms  idle %.2f  %s" % (w * W / 1e6, win[w] / 1e6, [(k, round(v / 1e3)) for k, v in wk[w].most_common(3)]))
