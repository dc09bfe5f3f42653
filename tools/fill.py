"""How full is the GPU over an iteration?  From a rocprofv3 kernel trace: at every instant the workgroups of the kernels in flight (grid / workgroup size), summed;
time per fill class (idle, < 64 workgroups, < 256, < 1024, more) and the kernels that run alone while under-filled.   usage: fill.py <trace dir>"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    w = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    ev.append((s, e, max(1, g // max(w, 1)), r["Kernel_Name"][:50]))
ev.sort()
n = len(ev); ev = ev[n // 2:]
pts = []
for i, (s, e, wg, k) in enumerate(ev):
    pts.append((s, 1, i)); pts.append((e, -1, i))
pts.sort()
live = set(); last = pts[0][0]
cls = collections.Counter(); alone = collections.Counter()
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        tot = sum(ev[j][2] for j in live)
        c = "idle" if not live else "<64 wgs" if tot < 64 else "<256 wgs" if tot < 256 else "<1024 wgs" if tot < 1024 else ">=1024 wgs"
        cls[c] += dt
        if live and tot < 256:
            for j in live: alone[ev[j][3]] += dt / len(live)
    last = t
    if d == 1: live.add(i)
    else: live.discard(i)
span = sum(cls.values())
print("span %.1f ms" % (span / 1e6))
for c in ("idle", "<64 wgs", "<256 wgs", "<1024 wgs", ">=1024 wgs"): print("  %-12s %7.2f ms  %5.1f %%" % (c, cls[c] / 1e6, 100.0 * cls[c] / span))
print("kernels in flight while fewer than 256 workgroups are (ms):")
for k, v in alone.most_common(14): print("  %7.2f  %s" % (v / 1e6, k))
