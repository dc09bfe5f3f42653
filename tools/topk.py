"""Longest dispatches of the last step in a rocprofv3 kernel trace (duration, grid, kernel)."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 8
last = rows[-n:]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print("dispatches in last step", n, "sum ms", tot / 1e6)
agg = collections.defaultdict(lambda: [0, 0])
for r in last:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    key = (r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    agg[key][0] += d; agg[key][1] += 1
for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print("%8.3f ms  n=%2d avg %7.3f  grid %s,%s,%s  %s" % (d / 1e6, c, d / c / 1e6, k[1], k[2], k[3], k[0]))
