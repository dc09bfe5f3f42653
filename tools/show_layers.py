#!/usr/bin/env python3
"""print a tools/layer_table.py csv sorted by time per iteration"""
import csv, sys
lines = [l for l in open(sys.argv[1]) if not l.startswith('#')]
rows = [r for r in csv.DictReader(lines) if r.get('ms') and r.get('launches_per_iteration')]
def tot(r):
    try: return float(r['ms']) * float(r['launches_per_iteration'])
    except ValueError: return 0.0
T = 0
for r in sorted(rows, key=lambda r: -tot(r)):
    T += tot(r)
    print("%-34s %-6s %8.1f GF %7.3f ms x%-4s = %6.2f  %6.0f TF  %s" % (r['layer'][:34], r['op'], float(r['gflop'] or 0), float(r['ms']), r['launches_per_iteration'], tot(r), float(r['tflops'] or 0), r['kernel'][:60]))
print("total ms per iteration:", round(T, 2))
