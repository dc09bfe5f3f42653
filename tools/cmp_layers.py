"""Compare two layer tables (tools/layer_table.py --csv): rows whose time moved by more than 3 %, and the weighted totals."""
import csv, sys
def load(f):
    lines = [l for l in open(f) if not l.startswith('#')]
    return {(r['layer'], r['op']): r for r in csv.DictReader(lines) if r.get('ms') and r['op'] != 'all'}
a, b = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0.0
for k in a:
    if k not in b: continue
    w = float(a[k]['launches_per_iteration']); x, y = float(a[k]['ms']), float(b[k]['ms'])
    ta += w * x; tb += w * y
    if abs(x - y) > 0.03 * x: print("%-34s %-6s x%-4s %.3f -> %.3f  (%+.2f ms/iter)  %s" % (k[0], k[1], a[k]['launches_per_iteration'], x, y, w * (y - x), b[k]['kernel'][:90]))
print("total %.2f -> %.2f ms per iteration" % (ta, tb))
