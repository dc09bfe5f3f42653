#!/bin/bash
# SQ counters of one layer's kernels: tools/pmc_layer.sh <filter>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf /tmp/pmc1; mkdir -p gpurun_out/pmc
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d /tmp/pmc1 -o p --output-format csv -- python3 tools/one_layer.py "$1" > gpurun_out/pmc/run.log 2>&1 || { tail -5 gpurun_out/pmc/run.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc1/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
out = open("gpurun_out/pmc/summary.txt", "w")
for k, c in agg.items():
    if "dcv::" not in k or c.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    wc = c["SQ_WAVE_CYCLES"]
    print(k, "launches", cnt[k], file=out)
    for n in sorted(c): print("   %-28s %14.0f  /WAVE_CYCLES %.3f" % (n, c[n], c[n] / wc), file=out)
out.close()
PY
cat gpurun_out/pmc/summary.txt
