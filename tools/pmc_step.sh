#!/bin/bash
# Per-kernel PMC summary of TWO bench steps (separate passes, as the MI355X guide prescribes):
#   pass 1 FETCH_SIZE, pass 2 WRITE_SIZE (KB), pass 3 SQ + GRBM.  Output: gpurun_out/pmc_step/summary.json
#   (same layout as profiles/r01_pmc_summary.json, which bench.py reads for roofline.traffic)
#   usage: pmc_step.sh [config [precision [outdir]]]   (defaults: the fp32 headline, gpurun_out/pmc_step)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CFG=${1:-isogd-depth}; PREC=${2:-fp32}; OUT=${3:-gpurun_out/pmc_step}
export PMC_CFG=$CFG PMC_PREC=$PREC PMC_OUT=$OUT
export GIT_HEAD=${GIT_HEAD:-$(git rev-parse HEAD 2>/dev/null || cat .git_head 2>/dev/null || echo unknown)}
mkdir -p $OUT
CMD="python3 bench.py --config $CFG --precision $PREC --steps 2 --warmup 0 --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary"
run() { rm -rf /tmp/pmc_$1; timeout -k 10 400 rocprofv3 --pmc $2 -d /tmp/pmc_$1 -o p --output-format csv -- $CMD > $OUT/$1.log 2>&1 || { tail -3 $OUT/$1.log; exit 1; }; }
run FETCH "FETCH_SIZE" && run WRITE "WRITE_SIZE" && \
run SQ "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" || exit 1
python3 - <<'PY'
import csv, glob, collections, json, re
out = collections.defaultdict(lambda: collections.defaultdict(float))
for tag in ("FETCH", "WRITE", "SQ"):
    f = glob.glob(f"/tmp/pmc_{tag}/**/*counter_collection.csv", recursive=True)[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], tag) not in seen:
            seen.add((r["Dispatch_Id"], tag))
            out[k]["dur_ns_" + tag] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            out[k]["dispatches"] = max(out[k]["dispatches"], 0) + (1 if tag == "SQ" else 0)
import os, sys
sys.path.insert(0, os.getcwd())
from dcvgan_amd import native
from dcvgan_amd.configs import CONFIGS
cfg = os.environ["PMC_CFG"]
out["__meta__"] = {"config": cfg, "precision": os.environ["PMC_PREC"], "batch": CONFIGS[cfg].batchsize, "steps": 2, "csrc_sha256": native.csrc_digest(cl=os.environ["PMC_PREC"] == "bf16cl"), "git_head": os.environ.get("GIT_HEAD")}
json.dump(out, open(os.environ["PMC_OUT"] + "/summary.json", "w"), indent=1)
del out["__meta__"]
kb = sum(v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0) for v in out.values())
print("HBM GB per step (FETCH+WRITE, raw):", kb * 1024 / 2 / 1e9)
gui = max(v.get("GRBM_GUI_ACTIVE", 0) for v in out.values())
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("dur_ns_SQ", 0))[:8]:
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0); act = v.get("GRBM_GUI_ACTIVE", 0)
    print("%-60s %7.2f ms/step  mfma busy %.3f" % (k[:60], v.get("dur_ns_SQ", 0) / 2e6, busy / (act / 8 * 1024) if act else 0))
PY
