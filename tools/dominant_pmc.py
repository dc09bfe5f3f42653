"""PMC passes of tools/probe_dominant.py (see tools/round_profile.sh) -> per-launch HBM traffic and MFMA utilisation of the
dominant kernel.  FETCH_SIZE / WRITE_SIZE are in KiB... rocprofv3 reports them in KB (1024 B) per dispatch; FETCH_SIZE is
doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for 16-byte-per-lane streams on gfx950 — both operand
streams of this kernel are 16-byte LDS-DMA granules (patch staging + packed weights)."""
import csv
import glob
import json
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcvgan_amd import native

B, N = 70, 10
KERNEL = [None]


def per_dispatch(tag, names):
    f = glob.glob(f"/tmp/pd_{tag}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gather_gemm_dma_kernel" in r["Kernel_Name"]]
    m = re.search(r"gather_gemm_dma_kernel<[^>]*>", rows[-1]["Kernel_Name"])
    KERNEL[0] = m.group(0) if m else rows[-1]["Kernel_Name"]
    out = {}
    for n in names:
        vals = {}
        for r in rows:
            if r["Counter_Name"] == n:
                vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        v = [vals[k] for k in sorted(vals, key=int)][-N:]      # the timed launches
        out[n] = sum(v) / len(v)
    return out


fe = per_dispatch("FETCH", ["FETCH_SIZE"])["FETCH_SIZE"] * 1024
wr = per_dispatch("WRITE", ["WRITE_SIZE"])["WRITE_SIZE"] * 1024
sq = per_dispatch("SQ", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"])
F = B * 16
algo = (F * 128 * 32 * 32 + F * 64 * 64 * 64 + 128 * 64 * 16) * 4
print(json.dumps({
    "kernel": KERNEL[0], "git_head": os.environ.get("GIT_HEAD") or (open(".git_head").read().strip() if os.path.exists(".git_head") else None), "csrc_sha256": native.csrc_digest(), "layer": "cgen.up_blocks.5 forward", "batch": B,
    "fetch_size_bytes_raw": fe, "fetch_size_bytes_corrected_x2": 2 * fe, "write_size_bytes": wr,
    "hbm_bytes_per_launch": 2 * fe + wr, "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": (2 * fe + wr) / algo,
    # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD... normalised as in tools/pmc_step.sh: busy / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
    "mfma_busy": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8 * 1024) if sq["GRBM_GUI_ACTIVE"] else None,
    "lds_bank_conflict_over_wave_cycles": sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_WAVE_CYCLES"] if sq["SQ_WAVE_CYCLES"] else None,
    "counters_per_launch": sq}, indent=1))
