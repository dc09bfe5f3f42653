"""Per-kernel table of a tools/pmc_step.sh summary (per iteration: ms, launches, fetch / write GB, TB/s, MFMA busy, parked / issue-stalled waves, LDS conflicts).
usage: pmc_top.py <summary.json> [label]"""
import json, sys
d = json.load(open(sys.argv[1]))
meta = d.pop("__meta__", {})
steps = float(meta.get("steps", 2))
print("# rocprofv3 --pmc (separate passes: FETCH_SIZE; WRITE_SIZE; SQ_* + GRBM_GUI_ACTIVE) -- python3 bench.py --config %s --precision %s --steps %d --warmup 0 ...  (tools/pmc_step.sh)" % (meta.get("config"), meta.get("precision"), steps))
print("# %s: %s" % (sys.argv[2] if len(sys.argv) > 2 else "", meta))
print("# per iteration; fetch = FETCH_SIZE x 2 (16-byte-per-lane streams, MI355X guide), GB; TB/s = (fetch x 2 + write) / kernel time; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024); LDS conflicts as a FRACTION: SQ_LDS_BANK_CONFLICT / SQ_WAVE_CYCLES (as profiles/r0*_dominant_kernel_pmc.json)")
tf = tw = tt = 0.0
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("dur_ns_SQ", 0)):
    ms = v.get("dur_ns_SQ", 0) / steps / 1e6
    f = v.get("FETCH_SIZE", 0) * 1024 * 2 / steps / 1e9; w = v.get("WRITE_SIZE", 0) * 1024 / steps / 1e9
    tf += f; tw += w; tt += ms
    if ms < 0.15: continue
    act = v.get("GRBM_GUI_ACTIVE", 0); wc = v.get("SQ_WAVE_CYCLES", 0); busy = v.get("SQ_BUSY_CYCLES", 0)
    print("%-52s %6.2f ms  launches %5.1f  fetch %6.2f GB  write %6.2f GB  %5.2f TB/s  mfma busy %.3f  waves parked %.2f  issue-stalled %.2f  LDS bank-conflict cycles / wave cycles %.4f" % (
        k[:52], ms, v.get("dispatches", 0) / steps, f, w, (f + w) / ms if ms else 0, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (act / 8 * 1024) if act else 0,
        v.get("SQ_WAIT_ANY", 0) / wc if wc else 0, v.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0, v.get("SQ_LDS_BANK_CONFLICT", 0) / wc if wc else 0))
print("# whole iteration: fetch x 2 = %.1f GB, write = %.1f GB, kernel time (one kernel at a time under the counters) %.1f ms" % (tf, tw, tt))
