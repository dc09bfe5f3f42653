"""rocprofv3 --kernel-trace CSV of a bench.py run -> average duration of the dominant kernel's probe launches (the last 10
dispatches of the kernel bench.py names), beside the HIP-event figure bench.py printed.  Usage: dominant_from_trace.py trace.csv bench.log"""
import csv
import json
import re
import sys

line = [l for l in open(sys.argv[2]) if l.startswith('{"metric"')][-1]
b = json.loads(line)
k = b["roofline"].get("dominant_kernel") or b["roofline"]["kernel"]      # round 4 moved the kernel figures under roofline.dominant_kernel
want = re.sub(r"\s*\(.*", "", k["kernel"]).replace(" ", "")          # gather_gemm_dma_kernel<2,2,1,4,false,true>
rows = [r for r in csv.DictReader(open(sys.argv[1])) if want in r["Kernel_Name"].replace(" ", "")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
probe = rows[-10:]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in probe]
avg = sum(dur) / len(dur)
print(json.dumps({"kernel": k["kernel"], "layer": k["layer"], "dispatches_of_this_instance_in_run": len(rows), "probe_launches": len(probe),
                  "rocprof_avg_ms": avg, "rocprof_min_ms": min(dur), "rocprof_max_ms": max(dur), "hip_event_avg_ms": k["ms"],
                  "gflop_per_launch": k["gflop_per_launch"], "tflops_from_rocprof": k["gflop_per_launch"] / avg,
                  "frac_of_fp32_mfma_peak": k["gflop_per_launch"] / avg / 157.3}, indent=1))
