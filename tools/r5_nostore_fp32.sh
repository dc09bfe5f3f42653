#!/bin/bash
# what do the dominant kernel's stores cost?  ("the dominant kernel writes 1.50x its output ... either try it or close it with a measured A/B")
# The same kernel with EVERY store dropped by the buffer range check (DCV_DEBUG_NOSTORE=1: an empty output descriptor) against the shipped one, alternating, one box;
# then the fp32 layer table of the headline config both ways.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5p; mkdir -p $O
for i in 1 2 3; do
  echo "stores on : $(timeout -k 10 120 python3 tools/probe_dominant.py 70 30 2>&1 | tail -n 1)"
  echo "stores off: $(DCV_DEBUG_NOSTORE=1 timeout -k 10 120 python3 tools/probe_dominant.py 70 30 2>&1 | tail -n 1)"
done
timeout -k 10 300 python3 tools/layer_table.py isogd-depth --csv $O/layers_stores_on.csv > $O/layers_on.txt 2>&1; echo "stores on : $(tail -n 1 $O/layers_on.txt)"
DCV_DEBUG_NOSTORE=1 timeout -k 10 300 python3 tools/layer_table.py isogd-depth --csv $O/layers_stores_off.csv > $O/layers_off.txt 2>&1; echo "stores off: $(tail -n 1 $O/layers_off.txt)"
