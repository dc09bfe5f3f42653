#!/bin/bash
# LDS-DMA rate microbenchmark + idle-gap analysis of the bf16cl iteration
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5s; mkdir -p $O
timeout -k 10 120 ./tools/lds_dma_rate.bin > $O/lds_dma_rate.txt 2>&1 || { tail -5 $O/lds_dma_rate.txt; exit 1; }
cat $O/lds_dma_rate.txt
rm -rf /tmp/kt; timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt -o t --output-format csv -- python3 bench.py --config surreal-depth1 --precision bf16cl --steps 6 --warmup 4 --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary > $O/trace_bench.log 2>&1 || { tail -5 $O/trace_bench.log; exit 1; }
python3 tools/gaps.py /tmp/kt > $O/gaps.txt 2>&1; cat $O/gaps.txt
