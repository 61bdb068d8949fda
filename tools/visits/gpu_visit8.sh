#!/bin/bash
set -u
OUT=gpurun_out/r3h; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
timeout 300 python tools/trace_head.py --split --json $OUT/trace_split.json > $OUT/trace_split.txt 2>&1; cat $OUT/trace_split.txt | tail -60
timeout 300 python tools/trace_head.py --fast > $OUT/trace_fast.txt 2>&1; grep -E "round =|phase ms|lifetime|gather|encode|mfma|store|march|composite|scan|refill" $OUT/trace_fast.txt | head -40
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o k --output-format csv -- python $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap --precision split > $R/$OUT/prof.log 2>&1); head -6 $OUT/prof/k_kernel_stats.csv | cut -c1-150
