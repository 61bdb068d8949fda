#!/bin/bash
set -u
OUT=gpurun_out/r3i; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
for T in _ab/309a120 _ab/33fa93d .; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$(basename $T) -o k --output-format csv -- python $R/$T/bench.py --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap > $R/$OUT/prof_$(basename $T).log 2>&1)
  echo "== $T"; grep -E "k_torso_finish|k_cond_encode|k_frame_init|k_head_phase" $OUT/prof_$(basename $T)/k_kernel_stats.csv | awk -F'","' '{printf "%-60s calls %s avg_ns %s\n", substr($1,2,58), $2, $4}'
done
timeout 300 python tools/trace_head.py --split --json $OUT/trace_split.json > $OUT/trace_split.txt 2>&1; tail -70 $OUT/trace_split.txt
