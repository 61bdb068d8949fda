#!/bin/bash
# round 6, visit y: on the FINAL tree, with the final PMC / instruction-mix records committed -- the driver's command, the split and fast lines, and a
# 300-frame soak of both strict tiers against the reference's own kernels
set -u
OUT=gpurun_out/r7n; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo rc=$?; cp gpurun_out/bench_details.json $OUT/
cut -c1-400 $OUT/bench_driver_command.json
timeout 900 python bench.py --precision split --no-cpu-baseline --no-train > $OUT/bench_split.json 2> $OUT/bench_split.err; cut -c1-200 $OUT/bench_split.json
timeout 900 python bench.py --fast --no-cpu-baseline --no-train > $OUT/bench_fast.json 2> $OUT/bench_fast.err; cut -c1-200 $OUT/bench_fast.json
timeout 900 python tools/parity_hunt.py --T 300 --frames "" --out $OUT/soak_300_frames_vs_reference_kernels_fp32.json > $OUT/soak_fp32.log 2>&1; tail -2 $OUT/soak_fp32.log | cut -c1-300
timeout 900 python tools/parity_hunt.py --T 300 --frames "" --precision split --out $OUT/soak_300_frames_vs_reference_kernels_split.json > $OUT/soak_split.log 2>&1; tail -2 $OUT/soak_split.log | cut -c1-300
