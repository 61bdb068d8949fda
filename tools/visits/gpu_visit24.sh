#!/bin/bash
# Scheduler-strategy A/B (build variants with -mllvm -amdgpu-sched-strategy=...): fp32 and split lines, interleaved on one box.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3aa; mkdir -p $OUT
for rep in 1 2; do
  for lib in "" maxilp itilp; do
    L=$REPO/geneface_amd/csrc/libgeneface_hip${lib:+_$lib}.so
    GF_HIP_LIB=$L timeout 300 python bench.py --steps 60 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32  lib=%-8s fps=%.1f kernel_ms=%.4f' % ('${lib:-base}', d['value'], d['roofline']['kernel_ms_per_frame']))" | tee -a $OUT/ab.txt
  done
  for lib in "" maxilp; do
    L=$REPO/geneface_amd/csrc/libgeneface_hip${lib:+_$lib}.so
    GF_HIP_LIB=$L timeout 300 python bench.py --precision split --steps 60 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split lib=%-8s fps=%.1f kernel_ms=%.4f' % ('${lib:-base}', d['value'], d['roofline']['kernel_ms_per_frame']))" | tee -a $OUT/ab.txt
  done
done
