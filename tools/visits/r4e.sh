#!/bin/bash
# round 4, visit e: skinny layers from the accumulators -- full GPU suite, same-box A/B against -DGF_SKINNY_FROM_LDS (fp32 and split), timelines
set -u
OUT=gpurun_out/r4e; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q -rfE -s > $OUT/pytest_full.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_full.log | cut -c1-300
for prec in fp32 split; do
  for rep in 1 2 3; do
    for lib in "" "_skinnylds"; do
      GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip$lib.so timeout 300 python bench.py --precision $prec --steps 60 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('AB prec=$prec lib=%-8s fps=%.1f kernel_ms=%.4f frac=%s' % ('${lib:-registers}', d['value'], r['kernel_ms_per_frame'], r.get('frac')))" | tee -a $OUT/ab_skinny_from_registers.txt
    done
  done
done
timeout 300 python tools/trace_head.py --json $OUT/trace_fp32.json > $OUT/trace_fp32.txt 2>&1; grep -E "phase ms|encode|rows|store H|round =" $OUT/trace_fp32.txt | cut -c1-200
timeout 300 python tools/trace_head.py --split --json $OUT/trace_split.json > $OUT/trace_split.txt 2>&1; grep -E "phase ms|encode|rows|store H|round =" $OUT/trace_split.txt | cut -c1-200
