#!/bin/bash
# Split write-back: one packed conversion feeding both fused ops (product) against the compiler's value-by-value form (variant nopk).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3ab; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "split or reproducible or interfere" 2>&1 | tail -3 | tee $OUT/pytest_split.log
for rep in 1 2 3; do
  for lib in "" nopk; do
    L=$REPO/geneface_amd/csrc/libgeneface_hip${lib:+_$lib}.so
    GF_HIP_LIB=$L timeout 300 python bench.py --precision split --steps 100 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split lib=%-8s fps=%.1f kernel_ms=%.4f' % ('${lib:-base}', d['value'], d['roofline']['kernel_ms_per_frame']))" | tee -a $OUT/ab.txt
  done
done
