#!/bin/bash
# Split tier: the three skinny layers on the matrix pipe (product) against the VALU rows (variant valurows).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3ae; mkdir -p $OUT
timeout 300 python tools/trace_head.py --split > $OUT/trace_split.txt 2>&1; grep -E "rows|sigma row|SH \+|round =" $OUT/trace_split.txt | head -12
for lib in "" valurows; do
  L=$REPO/geneface_amd/csrc/libgeneface_hip${lib:+_$lib}.so
  GF_HIP_LIB=$L timeout 300 python bench.py --precision split --steps 100 --warmup 10 --no-stress --png-frames 0 --parity-frames 2 --cpu-frames 1 --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split lib=%-8s fps=%.1f kernel_ms=%.4f max_abs=%.3g psnr=%.2f' % ('${lib:-base}', d['value'], d['roofline']['kernel_ms_per_frame'], d['parity']['max_abs_rgb'], d['parity']['psnr_db']))" | tee -a $OUT/ab.txt
done
for rep in 1 2 3; do
  for lib in "" valurows; do
    L=$REPO/geneface_amd/csrc/libgeneface_hip${lib:+_$lib}.so
    GF_HIP_LIB=$L timeout 300 python bench.py --precision split --steps 100 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split lib=%-8s fps=%.1f kernel_ms=%.4f' % ('${lib:-base}', d['value'], d['roofline']['kernel_ms_per_frame']))" | tee -a $OUT/ab.txt
  done
done
