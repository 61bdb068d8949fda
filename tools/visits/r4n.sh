#!/bin/bash
# round 4, visit n: the fast tier's skinny layers from the accumulators -- its tests, same-box A/B against -DGF_SKINNY_FROM_LDS
set -u
OUT=gpurun_out/r4n; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -q -k "fast or in_flight or reproducible" 2>&1 | tail -3
for rep in 1 2 3; do
  for lib in "" "_skinnylds"; do
    GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip$lib.so timeout 300 python bench.py --fast --steps 60 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('AB prec=fast lib=%-10s fps=%.1f kernel_ms=%.4f' % ('${lib:-registers}', d['value'], r['kernel_ms_per_frame']))" | tee -a $OUT/ab_fast_skinny.txt
  done
done
timeout 600 python tools/fast_diag.py --precision fast --size 256 --frames 200 --plain --steps stress 2>&1 | tail -4 | cut -c1-200
