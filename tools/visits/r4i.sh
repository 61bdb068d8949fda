#!/bin/bash
# round 4, visit i: LDS poisoning of the final kernels (diag build): every LDS word a workgroup has not written reads as NaN -- the new skinny-layer
# partials live in columns / pad words of the activation rows; a frame that changes under poisoning would prove an uninitialised read
set -u
OUT=gpurun_out/r4i; mkdir -p $OUT
export TMPDIR=/tmp
for prec in fp32 split fast; do
  timeout 600 python tools/fast_diag.py --precision $prec --size 256 --frames 40 --steps poison,group > $OUT/diag_$prec.log 2>&1; tail -12 $OUT/diag_$prec.log | cut -c1-220
done
