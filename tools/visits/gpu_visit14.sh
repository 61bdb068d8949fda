#!/bin/bash
set -u
OUT=gpurun_out/r3n; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
for rep in 1 2; do
for V in "" _serialwalk _batch8 _t512 _t256 _t512b8; do
  GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip$V.so timeout 300 python bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-stress --png-frames 0 --profile-frames 16 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('lib=%-15s init_ms %.4f' % ('${V:-base}', r['marcher']['ms']))" | tee -a $OUT/init_ab2.txt
done
done
