#!/bin/bash
set -u
OUT=gpurun_out/r3l; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
for V in "" _serialwalk _nowalk _nostore _nowalknostore _batch8 _batch2; do
  GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip$V.so timeout 300 python bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-stress --png-frames 0 --profile-frames 16 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('lib=%-15s init_ms %.4f hit %.0f' % ('${V:-base}', r['marcher']['ms'], r['marcher']['hit_rays']))" | tee -a $OUT/init_ab.txt
done
