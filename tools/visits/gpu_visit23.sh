#!/bin/bash
set -u
OUT=gpurun_out/r3w; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
for rep in 1 2 3; do
  (cd $R/_ab/fc725b3 && timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress --png-frames 0 --profile-frames 8 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('tree=fc725b3 lib=base   fp32 fps %.1f' % (d['value']))") | tee -a $OUT/static_fill_ab.txt
for V in "" _dyn; do
  for P in fp32 split; do
  GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip$V.so timeout 300 python bench.py --precision $P --steps 60 --warmup 10 --no-cpu-baseline --no-stress --png-frames 0 --profile-frames 8 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('tree=HEAD    lib=%-6s %-5s fps %.1f kernel_ms %.4f' % ('${V:-static}', '$P', d['value'], r['kernel_ms_per_frame']))" | tee -a $OUT/static_fill_ab.txt
  done
done; done
