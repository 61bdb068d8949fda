#!/bin/bash
set -u
OUT=gpurun_out/r3p; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
for rep in 1 2 3; do
for V in "" _xcd; do
for P in fp32 split; do
  GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip$V.so timeout 300 python bench.py --precision $P --steps 60 --warmup 10 --no-cpu-baseline --no-stress --png-frames 0 --profile-frames 4 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('lib=%-6s %-5s fps %.1f kernel_ms %.4f' % ('${V:-base}', '$P', d['value'], r['kernel_ms_per_frame']))" | tee -a $OUT/xcd_ab.txt
done; done; done
