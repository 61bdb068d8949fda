#!/bin/bash
set -u
OUT=gpurun_out/r3k; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_entry_point.py -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest_render.log; tail -4 $OUT/pytest_render.log
for V in "" _serialwalk "" _serialwalk; do
  GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip$V.so timeout 300 python bench.py --precision split --no-cpu-baseline --no-stress --png-frames 0 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('split lib=%-12s fps %.1f init_ms %.4f' % ('${V:-base}', d['value'], r['marcher']['ms']))"
done
for V in "" _serialwalk; do
  GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip$V.so timeout 300 python bench.py --no-cpu-baseline --no-stress --png-frames 0 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('fp32 lib=%-12s fps %.1f init_ms %.4f' % ('${V:-base}', d['value'], r['marcher']['ms']))"
done
