#!/bin/bash
set -u
OUT=gpurun_out/r3u; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
for rep in 1 2 3; do
for T in _ab/6d07113 _ab/fc725b3 .; do
  (cd $R/$T && timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress --png-frames 0 --profile-frames 8 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('tree=%-12s fp32 fps %.1f avg_launch_ms %.4f frac %.4f' % ('$(basename $T)', d['value'], r['avg_launch_ms'], r['frac']))") | tee -a $OUT/fp32_ab.txt
done; done
