#!/bin/bash
set -u
OUT=gpurun_out/r3o; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest.log; tail -3 $OUT/pytest.log
for P in fp32 split fast; do
timeout 300 python bench.py --precision $P --no-cpu-baseline --no-stress --png-frames 0 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('$P fps %.1f ms %.4f init_ms %.4f kernel_ms %.4f frac %s' % (d['value'], d['ms_per_step'], r['marcher']['ms'], r['kernel_ms_per_frame'], r.get('frac')))" | tee -a $OUT/bench_after.txt
done
