#!/bin/bash
set -u
OUT=gpurun_out/r3m; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_entry_point.py -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest_render.log; tail -3 $OUT/pytest_render.log
for P in fp32 split fast; do
timeout 300 python bench.py --precision $P --no-cpu-baseline --no-stress --png-frames 0 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('$P fps %.1f ms %.4f init_ms %.4f kernel_ms %.4f' % (d['value'], d['ms_per_step'], r['marcher']['ms'], r['kernel_ms_per_frame']))" | tee -a $OUT/bench_after_init_fix.txt
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o k --output-format csv -- python $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap > $R/$OUT/prof.log 2>&1); head -7 $OUT/prof/k_kernel_stats.csv | cut -c1-150
