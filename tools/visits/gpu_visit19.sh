#!/bin/bash
set -u
OUT=gpurun_out/r3s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -q -x -k "split or in_flight or reproducible or properties" 2>&1 | tail -4 > $OUT/pytest_split.log; tail -2 $OUT/pytest_split.log
for P in split fp32; do
timeout 300 python bench.py --precision $P --no-cpu-baseline --no-stress --png-frames 0 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('$P fps %.1f ms %.4f kernel_ms %.4f phases %s' % (d['value'], d['ms_per_step'], r['kernel_ms_per_frame'], r['example_frame']['phase_ms']))" | tee -a $OUT/bench_after.txt
done
timeout 300 python tools/trace_head.py --split > $OUT/trace_split.txt 2>&1; grep -E "round =|mfma|store H|-> layer" $OUT/trace_split.txt | head -32
