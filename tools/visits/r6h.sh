#!/bin/bash
# round 6, visit h: the AMP tier against the reference's own Python under autocast (tests/test_gpu_refpy.py) and as a closed loop against the
# fp32 loop (tests/test_gpu_closed_loop.py)
set -u
OUT=gpurun_out/r6h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_refpy.py -m gpu -q -s -k "autocast" > $OUT/pytest_refpy_amp.log 2>&1; grep -A40 "^AMP step, relative" $OUT/pytest_refpy_amp.log | head -50; tail -3 $OUT/pytest_refpy_amp.log
timeout 900 python -m pytest tests/test_gpu_closed_loop.py -m gpu -q -s -k "amp_loop" 2>&1 | tail -15 > $OUT/pytest_closed_loop_amp.log; tail -4 $OUT/pytest_closed_loop_amp.log
cp gpurun_out/closed_loop_amp_loss_curves.json $OUT/ 2>/dev/null
