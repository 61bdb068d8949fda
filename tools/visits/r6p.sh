#!/bin/bash
# round 6, visit p: the binning pass in front of the table scatter -- tests (bit comparison with the unbinned scatter), the training tests, then the
# AMP / fp32 steps with and without it (GF_GRID_BINNING=0), and the AMP step's kernel profile
set -u
OUT=gpurun_out/r6p; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -12 > $OUT/pytest.log; tail -6 $OUT/pytest.log
for i in 1 2 3; do
timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_binned.jsonl | cut -c1-200
GF_GRID_BINNING=0 timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_unbinned.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32_binned.jsonl | cut -c1-200
GF_GRID_BINNING=0 timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32_unbinned.jsonl | cut -c1-200
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_train_amp -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --amp --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_train_amp.log 2>&1); head -12 $OUT/prof_train_amp/k_kernel_stats.csv | cut -c1-150
