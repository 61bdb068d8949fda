#!/bin/bash
# round 6, visit w: k_grid_bin with the per-level gradient loads batched -- the scatter as called by the training node, the AMP step, the grid tests
set -u
OUT=gpurun_out/r6w; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2; do timeout 200 python tools/grid_backward_binned.py 2>/dev/null | tail -1 | tee -a $OUT/binned.jsonl | cut -c1-400; done
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "binned or grid or amp_training" 2>&1 | tail -4
for i in 1 2 3; do timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp.jsonl | cut -c1-200; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_train_amp -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --amp --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_train_amp.log 2>&1); grep -E "k_grid_bin|k_grid_backward" $OUT/prof_train_amp/k_kernel_stats.csv | cut -c1-60,200-260
