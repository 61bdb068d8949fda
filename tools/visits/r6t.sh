#!/bin/bash
# round 6, visit t: the AMP tier's saved rows and gradient rows leave LDS with coalesced stores (rows16_to_global) -- tests, AMP step, kernel profile
set -u
OUT=gpurun_out/r6t; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -12 > $OUT/pytest.log; tail -4 $OUT/pytest.log
for i in 1 2 3; do timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp.jsonl | cut -c1-200; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_train_amp -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --amp --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_train_amp.log 2>&1); head -8 $OUT/prof_train_amp/k_kernel_stats.csv | cut -c1-150
