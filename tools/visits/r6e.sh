#!/bin/bash
# round 6, visit e: the AMP training tier (f16 forward with binary16 saves, half operands in the weight-gradient products) -- tests, then the
# step's rate: fp32 / AMP on the f16 tier / AMP with the exact-fp32 node (round 5), interleaved, and the kernel profile of the AMP step
set -u
OUT=gpurun_out/r6e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_closed_loop.py -m gpu -q 2>&1 | tail -12 > $OUT/pytest.log; tail -6 $OUT/pytest.log
for i in 1 2 3; do
timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32.jsonl | cut -c1-230
timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_f16_tier.jsonl | cut -c1-230
timeout 300 python tools/bench_train.py --amp --amp-f32-field 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_fp32_node.jsonl | cut -c1-230
done
for m in "" "--amp"; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_train$m -o k --output-format csv -- python $OLDPWD/tools/bench_train.py $m --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_train$m.log 2>&1); echo "== profile $m"; head -12 $OUT/prof_train$m/k_kernel_stats.csv | cut -c1-170
done
