#!/bin/bash
# round 6, visit 7a: the condition encoder under training as one autograd node (gf_cond_train_forward / _backward) -- tests, the steps with / without
set -u
OUT=gpurun_out/r7a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "condition_encoder or amp_training or render_training" 2>&1 | tail -12 > $OUT/pytest.log; tail -6 $OUT/pytest.log
for i in 1 2 3; do
timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_cond_node.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py --amp --cond-ops 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_cond_torch.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32_cond_node.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py --cond-ops 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32_cond_torch.jsonl | cut -c1-200
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_train_amp -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --amp --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_train_amp.log 2>&1); grep -E "k_cond_train" $OUT/prof_train_amp/k_kernel_stats.csv | cut -c1-120
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r7a/prof_train_amp/k_kernel_stats.csv')))
print('kernels ms/step', sum(int(r['TotalDurationNs']) for r in rows)/48/1e6, 'launches/step', sum(int(r['Calls']) for r in rows)/48)
P
