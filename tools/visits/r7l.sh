#!/bin/bash
# round 6, visit 7l: the tail of the torso training branch as one node (gf_torso_blend_train_*) -- tests, the torso step with / without
set -u
OUT=gpurun_out/r7l; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_closed_loop.py -m gpu -q -x -k "torso" 2>&1 | tail -5
for i in 1 2 3; do
timeout 300 python tools/bench_train.py --torso 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_fused_blend.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py --torso --torso-blend-ops 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_blend_ops.jsonl | cut -c1-200
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_torso -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --torso --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_torso.log 2>&1)
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r7l/prof_torso/k_kernel_stats.csv')))
print('kernels ms/step', sum(int(r['TotalDurationNs']) for r in rows)/48/1e6, 'launches/step', sum(int(r['Calls']) for r in rows)/48)
P
