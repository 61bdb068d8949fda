#!/bin/bash
# round 6, visit c: the fused torso training field -- tests against the op graph and the oracle, then the torso step's rate before / after
# (bench_train --torso with the field pinned to the op graph = the tree before this change) and its kernel profile
set -u
OUT=gpurun_out/r6c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_train.log; tail -6 $OUT/pytest_train.log
for i in 1 2 3; do
timeout 300 python tools/bench_train.py --torso 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_fused.jsonl | cut -c1-260
timeout 300 python tools/bench_train.py --torso --op-graph 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_op_graph.jsonl | cut -c1-260
done
timeout 300 python tests/train_rate_reference.py --torso 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_reference_kernels.jsonl | cut -c1-260
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_train_torso -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --torso --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_train_torso.log 2>&1); head -14 $OUT/prof_train_torso/k_kernel_stats.csv | cut -c1-150
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6c/prof_train_torso/k_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('torso step: kernel ms per step', tot/48/1e6, 'launches per step', calls/48)
PY
