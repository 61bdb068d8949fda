#!/bin/bash
# round 6, visit 7i: the torso field's weight gradients in two launches (gf_torso_wgrad) -- tests, the torso step with / without
set -u
OUT=gpurun_out/r7i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_closed_loop.py -m gpu -q -x -k "torso" 2>&1 | tail -5
for i in 1 2 3; do
timeout 300 python tools/bench_train.py --torso 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_fused_dw.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py --torso --torso-gemm-wgrad 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_library_dw.jsonl | cut -c1-200
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_torso -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --torso --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_torso.log 2>&1); grep -E "k_torso" $OUT/prof_torso/k_kernel_stats.csv | cut -c1-130
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r7i/prof_torso/k_kernel_stats.csv')))
print('kernels ms/step', sum(int(r['TotalDurationNs']) for r in rows)/48/1e6, 'launches/step', sum(int(r['Calls']) for r in rows)/48)
P
