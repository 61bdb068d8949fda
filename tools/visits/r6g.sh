#!/bin/bash
# round 6, visit g: the fused weight-gradient kernel (gf_field_wgrad16: one launch + a fixed-order reduction instead of ten batched library
# products) -- tests, then fp32 / AMP (fused dW) / AMP (library dW) interleaved, and the AMP step's kernel profile
set -u
OUT=gpurun_out/r6g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q 2>&1 | tail -12 > $OUT/pytest.log; tail -6 $OUT/pytest.log
for i in 1 2 3; do
timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32.jsonl | cut -c1-230
timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_fused_dw.jsonl | cut -c1-230
timeout 300 python tools/bench_train.py --amp --amp-gemm-wgrad 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_library_dw.jsonl | cut -c1-230
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_train_amp -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --amp --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_train_amp.log 2>&1); head -16 $OUT/prof_train_amp/k_kernel_stats.csv | cut -c1-170
