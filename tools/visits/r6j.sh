#!/bin/bash
# round 6, visit j: the exact tier's weight-gradient products in one launch (gf_field_wgrad32, f32 MFMA) -- tests, then the fp32 step with the
# fused kernel / with the library products interleaved, the AMP step beside them, and the fp32 step's kernel profile
set -u
OUT=gpurun_out/r6j; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q 2>&1 | tail -12 > $OUT/pytest.log; tail -6 $OUT/pytest.log
for i in 1 2 3; do
timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32_fused_dw.jsonl | cut -c1-230
timeout 300 python tools/bench_train.py --gemm-wgrad 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32_library_dw.jsonl | cut -c1-230
timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp.jsonl | cut -c1-230
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_train -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_train.log 2>&1); head -12 $OUT/prof_train/k_kernel_stats.csv | cut -c1-170
