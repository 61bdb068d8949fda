#!/bin/bash
# round 6, visit 7h: the torso task's step with the field on every sampled pixel (no compaction, no host sync) against the compacted form
set -u
OUT=gpurun_out/r7h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_closed_loop.py -m gpu -q -x -k "torso" 2>&1 | tail -5
for i in 1 2 3; do
timeout 300 python tools/bench_train.py --torso 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_dense.jsonl | cut -c1-260
timeout 300 python tools/bench_train.py --torso --torso-compact 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_compact.jsonl | cut -c1-260
done
timeout 300 python tests/train_rate_reference.py --torso 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_reference_kernels.jsonl | cut -c1-200
