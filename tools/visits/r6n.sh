#!/bin/bash
# round 6, visit n: the table scatter with compacted hits (gb_points_compact) against the same library without it (-DGF_GB_COMPACT_PARTS=99), level by
# level and whole; the grid / training tests; the AMP and fp32 steps
set -u
OUT=gpurun_out/r6n; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py tests/test_gpu_vs_ref_kernels.py -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest.log; tail -4 $OUT/pytest.log
for rep in 1 2; do
timeout 300 python tools/grid_backward_levels.py 2>/dev/null | tail -1 | tee -a $OUT/levels_compact.jsonl | cut -c1-1000
GF_HIP_LIB=$PWD/geneface_amd/csrc/libgeneface_hip_nocompact.so timeout 300 python tools/grid_backward_levels.py 2>/dev/null | tail -1 | tee -a $OUT/levels_nocompact.jsonl | cut -c1-1000
done
for i in 1 2; do
timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_compact.jsonl | cut -c1-200
GF_HIP_LIB=$PWD/geneface_amd/csrc/libgeneface_hip_nocompact.so timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp_nocompact.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32_compact.jsonl | cut -c1-200
done
