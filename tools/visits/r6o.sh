#!/bin/bash
# round 6, visit o: the table scatter with the loads of 4 (product) / 1 (the loop before) / 8 points per lane in flight together
set -u
OUT=gpurun_out/r6o; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for v in "" _gbb1 _gbb8; do
GF_HIP_LIB=$PWD/geneface_amd/csrc/libgeneface_hip$v.so timeout 300 python tools/grid_backward_levels.py 2>/dev/null | tail -1 | tee -a $OUT/levels${v:-_product}.jsonl | cut -c1-700
done
done
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py tests/test_gpu_vs_ref_kernels.py -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest.log; tail -4 $OUT/pytest.log
