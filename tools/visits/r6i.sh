#!/bin/bash
# round 6, visit i: CU-mask reservation for the small kernels, measured: fps and device time per frame of the pipelined
# loop with the persistent head grids kept off R CUs, R = 0 / 8 / 16 / 32 / 64, product library as control; fp32 and split
set -u
OUT=gpurun_out/r6i; mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/geneface_amd/csrc/libgeneface_hip_cumask.so
for rep in 1 2; do
timeout 200 python tools/cumask_ab.py 2>/dev/null | tail -1 | tee -a $OUT/cumask_fp32.jsonl
for R in 0 8 16 32 64; do
GF_HIP_LIB=$L GF_CUMASK_RESERVE=$R timeout 200 python tools/cumask_ab.py 2>/dev/null | tail -1 | tee -a $OUT/cumask_fp32.jsonl
done
done
GF_HIP_LIB=$L GF_CUMASK_RESERVE=16 GF_CUMASK_HIGH=1 timeout 200 python tools/cumask_ab.py 2>/dev/null | tail -1 | tee -a $OUT/cumask_fp32.jsonl
for R in 0 16 32; do
GF_HIP_LIB=$L GF_CUMASK_RESERVE=$R timeout 200 python tools/cumask_ab.py --precision split 2>/dev/null | tail -1 | tee -a $OUT/cumask_split.jsonl
done
