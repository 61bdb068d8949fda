#!/bin/bash
# round 6, visit r: slices per list in the binned scatter (flush budget, minimum slices, workgroups per level): tools/grid_backward_binned.py on the
# tunable A/B library
set -u
OUT=gpurun_out/r6r; mkdir -p $OUT
L=$PWD/geneface_amd/csrc/libgeneface_hip_gbtune.so
run() { env "$@" GF_HIP_LIB=$L timeout 200 python tools/grid_backward_binned.py 2>/dev/null | tail -1 | tee -a $OUT/tune.jsonl | cut -c1-400; }
timeout 200 python tools/grid_backward_binned.py 2>/dev/null | tail -1 | tee -a $OUT/tune.jsonl | cut -c1-400
run GF_GB_DUMMY=1
run GF_GB_FLUSH_LOG2=21
run GF_GB_FLUSH_LOG2=20
run GF_GB_FLUSH_LOG2=20 GF_GB_MIN_SLICES=4
run GF_GB_FLUSH_LOG2=19 GF_GB_MIN_SLICES=2
run GF_GB_WGS=256
run GF_GB_WGS=256 GF_GB_FLUSH_LOG2=23
run GF_GB_WGS=64 GF_GB_FLUSH_LOG2=21
