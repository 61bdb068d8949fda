#!/bin/bash
# round 6, visit m: the table scatter level by level (tools/grid_backward_levels.py), product library and a 20 000-entry variant (7 instead of 8 row
# partitions on the 65 536-row levels)
set -u
OUT=gpurun_out/r6m; mkdir -p $OUT
for rep in 1 2; do
timeout 300 python tools/grid_backward_levels.py 2>/dev/null | tail -1 | tee -a $OUT/levels_base.jsonl | cut -c1-900
GF_HIP_LIB=$PWD/geneface_amd/csrc/libgeneface_hip_gb20k.so timeout 300 python tools/grid_backward_levels.py 2>/dev/null | tail -1 | tee -a $OUT/levels_gb20k.jsonl | cut -c1-900
done
