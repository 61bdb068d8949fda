#!/bin/bash
# round 6, visit s: what the save stream costs k_field_points16 / k_field_backward16 -- a probe library without the [M,128] row stores (timing only)
set -u
OUT=gpurun_out/r6s; mkdir -p $OUT
export TMPDIR=/tmp
for v in "" _nosaves; do
(cd /tmp && GF_HIP_LIB=$OLDPWD/geneface_amd/csrc/libgeneface_hip$v.so timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof$v -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --amp --steps 32 --warmup 16 > $OLDPWD/$OUT/prof$v.log 2>&1); grep -E "k_field_points16|k_field_backward16|k_field_wgrad16" $OUT/prof$v/k_kernel_stats.csv | cut -c1-140
done
