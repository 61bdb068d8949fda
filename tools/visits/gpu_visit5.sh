#!/bin/bash
set -u
OUT=gpurun_out/r3e; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
for V in "" _small64 _small0 _slp; do
  GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip$V.so timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train lib=%-9s ms/step %.2f' % ('${V:-base}', d['ms_per_step']))" | tee -a $OUT/train_ab.txt
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_train -o k --output-format csv -- python $R/tools/bench_train.py --steps 32 --warmup 16 > $R/$OUT/prof_train.log 2>&1); head -12 $OUT/prof_train/k_kernel_stats.csv | cut -c1-150
GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip_small0.so bash -c "cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_train_small0 -o k --output-format csv -- python $R/tools/bench_train.py --steps 32 --warmup 16 > $R/$OUT/prof_train_small0.log 2>&1"; head -8 $OUT/prof_train_small0/k_kernel_stats.csv | cut -c1-150
