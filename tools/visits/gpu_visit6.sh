#!/bin/bash
set -u
OUT=gpurun_out/r3f; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
for V in _small0 "" _slp "" _small64 ""; do
  GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip$V.so timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train lib=%-9s ms/step %.2f points %d' % ('${V:-base}', d['ms_per_step'], d['points_last_step']))" | tee -a $OUT/train_ab.txt
done
timeout 300 python tools/bench_train.py --steps 32 2>/dev/null | tail -1 | cut -c1-200
