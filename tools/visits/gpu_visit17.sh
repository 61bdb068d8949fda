#!/bin/bash
set -u
OUT=gpurun_out/r3q; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -q -x 2>&1 | tail -4 > $OUT/pytest_render.log; tail -2 $OUT/pytest_render.log
for V in "" _td1 _td2 _td3 _td4; do
  (cd /tmp && GF_HIP_LIB=$R/geneface_amd/csrc/libgeneface_hip$V.so timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof$V -o k --output-format csv -- python $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap --precision split > $R/$OUT/prof$V.log 2>&1)
  echo "lib=${V:-base} $(grep k_torso_finish $OUT/prof$V/k_kernel_stats.csv | awk -F'","' '{print "torso avg_ns", $4}') $(grep 'k_head_phase' $OUT/prof$V/k_kernel_stats.csv | awk -F'","' '{print "head avg_ns", $4}')" | tee -a $OUT/torso_diag.txt
done
for P in fp32 split; do
timeout 300 python bench.py --precision $P --no-cpu-baseline --no-stress --png-frames 0 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());r=d['roofline'];print('$P fps %.1f ms %.4f kernel_ms %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms_per_frame']))" | tee -a $OUT/bench_after.txt
done
