#!/bin/bash
# encode8 rewritten (one index expression for hashed and dense levels, explicit issue / consume batches) against the previous tree's library
# (_ab/old, built from `git archive` of the commit before): all three tiers, interleaved on one box; parity tests of the new one first.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3af; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vs_ref_kernels.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
for rep in 1 2 3; do
  for prec in fp32 split fast; do
    for lib in new old; do
      L=$REPO/geneface_amd/csrc/libgeneface_hip.so; [ $lib = old ] && L=$REPO/_ab/old/geneface_amd/csrc/libgeneface_hip.so
      GF_HIP_LIB=$L timeout 300 python bench.py --precision $prec --steps 100 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec lib=$lib fps=%.1f kernel_ms=%.4f' % (d['value'], d['roofline']['kernel_ms_per_frame']))" | tee -a $OUT/ab.txt
    done
  done
done
