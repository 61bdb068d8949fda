#!/bin/bash
# round 4, visit g: the look-ahead sample cache -- byte identity against -DGF_NO_LOOKAHEAD_CACHE, same-box A/B (fp32 and split), fast tier line
set -u
OUT=gpurun_out/r4g; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python tools/frame_digests.py > $OUT/digests_cache.txt 2> $OUT/digests_cache.err
GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip_nolacache.so timeout 600 python tools/frame_digests.py > $OUT/digests_nocache.txt 2> $OUT/digests_nocache.err
if diff -q $OUT/digests_cache.txt $OUT/digests_nocache.txt > /dev/null; then echo "BYTE-IDENTICAL: $(wc -l < $OUT/digests_cache.txt) frame digests agree"; else echo "DIGESTS DIFFER"; diff $OUT/digests_cache.txt $OUT/digests_nocache.txt | head; fi
for prec in fp32 split; do
  for rep in 1 2 3; do
    for lib in "" "_nolacache"; do
      GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip$lib.so timeout 300 python bench.py --precision $prec --steps 60 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('AB prec=$prec lib=%-10s fps=%.1f kernel_ms=%.4f frac=%s' % ('${lib:-cache}', d['value'], r['kernel_ms_per_frame'], r.get('frac')))" | tee -a $OUT/ab_lookahead_cache.txt
    done
  done
done
timeout 300 python bench.py --fast --no-cpu-baseline --no-stress --png-frames 0 > $OUT/bench_fast.json 2> $OUT/bench_fast.err; python -c "import json; d=json.load(open('$OUT/bench_fast.json')); print('fast tier', d['value'], d['ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_sweep.py -m gpu -q -x 2>&1 | tail -3
