#!/bin/bash
# round 4, visit q: pool rays owned by 4 waves x 32 lanes instead of 2 x 64 -- byte identity against -DGF_OWNER_WAVES=2, same-box A/B (fp32, split)
set -u
OUT=gpurun_out/r4q; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python tools/frame_digests.py > $OUT/digests_own4.txt 2> $OUT/digests_own4.err
GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip_own2.so timeout 600 python tools/frame_digests.py > $OUT/digests_own2.txt 2> $OUT/digests_own2.err
if diff -q $OUT/digests_own4.txt $OUT/digests_own2.txt > /dev/null; then echo "BYTE-IDENTICAL: $(wc -l < $OUT/digests_own4.txt) frame digests agree"; else echo "DIGESTS DIFFER"; diff $OUT/digests_own4.txt $OUT/digests_own2.txt | head; tail -3 $OUT/digests_own4.err; fi
for prec in fp32 split; do
  for rep in 1 2 3; do
    for lib in "" "_own2"; do
      GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip$lib.so timeout 300 python bench.py --precision $prec --steps 60 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('AB prec=$prec lib=%-6s fps=%.1f kernel_ms=%.4f frac=%s' % ('${lib:-own4}', d['value'], r['kernel_ms_per_frame'], r.get('frac')))" | tee -a $OUT/ab_owner_waves.txt
    done
  done
done
