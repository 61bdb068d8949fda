#!/bin/bash
# round 6, visit p: the TRAINING step on the final tree under rocprofv3 -- kernel stats and three PMC passes (HBM reads, HBM writes + L2, SQ busy /
# matrix-pipe busy) for the fp32 step, the AMP step and the torso step ("a fresh rocprofv3 + PMC summary of the training step")
set -u
OUT=gpurun_out/r7p; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
for mode in "fp32:" "amp:--amp" "torso:--torso"; do
  tag=${mode%%:*}; flags=${mode#*:}
  timeout 300 python tools/bench_train.py $flags 2>/dev/null | tail -1 > $OUT/bench_train_$tag.json; cut -c1-240 $OUT/bench_train_$tag.json
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_$tag -o k --output-format csv -- python $REPO/tools/bench_train.py $flags --steps 32 --warmup 16 > $REPO/$OUT/prof_$tag.log 2>&1); head -5 $OUT/prof_$tag/k_kernel_stats.csv | cut -c1-150
  pass() { local name=$1; shift
    (cd /tmp && timeout 600 rocprofv3 --pmc "$@" -d $REPO/$OUT/pmct${tag}_$name -o c --output-format csv -- python $REPO/tools/bench_train.py $flags --steps 4 --warmup 17 > $REPO/$OUT/pmct${tag}_$name.log 2>&1) || echo "pass $tag $name failed"
  }
  pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
  pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  pass sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
  python tools/pmc_summary.py $OUT pmct$tag train > $OUT/pmct${tag}_summary.txt 2>&1; tail -3 $OUT/pmct${tag}_summary.txt | cut -c1-160
done
ls $OUT | head -40
