#!/bin/bash
# round 6, visit d: (1) the new closed-loop torso stage + the touched suites; (2) the INSTRUCTION MIX of k_head_phase<0/1/2> from the SQ counters
# (VALU / SALU / LDS / VMEM / SMEM / MFMA wave-instructions per launch, busy cycles) -- the input of tools/issue_roofline.py;
# (3) one bench line per tier with the per-frame round counts the mix is normalised by; (4) the torso step again (one compaction per step).
set -u
OUT=gpurun_out/r6d; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_grid_update.py tests/test_gpu_train.py -m gpu -q 2>&1 | tail -12 > $OUT/pytest.log; tail -5 $OUT/pytest.log
(cd /tmp && rocprofv3 --list-avail > $REPO/$OUT/counters_avail.txt 2>&1); grep -c "SQ_" $OUT/counters_avail.txt; grep -o "SQ_INSTS_[A-Z0-9_]*" $OUT/counters_avail.txt | sort -u | tr '\n' ' '
pass() { # tier-tag, bench flags, pass name, counters...
  local tag=$1 flags=$2 name=$3; shift 3
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" -d $REPO/$OUT/pmci${tag}_$name -o c --output-format csv -- python $REPO/bench.py $flags --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 0 --no-overlap > $REPO/$OUT/pmci${tag}_$name.log 2>&1) || echo "pass $tag $name failed"
}
for tier in "fp32:" "split:--precision split" "fast:--fast"; do
  tag=${tier%%:*}; flags=${tier#*:}
  pass $tag "$flags" valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
  pass $tag "$flags" mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT
  pass $tag "$flags" mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass $tag "$flags" act SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY
  python tools/pmc_summary.py $OUT pmci$tag > $OUT/pmci${tag}_summary.txt 2>&1; grep -A40 "== k_head_phase" $OUT/pmci${tag}_summary.txt | grep -E "==|SQ_INSTS|BUSY|ACTIVE|WAVES|WAVE_CYCLES|GRBM|duration|dispatches" | head -40
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-stress --png-frames 0 --no-cpu-baseline --full-line > $OUT/bench_fp32_short.json 2>/dev/null; cut -c1-200 $OUT/bench_fp32_short.json
timeout 300 python bench.py --precision split --steps 20 --warmup 5 --no-stress --png-frames 0 --no-cpu-baseline --full-line > $OUT/bench_split_short.json 2>/dev/null; cut -c1-200 $OUT/bench_split_short.json
timeout 300 python bench.py --fast --steps 20 --warmup 5 --no-stress --png-frames 0 --no-cpu-baseline --full-line > $OUT/bench_fast_short.json 2>/dev/null; cut -c1-200 $OUT/bench_fast_short.json
for i in 1 2 3; do timeout 300 python tools/bench_train.py --torso 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_fused.jsonl | cut -c1-200; done
