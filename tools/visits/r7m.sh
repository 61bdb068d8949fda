#!/bin/bash
# round 6, visit x: the records of the FINAL kernel sources -- PMC (traffic, busy cycles) and rocprofv3 kernel stats of the fp32 frame loop, the SQ
# instruction mix of k_head_phase<0/1/2> (input of tools/issue_roofline.py), short bench lines of the three tiers
set -u
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r7m_pytest.log; tail -3 gpurun_out/r7m_pytest.log
OUT=gpurun_out/r7m; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
bash tools/gpu_round.sh r7m pmc prof > $OUT/gpu_round.log 2>&1; tail -3 $OUT/gpu_round.log | cut -c1-200
pass() { local tag=$1 flags=$2 name=$3; shift 3
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" -d $REPO/$OUT/pmci${tag}_$name -o c --output-format csv -- python $REPO/bench.py $flags --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 0 --no-overlap > $REPO/$OUT/pmci${tag}_$name.log 2>&1) || echo "pass $tag $name failed"
}
for tier in "fp32:" "split:--precision split" "fast:--fast"; do
  tag=${tier%%:*}; flags=${tier#*:}
  pass $tag "$flags" valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
  pass $tag "$flags" mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT
  pass $tag "$flags" mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass $tag "$flags" act SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY
  python tools/pmc_summary.py $OUT pmci$tag > $OUT/pmci${tag}_summary.txt 2>&1; tail -2 $OUT/pmci${tag}_summary.txt | cut -c1-200
done
timeout 300 python bench.py --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 3 --no-overlap --full-line > $OUT/bench_counter_pass_frames.json 2>/dev/null; python -c "
import json;d=json.loads(open('$OUT/bench_counter_pass_frames.json').read());print('samples_per_frame', d['roofline']['samples_per_frame'])"
ls $OUT | head -40
