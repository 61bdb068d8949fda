#!/bin/bash
# One GPU-box visit: parity tests, bench line, head-kernel timeline, rocprofv3 kernel stats, PMC passes.
# usage: tools/gpu_round.sh <tag> [steps...]   (steps: test bench benchfast benchsplit trace ttrace prof pmc pmcsplit pk train clock ab:<variant> abm:<v1>,<v2>; default: test bench trace prof pmc)
set -u
TAG=${1:-r1}; shift || true
STEPS=${*:-test bench trace prof pmc}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for s in $STEPS; do
  case $s in
    test)  timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log ;;
    testv:*) V=${s#testv:}; GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip_$V.so timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_$V.log; tail -3 $OUT/pytest_$V.log ;;
    bench) timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json ;;
    benchfast) timeout 600 python bench.py --fast --no-cpu-baseline > $OUT/bench_fast.json 2> $OUT/bench_fast.err; cat $OUT/bench_fast.json
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_fast -o k --output-format csv -- python $REPO/bench.py --fast --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap > $OUT/prof_fast.log 2>&1); head -6 $OUT/prof_fast/k_kernel_stats.csv | cut -c1-160 ;;
    benchsplit) timeout 600 python bench.py --precision split > $OUT/bench_split.json 2> $OUT/bench_split.err; cut -c1-300 $OUT/bench_split.json
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_split -o k --output-format csv -- python $REPO/bench.py --precision split --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap > $OUT/prof_split.log 2>&1); head -6 $OUT/prof_split/k_kernel_stats.csv | cut -c1-160
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_split_overlap -o k --output-format csv -- python $REPO/bench.py --precision split --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline > $OUT/prof_split_overlap.log 2>&1); head -4 $OUT/prof_split_overlap/k_kernel_stats.csv | cut -c1-160 ;;
    pmcsplit) (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F32 -d $OUT/pmcs_sq -o sq --output-format csv -- python $REPO/bench.py --precision split --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 0 --no-overlap > $OUT/pmcs_sq.log 2>&1)
           (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmcs_fetch -o f --output-format csv -- python $REPO/bench.py --precision split --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 0 --no-overlap > $OUT/pmcs_fetch.log 2>&1)
           (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmcs_write -o w --output-format csv -- python $REPO/bench.py --precision split --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 0 --no-overlap > $OUT/pmcs_write.log 2>&1)
           python tools/pmc_summary.py $OUT pmcs > $OUT/pmc_split_summary.txt 2>&1; tail -5 $OUT/pmc_split_summary.txt ;;
    ab:*)  # A/B of an experiment library against the product one, interleaved, short benches: ab:<variant>
           V=${s#ab:}
           for rep in 1 2 3; do
             for lib in "" "_$V"; do
               GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip$lib.so timeout 300 python bench.py --steps 60 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB lib=%-10s fps=%.1f frac=%.4f kernel_ms=%.4f' % ('${lib:-base}', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_frame']))" | tee -a $OUT/ab_$V.txt
             done
           done ;;
    abm:*) # several experiment libraries against the product one, interleaved: abm:<v1>,<v2>,...
           VS=${s#abm:}
           for rep in 1 2 3; do
             for lib in "" $(echo $VS | tr ',' ' '); do
               L=$REPO/geneface_amd/csrc/libgeneface_hip${lib:+_$lib}.so
               GF_HIP_LIB=$L timeout 300 python bench.py --steps 60 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB lib=%-10s fps=%.1f frac=%.4f kernel_ms=%.4f' % ('${lib:-base}', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_frame']))" | tee -a $OUT/abm.txt
             done
           done ;;
    train) # training tier: step rate (fused Adam), the same with AMP, and the per-kernel profile of the fp32 step
           timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 > $OUT/bench_train.json; cut -c1-200 $OUT/bench_train.json
           timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 > $OUT/bench_train_amp.json; cut -c1-200 $OUT/bench_train_amp.json
           (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_train -o k --output-format csv -- python $REPO/tools/bench_train.py --steps 32 --warmup 16 > $OUT/prof_train.log 2>&1); head -8 $OUT/prof_train/k_kernel_stats.csv | cut -c1-160 ;;
    clock) # shader clock inside the head kernel vs a steady MFMA probe, with the firmware's view (amd-smi)
           timeout 300 bash tools/clock_probe.sh $TAG/clock 8000 2>&1 | tail -30 ;;
    trace) [ -f geneface_amd/csrc/libgeneface_hip_trace.so ] || python -m geneface_amd.csrc.build --trace > /dev/null   # (the instrumented library is not shipped: .gpurunignore)
           timeout 300 python tools/trace_head.py --json $OUT/trace.json > $OUT/trace.txt 2>&1; cat $OUT/trace.txt ;;
    tracev:*) V=${s#tracev:}; GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip_$V.so timeout 300 python tools/trace_head.py > $OUT/trace_$V.txt 2>&1; grep -E "phase ms|lifetime|round =" $OUT/trace_$V.txt ;;
    ttrace) [ -f geneface_amd/csrc/libgeneface_hip_trace.so ] || python -m geneface_amd.csrc.build --trace > /dev/null
           timeout 300 python tools/trace_torso.py > $OUT/trace_torso.txt 2>&1; tail -16 $OUT/trace_torso.txt ;;
    pk)    # packed-FP32 regression (NOTES 9.5): the library built WITH the SLP vectoriser, rendered over and over; the product as control
           python tools/pk_regress.py --build-only > $OUT/slp_census.json 2>/dev/null
           GF_HIP_LIB=$REPO/geneface_amd/csrc/libgeneface_hip_slp.so timeout 600 python tools/pk_regress.py --frames 3000 > $OUT/pk_regress_slp.json 2>/dev/null; cut -c1-400 $OUT/pk_regress_slp.json ;;
    trace1) GF_HEAD_GRID=256 timeout 300 python tools/trace_head.py --json $OUT/trace_1wg.json > $OUT/trace_1wg.txt 2>&1; cat $OUT/trace_1wg.txt ;;
    prof)  # one frame in flight: per-kernel durations are those of the kernel alone (what bench.py's roofline leg times)
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o k --output-format csv -- python $REPO/bench.py --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap > $OUT/prof.log 2>&1); head -8 $OUT/prof/k_kernel_stats.csv | cut -c1-160
           # the default command (three frames in flight: durations include the time a kernel shares the GPU with its neighbour frame)
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_overlap -o k --output-format csv -- python $REPO/bench.py --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline > $OUT/prof_overlap.log 2>&1); head -4 $OUT/prof_overlap/k_kernel_stats.csv | cut -c1-160 ;;
    pmc)   (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F32 -d $OUT/pmc_sq -o sq --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 0 --no-overlap > $OUT/pmc_sq.log 2>&1)
           (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_fetch -o f --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 0 --no-overlap > $OUT/pmc_fetch.log 2>&1)
           (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o w --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 0 --no-overlap > $OUT/pmc_write.log 2>&1)
           python tools/pmc_summary.py $OUT ;;
  esac
done
