#!/bin/bash
# round 3, visit 2: split tier -- probe, tests, bench lines at 2/3/4 frames in flight, kernel profile
set -u
OUT=gpurun_out/r3b; mkdir -p $OUT; export TMPDIR=/tmp
tools/mfma_denorm_probe.bin 2>&1 | tee $OUT/mfma_denorm_probe.txt
timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -q -x -k "split or frame_256 or frame_loop_512 or in_flight or reproducible" 2>&1 | tail -25 > $OUT/pytest_split.log; tail -6 $OUT/pytest_split.log
for IF in 2 3 4; do
  timeout 300 python bench.py --precision split --no-cpu-baseline --no-stress --png-frames 0 --in-flight $IF > $OUT/bench_split_if$IF.json 2>/dev/null; python -c "
import json;d=json.load(open('$OUT/bench_split_if$IF.json'));r=d['roofline'];print('split if=$IF fps %.1f ms %.4f kernel_ms %.4f samples %.0f' % (d['value'], d['ms_per_step'], r['kernel_ms_per_frame'], r['samples_per_frame']))"
done
timeout 600 python bench.py --precision split --cpu-frames 2 --parity-frames 4 > $OUT/bench_split.json 2> $OUT/bench_split.err; cut -c1-400 $OUT/bench_split.json
timeout 300 python bench.py --fast --no-cpu-baseline --no-stress --png-frames 0 > $OUT/bench_fast.json 2>/dev/null; cut -c1-200 $OUT/bench_fast.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_split -o k --output-format csv -- python $OLDPWD/bench.py --precision split --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap > $OLDPWD/$OUT/prof_split.log 2>&1); head -6 $OUT/prof_split/k_kernel_stats.csv | cut -c1-160
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $OUT/pytest.log; tail -6 $OUT/pytest.log
