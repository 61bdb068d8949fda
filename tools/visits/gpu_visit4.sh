#!/bin/bash
set -u
OUT=gpurun_out/r3d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 600 python tools/png_bench.py $OUT/png_bench.json 2>/dev/null | cut -c1-250
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));r=d['roofline'];print('fp32 fps %.1f ms %.4f frac %.4f kernel_ms %.4f png %s' % (d['value'], d['ms_per_step'], r['frac'], r['kernel_ms_per_frame'], d['with_png'])); print(d['parity']); print('heavy', d['heavy_fixture']['value'], d['heavy_fixture']['samples_per_frame'], d['heavy_fixture']['roofline_frac'])"
timeout 600 python bench.py --precision split > $OUT/bench_split.json 2> $OUT/bench_split.err; python -c "
import json;d=json.load(open('$OUT/bench_split.json'));r=d['roofline'];print('split fps %.1f ms %.4f kernel_ms %.4f png %s' % (d['value'], d['ms_per_step'], r['kernel_ms_per_frame'], d['with_png'])); print(d['parity']); print(d['stress_fixture']['parity']); print('if', d['config']['frames_in_flight'])"
timeout 300 python bench.py --precision split --no-cpu-baseline --no-stress --png-frames 0 --in-flight 4 2>/dev/null | cut -c1-120
timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 > $OUT/bench_train.json; cut -c1-300 $OUT/bench_train.json
