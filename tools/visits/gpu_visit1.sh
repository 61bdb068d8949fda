#!/bin/bash
# round 3, visit 1: full GPU suite, bench line (+ A/B of the batched cond encoder), the 375-of-3000 shard through the entry point
set -u
OUT=gpurun_out/r3a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 300 python bench.py --no-prepare --no-cpu-baseline --no-stress --png-frames 0 > $OUT/bench_noprep.json 2>/dev/null; cut -c1-300 $OUT/bench_noprep.json
timeout 300 python bench.py --no-cpu-baseline --no-stress --png-frames 0 --in-flight 3 > $OUT/bench_if3.json 2>/dev/null; cut -c1-300 $OUT/bench_if3.json
timeout 600 python tools/shard_run.py --json $OUT/shard_375_of_3000.json 2> $OUT/shard.err | cut -c1-900; tail -3 $OUT/shard.err
