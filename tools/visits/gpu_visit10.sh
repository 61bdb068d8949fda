#!/bin/bash
set -u
OUT=gpurun_out/r3j; mkdir -p $OUT; export TMPDIR=/tmp
for P in fp32 split; do
  timeout 600 python tools/shard_run.py --precision $P --json $OUT/shard_375_of_3000_$P.json 2> $OUT/shard_$P.err | cut -c1-400
  timeout 900 python tools/shard_run.py --precision $P --ranks 1 --rank 0 --json $OUT/whole_3000_$P.json 2>> $OUT/shard_$P.err | cut -c1-400
done
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));print('fp32 fps %.1f host_enq_ms %s split %s' % (d['value'], d['host_enqueue_ms_per_step'], {k:d['split_tier'][k] for k in ('value','host_enqueue_ms_per_step','frames_in_flight')})); print(d['split_tier']['parity'])"
timeout 300 python bench.py --fast --no-cpu-baseline --no-stress --png-frames 0 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());print('fast fps %.1f host_enq_ms %s' % (d['value'], d['host_enqueue_ms_per_step']))"
