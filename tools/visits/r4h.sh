#!/bin/bash
# round 4, visit h: configs[3] at its per-GPU shape through the entry point on the final tree (a 375-frame shard of 3000 frames, and all 3000 on one GPU)
set -u
OUT=gpurun_out/r4h; mkdir -p $OUT
export TMPDIR=/tmp
for prec in fp32 split; do
  timeout 600 python tools/shard_run.py --precision $prec --json $OUT/shard_375_of_3000_$prec.json > $OUT/shard_$prec.log 2>&1; tail -2 $OUT/shard_$prec.log | cut -c1-300
  timeout 900 python tools/shard_run.py --precision $prec --ranks 1 --rank 0 --files-only --json $OUT/whole_3000_$prec.json > $OUT/whole_$prec.log 2>&1; tail -2 $OUT/whole_$prec.log | cut -c1-300
done
