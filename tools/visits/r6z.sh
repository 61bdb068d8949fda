#!/bin/bash
# round 6, visit z: the condition encoder as two HIP graphs (RADNeRF.graph_cond_encoder) in the training step -- AMP and fp32, with / without
set -u
OUT=gpurun_out/r6z; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2; do
timeout 300 python tools/bench_train.py --amp --graph-cond 2>$OUT/err_amp_graph.txt | tail -1 | tee -a $OUT/bench_train_amp_graph_cond.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py --amp 2>/dev/null | tail -1 | tee -a $OUT/bench_train_amp.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py --graph-cond 2>$OUT/err_fp32_graph.txt | tail -1 | tee -a $OUT/bench_train_fp32_graph_cond.jsonl | cut -c1-200
timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | tee -a $OUT/bench_train_fp32.jsonl | cut -c1-200
done
tail -5 $OUT/err_amp_graph.txt | cut -c1-300
