#!/bin/bash
set -u
OUT=gpurun_out/r3r; mkdir -p $OUT; export TMPDIR=/tmp
GF_HEAD_GRID=256 timeout 300 python tools/trace_head.py --split > $OUT/trace_split_1wg.txt 2>&1; grep -E "round =|mfma|store H|encode|march|rows" $OUT/trace_split_1wg.txt | head -24
timeout 300 python tools/trace_head.py --split > $OUT/trace_split_2wg.txt 2>&1; grep -E "round =|mfma|store H" $OUT/trace_split_2wg.txt | head -16
