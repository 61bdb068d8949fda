#!/bin/bash
# round 6, visit a: the whole GPU suite on the tree with ADVICE r5's fixes; the host-side term of the 8-rank curve (8 processes x the native PNG
# writer, one directory vs per-rank directories, tmpfs and disk) on the GPU box's host cores; the entry point's 8-rank fan-out with the ranks
# sharing the one GPU, file by file against the 1-rank run (3000 frames); bench.py's own 8-ranks-share-one-GPU line with the PNG leg.
set -u
OUT=gpurun_out/r6a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 600 python tools/png_scale.py --ranks 1,2,4,8 --workers 8,16,32 --frames 375 --out $OUT/png_scale_tmpfs.json > $OUT/png_scale_tmpfs.log 2>&1; tail -4 $OUT/png_scale_tmpfs.log | cut -c1-260
timeout 600 python tools/png_scale.py --ranks 8 --workers 16,32 --frames 375 --pace 750 --out $OUT/png_scale_tmpfs_paced_750.json > $OUT/png_scale_paced.log 2>&1; tail -4 $OUT/png_scale_paced.log | cut -c1-260
mkdir -p /tmp/gf_png_disk; timeout 600 python tools/png_scale.py --ranks 8 --workers 32 --frames 375 --base /tmp/gf_png_disk --out $OUT/png_scale_disk.json > $OUT/png_scale_disk.log 2>&1; tail -2 $OUT/png_scale_disk.log | cut -c1-260
timeout 900 python tools/fanout_check.py --frames 3000 --ranks 8 --json $OUT/fanout_8_ranks_one_gpu.json > $OUT/fanout.log 2>&1; tail -2 $OUT/fanout.log | cut -c1-600
timeout 900 python bench.py --gpus 8 --ranks-share-gpu --steps 20 --warmup 5 --no-stress --no-cpu-baseline > $OUT/bench_8_ranks_share_gpu.json 2> $OUT/bench_8_ranks_share_gpu.err; cut -c1-400 $OUT/bench_8_ranks_share_gpu.json
