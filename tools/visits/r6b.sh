#!/bin/bash
# round 6, visit b: (1) tools/shadow_probe.hip -- what one wave per SIMD hides in the shadow of its own MFMA stream, against two waves per SIMD
# that alternate MFMA and VALU segments (the measurement that prices the 1-wave-per-SIMD / 512-register design point);
# (2) the box's CPU allowance (cgroup quota vs reported cores) and the PNG writers at stored blocks / planned pools; (3) bench.py with 8 ranks
# sharing the GPU, PNG leg on; (4) the torso training step: product vs the same host code over the reference's kernels (baseline before any
# fusing); (5) the whole GPU suite, not stopping at the first failure.
set -u
OUT=gpurun_out/r6b; mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/shadow_probe.hip -o /tmp/shadow_probe.bin 2>/dev/null
timeout 600 /tmp/shadow_probe.bin > $OUT/shadow_probe.txt 2>&1; cat $OUT/shadow_probe.txt
{ echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1)";
  python -c "import os; from geneface_amd.png import effective_cpus, plan_writer; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'effective', effective_cpus(), 'plan(1)', plan_writer(1), 'plan(8)', plan_writer(8))";
  lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"; } > $OUT/host_cpus.txt 2>&1; cat $OUT/host_cpus.txt
timeout 600 python tools/png_scale.py --ranks 1,8 --workers 2,4,8 --frames 375 --level 0 --layouts shared --out $OUT/png_scale_stored_blocks.json > $OUT/png_scale_stored.log 2>&1; cut -c1-330 $OUT/png_scale_stored.log
timeout 600 python tools/png_scale.py --ranks 8 --workers 2,4 --frames 375 --level 1 --layouts shared --out $OUT/png_scale_rle_small_pools.json > $OUT/png_scale_rle_small.log 2>&1; cut -c1-330 $OUT/png_scale_rle_small.log
timeout 900 python bench.py --gpus 8 --ranks-share-gpu --steps 20 --warmup 5 --no-stress --no-cpu-baseline > $OUT/bench_8_ranks_share_gpu.json 2> $OUT/bench_8_ranks_share_gpu.err; python -c "
import json; d=json.load(open('$OUT/bench_8_ranks_share_gpu.json')); print('8 ranks sharing: value', d['value'], 'with_png', d.get('with_png'))"
for i in 1 2; do
timeout 300 python tools/bench_train.py --torso 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso.jsonl | cut -c1-330
timeout 300 python tests/train_rate_reference.py --torso 2>/dev/null | tail -1 | tee -a $OUT/bench_train_torso_reference_kernels.jsonl | cut -c1-330
done
timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | tee $OUT/bench_train_head.json | cut -c1-250
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_train_torso -o k --output-format csv -- python $OLDPWD/tools/bench_train.py --torso --steps 32 --warmup 16 > $OLDPWD/$OUT/prof_train_torso.log 2>&1); head -25 $OUT/prof_train_torso/k_kernel_stats.csv | cut -c1-150
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest.log; tail -5 $OUT/pytest.log
