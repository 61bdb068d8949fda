#!/bin/bash
# round 6, visit 8b: the five drawn-case scripts once more on seeds nobody has looked at (11-13), longer
set -u
OUT=gpurun_out/r8b; mkdir -p $OUT
for s in 11 12 13; do
  timeout 600 python tests/fuzz_ops_vs_reference_kernels.py --cases 1500 --seed $s --out $OUT/ops_$s.json > $OUT/ops_$s.log 2>&1; echo "ops $s rc=$?"
  timeout 600 python tests/fuzz_rays_vs_reference_kernels.py --cases 1500 --seed $s --out $OUT/rays_$s.json > $OUT/rays_$s.log 2>&1; echo "rays $s rc=$?"
  timeout 600 python tests/fuzz_frames_vs_reference_kernels.py --cases 2000 --seed $s --out $OUT/frames_$s.json > $OUT/frames_$s.log 2>&1; echo "frames $s rc=$?"
  timeout 600 python tests/fuzz_frame_loop_vs_reference_kernels.py --cases 500 --seed $s --out $OUT/loop_$s.json > $OUT/loop_$s.log 2>&1; echo "loop $s rc=$?"
  timeout 600 python tests/fuzz_train_vs_oracle.py --cases 1000 --seed $s --out $OUT/train_$s.json > $OUT/train_$s.log 2>&1; echo "train $s rc=$?"
done
for f in $OUT/*.log; do if ! tail -1 $f | grep -q '"cases"'; then echo "== $f"; tail -2 $f | cut -c1-600; fi; done
