#!/bin/bash
# round 6, visit t: 3 000-frame soaks (the length of BASELINE.json configs[3]) of the product's frames against the reference's own kernels on identical
# device rays: both strict tiers, both identities, and the shipped variants on the exact tier
set -u
OUT=gpurun_out/r7t; mkdir -p $OUT
run() { local name=$1; shift; timeout 1200 python tools/parity_hunt.py --T 3000 --frames "" "$@" --out $OUT/soak_3000_frames_vs_reference_kernels_$name.json > $OUT/soak_$name.log 2>&1; echo "$name rc=$?"; tail -2 $OUT/soak_$name.log | cut -c1-260; }
run fp32
run split --precision split
run fp32_identity_1000 --identity 1000
run split_identity_1000 --precision split --identity 1000
for v in hash hash_smoothstep smoothstep head_aware audio; do run fp32_$v --variant $v; done
