#!/bin/bash
# round 6, visit b: tools/shadow_probe.hip -- what one wave per SIMD hides in the shadow of its own MFMA stream, against two waves per SIMD that
# alternate MFMA and VALU segments (the measurement that prices the 1-wave-per-SIMD / 512-register design point)
set -u
OUT=gpurun_out/r6b; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/shadow_probe.hip -o /tmp/shadow_probe.bin 2>/dev/null
timeout 600 /tmp/shadow_probe.bin > $OUT/shadow_probe.txt 2>&1; cat $OUT/shadow_probe.txt
