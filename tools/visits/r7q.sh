#!/bin/bash
# round 6, visit q: the last tree of the round (Python-side changes after r7n: autocast-proof glue, .half() = the f16 tier, DDP test) -- the full GPU
# suite and the driver's command once more; the kernel sources (digest) are those of r7m / r7n
set -u
OUT=gpurun_out/r7q; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/pytest_gpu_full_suite.log; tail -3 $OUT/pytest_gpu_full_suite.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo rc=$?; cp gpurun_out/bench_details.json $OUT/
cut -c1-400 $OUT/bench_driver_command.json
python - <<'P'
import json
d = json.loads(open("gpurun_out/r7q/bench_driver_command.json").read())
r = d["roofline"]; t = d["train_step"]
print("fps", d["value"], "frac", r["frac"], "traffic_stale", r.get("traffic_stale"), "issue", d.get("issue_roofline", r.get("issue_roofline")))
print("train", {k: t.get(k) for k in ("ms_per_step", "amp_ms_per_step", "torso_ms_per_step")})
P
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
