#!/bin/bash
# round 4, visit a: the new whole-sequence parity tests, the parity hunt, the driver's exact bench command
set -u
OUT=gpurun_out/r4a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rfE -s > $OUT/pytest_full.log 2>&1; tail -25 $OUT/pytest_full.log | cut -c1-400
grep -E "^sweep:|^frame 14:" $OUT/pytest_full.log | cut -c1-600
timeout 600 python tools/parity_hunt.py --out $OUT/parity_hunt.json > $OUT/parity_hunt.log 2>&1; tail -12 $OUT/parity_hunt.log | cut -c1-900
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"; tail -3 $OUT/bench_driver_cmd.err | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4a/bench_driver_cmd.json"))
p = d["parity"]
print("value", d["value"], "frac", d["roofline"]["frac"], "parity max", p["max_abs_rgb"], "psnr", p["psnr_db"], "graze", p.get("grazing"), "pose", p["pose_mode"])
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "s_per_frame_by_threads")})
for k in ("stress_fixture", "heavy_fixture", "head_only", "split_tier"):
    v = d.get(k) or {}
    print(k, v.get("value"), (v.get("parity") or {}).get("max_abs_rgb"))
PY
