#!/bin/bash
# round 4, visit o: the committed tree at the end of the round -- GPU suite, smoke, the driver's command
set -u
OUT=gpurun_out/r4o; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rfE > $OUT/pytest_full.log 2>&1; tail -3 $OUT/pytest_full.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "driver-cmd bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4o/bench_driver_cmd.json"))
p = d["parity"]; r = d["roofline"]
print("value %.1f ms %.4f frac %.4f pipelined %.4f" % (d["value"], d["ms_per_step"], r["frac"], r["pipelined"]["frac"]), "traffic", r["traffic"], r["traffic_source"])
print("parity max %.3g psnr %.1f graze %s pose %s" % (p["max_abs_rgb"], p["psnr_db"], p["grazing"]["pixels"], {k: v for k, v in p["pose_mode"].items() if k != "note"}))
for k in ("stress_fixture", "heavy_fixture", "head_only", "split_tier", "with_png"):
    v = d.get(k) or {}
    print("  ", k, v.get("value"), v.get("roofline_frac"), (v.get("parity") or {}).get("max_abs_rgb"), v.get("error"))
t = d.get("train_step") or {}
print("train", t.get("ms_per_step"), (t.get("reference_kernels_same_host_code") or {}).get("ms_per_step"), t.get("speedup_vs_reference_kernels"), (t.get("gradient_parity") or {}).get("worst_relative_l2"), t.get("error"))
PY
