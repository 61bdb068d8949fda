#!/bin/bash
# round 4, visit f: validation of the final tree (paired rows + skinny layers from the accumulators) -- GPU suite, the driver's command, the default command, rocprofv3 kernel stats, PMC passes, split line
set -u
OUT=gpurun_out/r4f; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q -rfE -s > $OUT/pytest_full.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_full.log | cut -c1-300
grep -o "reference training step.*" $OUT/pytest_full.log | cut -c1-200
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "driver-cmd bench rc=$?"
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "default bench rc=$?"
python - <<'PY'
import json
for name in ("bench_driver_cmd", "bench"):
    try:
        d = json.load(open(f"gpurun_out/r4f/{name}.json"))
    except Exception as e:
        print(name, "unreadable", e); continue
    p = d["parity"]; r = d["roofline"]
    print(name, "value %.1f ms %.4f frac %.4f pipelined %.4f msamples/s %.1f" % (d["value"], d["ms_per_step"], r["frac"], r["pipelined"]["frac"], d["msamples_per_s"]))
    print("  parity max %.3g psnr %.1f graze %s pose %s" % (p["max_abs_rgb"], p["psnr_db"], p["grazing"]["pixels"], {k: v for k, v in p["pose_mode"].items() if k != "note"}))
    print("  cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "s_per_frame_by_threads")})
    for k in ("stress_fixture", "heavy_fixture", "head_only", "split_tier", "with_png"):
        v = d.get(k) or {}
        print("  ", k, v.get("value"), v.get("roofline_frac"), (v.get("parity") or {}).get("max_abs_rgb"), v.get("error"))
    print("  train", json.dumps(d.get("train_step"))[:900])
    print("  per_rank", json.dumps(d.get("per_rank"))[:300])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof -o k --output-format csv -- python $REPO/bench.py --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap > $REPO/$OUT/prof.log 2>&1); head -8 $OUT/prof/k_kernel_stats.csv | cut -c1-160
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_overlap -o k --output-format csv -- python $REPO/bench.py --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline > $REPO/$OUT/prof_overlap.log 2>&1); head -5 $OUT/prof_overlap/k_kernel_stats.csv | cut -c1-160
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_split -o k --output-format csv -- python $REPO/bench.py --precision split --steps 30 --warmup 5 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --no-overlap > $REPO/$OUT/prof_split.log 2>&1); head -5 $OUT/prof_split/k_kernel_stats.csv | cut -c1-160
B="python $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 0 --no-overlap"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F32 -d $REPO/$OUT/pmc_sq -o sq --output-format csv -- $B > $REPO/$OUT/pmc_sq.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $REPO/$OUT/pmc_fetch -o f --output-format csv -- $B > $REPO/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $REPO/$OUT/pmc_write -o w --output-format csv -- $B > $REPO/$OUT/pmc_write.log 2>&1)
python tools/pmc_summary.py $OUT 2>&1 | tail -30
timeout 900 python bench.py --precision split > $OUT/bench_split.json 2> $OUT/bench_split.err; echo "split bench rc=$?"; cut -c1-400 $OUT/bench_split.json
