#!/bin/bash
# What holds the shader clock while the head kernel runs?  Samples amd-smi (per-XCD average clocks, socket power, the firmware's
# "gfx clock below host limit" accounting by reason: power / thermal / low utilisation / total) during a long bench.py pass and, for
# comparison, during the steady MFMA probe with the same instruction mix (tools/mfma_probe.bin).   tools/clock_probe.sh <tag> <frames>
cd $(dirname $0)/..
OUT=gpurun_out/${1:-clock}; mkdir -p $OUT
SMI="/opt/rocm/bin/amd-smi metric -g 0 --clock --power --violation --json"
sample() {   # $1 = pid to follow, $2 = output file
  while kill -0 $1 2>/dev/null; do $SMI 2>/dev/null; echo "#----"; done > $2
}
python bench.py --steps ${2:-20000} --warmup 10 --no-cpu-baseline --no-stress --repeats 1 --png-frames 0 > $OUT/bench_long.json 2>/dev/null &
BP=$!
sample $BP $OUT/smi_bench.txt
wait $BP
if [ -x tools/mfma_probe.bin ]; then
  (for i in 1 2 3 4 5 6; do tools/mfma_probe.bin; done) > $OUT/mfma_probe.txt 2>&1 &
  PP=$!
  sample $PP $OUT/smi_probe.txt
  wait $PP
fi
python - <<PY
import json
def rows(path):
    out = []
    for t in open(path).read().split("#----"):
        t = t.strip()
        if t:
            try: out.append(json.loads(t)["gpu_data"][0])
            except Exception: pass
    return out
def val(x): return x["value"] if isinstance(x, dict) else x
def summarise(tag, path):
    try: rs = rows(path)
    except FileNotFoundError: return
    busy = [r for r in rs if val(r["power"]["socket_power"]) not in ("N/A",) and float(val(r["power"]["socket_power"])) > 600]
    print(f"== {tag}: {len(rs)} samples, {len(busy)} under load")
    if len(busy) < 3: return
    clk = [[float(val(r["clock"][f"gfx_{i}"]["clk"])) for i in range(8)] for r in busy]
    pw = [float(val(r["power"]["socket_power"])) for r in busy]
    print("   socket power W: min %.0f mean %.0f max %.0f" % (min(pw), sum(pw) / len(pw), max(pw)))
    print("   gfx clk MHz (mean over samples, per XCD):", [round(sum(c[i] for c in clk) / len(clk)) for i in range(8)])
    a, b = busy[0]["throttle"], busy[-1]["throttle"]
    def acc(t, k):
        v = t[k]
        return float(v["xcp_0"][0]) if isinstance(v, dict) else float(v)
    tot = acc(b, "accumulation_counter") - acc(a, "accumulation_counter")
    print("   accumulation_counter delta:", tot)
    for k in ("ppt_accumulated", "socket_thermal_accumulated", "vr_thermal_accumulated", "hbm_thermal_accumulated", "prochot_accumulated",
              "gfx_clk_below_host_limit_power_accumulated", "gfx_clk_below_host_limit_thermal_accumulated",
              "total_gfx_clk_below_host_limit_accumulated", "low_utilization_accumulated"):
        try:
            d = acc(b, k) - acc(a, k)
            print(f"   {k}: +{d:.0f} ({100 * d / max(tot, 1):.1f} % of the interval)")
        except Exception as e:
            print("   ", k, "n/a", e)
    mid = busy[len(busy) // 2]["throttle"]
    print("   status mid-run:", {k: (v["xcp_0"][0] if isinstance(v, dict) else v) for k, v in mid.items() if k.endswith("_status")})
summarise("head kernel (bench.py)", "$OUT/smi_bench.txt")
summarise("steady MFMA probe", "$OUT/smi_probe.txt")
d = json.load(open("$OUT/bench_long.json")); print("fps", d["value"], "frac", d["roofline"]["frac"])
PY
tail -3 $OUT/mfma_probe.txt 2>/dev/null
