#!/bin/bash
# Poll rocm-smi (power, sclk) while bench.py runs: is the head kernel's effective clock a power-cap effect?
cd $(dirname $0)/..
OUT=gpurun_out/${1:-power}; mkdir -p $OUT
python bench.py --steps ${2:-8000} --warmup 10 --no-cpu-baseline --profile-frames 0 > $OUT/bench_long.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' ' ; echo
done > $OUT/smi.txt
wait $BP
sort $OUT/smi.txt | uniq -c | sort -k1,1nr | head -12
cat $OUT/bench_long.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fps', d['value'])"
