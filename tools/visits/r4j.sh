#!/bin/bash
# round 4, visit j: frames in flight / hardware queues on the final kernels (the kernels got 5-15 % faster this round: has the optimum moved?)
set -u
OUT=gpurun_out/r4j; mkdir -p $OUT
export TMPDIR=/tmp
for q in 8 16; do
  for prec in fp32 split; do
    for n in 3 4 5 6; do
      GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --precision $prec --in-flight $n --steps 100 --warmup 10 --no-stress --png-frames 0 --no-cpu-baseline --profile-frames 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues=$q prec=$prec in_flight=$n fps=%.1f host_enqueue_ms=%.3f' % (d['value'], d.get('host_enqueue_ms_per_step') or 0))" | tee -a $OUT/in_flight_sweep.txt
    done
  done
done
