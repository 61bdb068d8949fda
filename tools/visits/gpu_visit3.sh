#!/bin/bash
set -u
OUT=gpurun_out/r3c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/split_diag.py 1 37 73 109 2>&1 | grep -v Warning | tee $OUT/split_diag.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/pytest.log; tail -8 $OUT/pytest.log
