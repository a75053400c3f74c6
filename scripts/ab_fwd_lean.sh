#!/bin/bash
# (the formulations this script compares lost: their code is in commit 1f05471 only -- check that commit out to re-run)
# A/B: the forward-only passes of the whole-step lists as march_forward_kernel (one load path + taps requested one iteration ahead)
set -e
mkdir -p gpurun_out/r05e
out=gpurun_out/r05e/ab_fwd_lean.txt
: > $out
python -m pytest tests/test_gpu_step.py tests/test_gpu_decisions.py -x -q -k "not dualrefine" > gpurun_out/r05e/fwd_lean_tests.txt 2>&1 || { tail -30 gpurun_out/r05e/fwd_lean_tests.txt; exit 1; }
tail -2 gpurun_out/r05e/fwd_lean_tests.txt
for rep in 1 2; do
for v in 0 1; do
  for mode in step distil; do
  python bench.py --mode $mode --opt fwd_lean=$v --no-cpu-baseline --train-steps 0 --steps 200 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$mode fwd_lean=$v', 'cold', round(d['ms_per_step'],4), 'warm', round(d['warm_ms_per_step'],4))" >> $out
  done
done
done
cat $out
