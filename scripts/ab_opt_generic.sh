#!/bin/bash
# ab_opt_generic.sh option v1 v2 ... : same-box A/B of one option of the headline step (two rounds)
set -e
opt=$1; shift
mkdir -p gpurun_out/r05g
out=gpurun_out/r05g/ab_$opt.txt
: > $out
for rep in 1 2; do
for v in "$@"; do
  python bench.py --mode step --opt $opt=$v --no-cpu-baseline --train-steps 0 --steps 300 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$opt=$v', 'cold', round(d['ms_per_step'],4), 'warm', round(d['warm_ms_per_step'],4))" >> $out
done
done
cat $out
