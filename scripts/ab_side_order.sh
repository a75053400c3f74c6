#!/bin/bash
# A/B: option side_order (1: the student's pass first on the side stream, the ensemble pass behind it -- beside the fused sweep)
set -e
mkdir -p gpurun_out/r05g
out=gpurun_out/r05g/ab_side_order.txt
: > $out
for rep in 1 2; do
for v in 0 1; do
  python bench.py --mode step --opt side_order=$v --no-cpu-baseline --train-steps 0 --steps 300 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('side_order=$v', 'cold', round(d['ms_per_step'],4), 'warm', round(d['warm_ms_per_step'],4))" >> $out
done
done
cat $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05g/tl -o tl -- python3 bench.py --mode step --opt side_order=1 --regime warm --no-cpu-baseline --train-steps 0 --steps 60 > gpurun_out/r05g/tl.log 2>&1
python scripts/step_timeline.py $(find gpurun_out/r05g/tl -name "*kernel_trace.csv" | head -1) > gpurun_out/r05g/timeline_side_order1.txt
cat gpurun_out/r05g/timeline_side_order1.txt
find gpurun_out/r05g/tl -name "*.csv" -size +20M -delete
