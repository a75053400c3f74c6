#!/bin/bash
# A/B: option tail_overlap (producer's backward + teacher's gradient sweep on the side stream, beside the epilogue and the reduction)
set -e
mkdir -p gpurun_out/r05f
out=gpurun_out/r05f/ab_tail.txt
: > $out
for rep in 1 2 3; do
for v in 0 1; do
  python bench.py --mode step --opt tail_overlap=$v --no-cpu-baseline --train-steps 0 --steps 300 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('tail_overlap=$v', 'cold', round(d['ms_per_step'],4), 'warm', round(d['warm_ms_per_step'],4), 'channels_last cold', d.get('channels_last',{}).get('ms_per_step'))" >> $out
done
done
cat $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05f/tl -o tl -- python3 bench.py --mode step --opt tail_overlap=1 --regime warm --no-cpu-baseline --train-steps 0 --steps 60 > gpurun_out/r05f/tl.log 2>&1
python scripts/step_timeline.py $(find gpurun_out/r05f/tl -name "*kernel_trace.csv" | head -1) > gpurun_out/r05f/timeline_tail.txt
cat gpurun_out/r05f/timeline_tail.txt
find gpurun_out/r05f/tl -name "*.csv" -size +20M -delete
