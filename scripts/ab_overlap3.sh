#!/bin/bash
# (the formulations this script compares lost: their code is in commit 1f05471 only -- check that commit out to re-run)
# A/B: option step_overlap 3 (the producer chain as the warp pass's first successor in the captured graph) against the default
set -e
mkdir -p gpurun_out/r05e
out=gpurun_out/r05e/ab_overlap3.txt
: > $out
for rep in 1 2; do
for v in 1 3; do
  python bench.py --mode step --opt step_overlap=$v --no-cpu-baseline --train-steps 0 --steps 200 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('step_overlap=$v', 'cold', round(d['ms_per_step'],4), 'warm', round(d['warm_ms_per_step'],4))" >> $out
done
done
cat $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05e/tl3 -o tl3 -- python3 bench.py --mode step --opt step_overlap=3 --regime warm --no-cpu-baseline --train-steps 0 --steps 60 > gpurun_out/r05e/tl3.log 2>&1
python scripts/step_timeline.py $(find gpurun_out/r05e/tl3 -name "*kernel_trace.csv" | head -1) > gpurun_out/r05e/timeline_overlap3.txt
cat gpurun_out/r05e/timeline_overlap3.txt
find gpurun_out/r05e/tl3 -name "*.csv" -size +20M -delete
