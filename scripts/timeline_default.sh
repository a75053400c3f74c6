#!/bin/bash
set -e
mkdir -p gpurun_out/r05e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05e/tl1 -o tl1 -- python3 bench.py --mode step --regime warm --no-cpu-baseline --train-steps 0 --steps 60 > gpurun_out/r05e/tl1.log 2>&1
python scripts/step_timeline.py $(find gpurun_out/r05e/tl1 -name "*kernel_trace.csv" | head -1) > gpurun_out/r05e/timeline_default.txt
cat gpurun_out/r05e/timeline_default.txt
find gpurun_out/r05e/tl1 -name "*.csv" -size +20M -delete
