#!/bin/bash
# DualRefine one-call step with / without the pose-update losses, and the operator route (gpurun_out/r05d/)
set -e
mkdir -p gpurun_out/r05d
python -m pytest tests/test_gpu_decisions.py -x -q -k "pose_update" > gpurun_out/r05d/pu_tests.txt 2>&1
for v in "" "--dr-pose-update" "--dr-pose-update --dr-default-scales" "--dr-default-scales"; do
  tag=$(echo "dr$v" | tr -d ' ' | tr '-' '_')
  python bench.py --mode dualrefine $v --no-cpu-baseline --train-steps 0 2>/dev/null | tail -1 > gpurun_out/r05d/$tag.json
done
python bench.py --mode dualrefine_ops --dr-pose-update --no-cpu-baseline --train-steps 0 2>/dev/null | tail -1 > gpurun_out/r05d/dr_ops_pose_update.json
python bench.py --mode dualrefine_ops --no-cpu-baseline --train-steps 0 2>/dev/null | tail -1 > gpurun_out/r05d/dr_ops.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05d/dr*.json')):
    d=json.loads(open(f).read())
    print(f.split('/')[-1], 'cold', round(d['ms_per_step'],4), 'warm', round(d.get('warm_ms_per_step',0),4))
PY
