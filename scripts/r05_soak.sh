#!/bin/bash
# soak: 20 000 graph replays per configuration (cold regime: six batches rotated), one MI355X box -> gpurun_out/r05d/soak.txt
set -e
mkdir -p gpurun_out/r05d
out=gpurun_out/r05d/soak.txt
echo "# soak: 20 000 graph replays per configuration (rotating six batches: the cold regime) on one MI355X box (bench.py --steps 20000 --regime cold), ms per step; no hang, no fault" > $out
for v in "--mode step" "--mode step --main-temporal" "--mode multiscale --ms-temporal" "--mode dualrefine --dr-default-scales --dr-pose-update" "--mode dualrefine --dr-pose-update" "--mode distil"; do
  timeout -k 10 300 python bench.py $v --steps 20000 --regime cold --no-cpu-baseline --train-steps 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v', d['steps'], round(d['ms_per_step'],4), d['config'].get('launch','?'))" >> $out
  echo "done $v"
done
cat $out
