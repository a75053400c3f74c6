#!/bin/bash
# Round 5, on the GPU box: rocprofv3 kernel stats of the headline bench in the COLD regime (the timed steps rotate over six
# batches, the north-star kernel over eight) and in the WARM one (one batch replayed), and of the whole training step.
# Outputs under gpurun_out/r05p/; scripts/copy_profiles.sh-style copies into profiles/ are made by hand (named r05_*).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for regime in cold warm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$regime -o s -- python3 $R/bench.py --regime $regime --no-cpu-baseline --train-steps 0 > $O/stats_$regime.log 2>&1 || exit 1
  echo "stats $regime done" >> $O/progress.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -o s -- python3 $R/bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline > $O/stats_train.log 2>&1 || exit 1
echo "stats train done" >> $O/progress.log
