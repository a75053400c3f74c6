#!/bin/bash
# (the formulations this script compares lost: their code is in commit 1f05471 only -- check that commit out to re-run)
set -e
mkdir -p gpurun_out/r05e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
rocprofv3 --kernel-trace -d gpurun_out/r05e/kf$v -o kf -- python3 bench.py --mode distil --opt fwd_lean=$v --regime warm --no-cpu-baseline --train-steps 0 --steps 300 > gpurun_out/r05e/kf$v.log 2>&1
echo "fwd_lean=$v" >> gpurun_out/r05e/kstat_fwd.txt
python scripts/kstats.py gpurun_out/r05e/kf$v/kf_results.db | head -8 >> gpurun_out/r05e/kstat_fwd.txt
rm -rf gpurun_out/r05e/kf$v
done
cat gpurun_out/r05e/kstat_fwd.txt
