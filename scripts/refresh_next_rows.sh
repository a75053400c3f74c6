#!/bin/bash
# On the GPU box (gpurun): measurements of the rows beyond the headline step -- operator route, temporal route, the
# training harness (N1), the temporal-hint producer (N2), the cost volume (N3) and the epipolar lookup / direct alignment (N4) -- with rocprofv3 kernel stats for
# the two kernels-only benches.  Outputs land in gpurun_out/next/; copy what should be judged into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/next
mkdir -p $O
cd $R
python bench.py --mode ops --no-cpu-baseline > $O/ops_bench.json 2> $O/ops.err || exit 1
python bench.py --mode temporal --no-cpu-baseline > $O/temporal_bench.json 2> $O/temporal.err || exit 1
python bench.py --mode multiscale_ops --no-cpu-baseline --train-steps 0 > $O/multiscale_ops_bench.json 2> $O/multiscale_ops.err || exit 1
python bench.py --mode dualrefine --no-cpu-baseline --train-steps 0 > $O/dualrefine_bench.json 2> $O/dualrefine.err || exit 1
python bench.py --mode train --steps 20 --warmup 5 > $O/train_bench.json 2> $O/train.err || exit 1
python scripts/bench_costvol.py > $O/costvol.txt 2>&1 || exit 1
python scripts/bench_dyn.py > $O/dyn.txt 2>&1 || exit 1
python scripts/bench_epipolar.py > $O/epipolar.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/costvol_stats -o s -- python3 $R/scripts/bench_costvol.py > $O/costvol_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dyn_stats -o s -- python3 $R/scripts/bench_dyn.py > $O/dyn_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/epipolar_stats -o s -- python3 $R/scripts/bench_epipolar.py > $O/epipolar_stats.log 2>&1 || exit 1
tail -1 $O/costvol.txt; tail -2 $O/dyn.txt; tail -2 $O/epipolar.txt
