#!/bin/bash
# Round 5, on the GPU box: the bench lines and rocprofv3 summaries that profiles/r05_* are copied from (gpurun_out/r05r/).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05r; mkdir -p $O
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || exit 1
echo "driver line done" >> $O/progress.log
python bench.py --loss-blc > $O/bench.json 2> $O/bench.err || exit 1
python bench.py --mode distil --train-steps 0 > $O/bench_distil.json 2> $O/bench_distil.err || exit 1
echo "bench lines done" >> $O/progress.log
python bench.py --mode multiscale --train-steps 0 --no-cpu-baseline > $O/bench_multiscale.json 2> $O/bench_multiscale.err || exit 1
python bench.py --mode multiscale --ms-temporal --train-steps 0 --no-cpu-baseline > $O/bench_multiscale_temporal.json 2> $O/bench_multiscale_temporal.err || exit 1
python bench.py --mode dualrefine --train-steps 0 --no-cpu-baseline > $O/bench_dualrefine.json 2> $O/bench_dualrefine.err || exit 1
python bench.py --main-temporal --train-steps 0 --no-cpu-baseline > $O/bench_main_temporal.json 2> $O/bench_main_temporal.err || exit 1
python bench.py --width 512 --train-steps 0 --no-cpu-baseline > $O/bench_cityscapes.json 2> $O/bench_cityscapes.err || exit 1
echo "mode lines done" >> $O/progress.log
cd /tmp && export TMPDIR=/tmp
for regime in cold warm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$regime -o s -- python3 $R/bench.py --regime $regime --no-cpu-baseline --train-steps 0 > $O/stats_$regime.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_distil_$regime -o s -- python3 $R/bench.py --mode distil --regime $regime --no-cpu-baseline --train-steps 0 > $O/stats_distil_$regime.log 2>&1 || exit 1
  echo "stats $regime done" >> $O/progress.log
done
[ -n "$SKIP_PMC" ] || { cd $R && bash scripts/r05_pmc.sh > $O/pmc.txt 2>&1; }
echo "pmc done" >> $O/progress.log
