#!/bin/bash
# same-box A/B of library variants (scripts/build_variant.py; "default" = the in-tree library), --distil and the headline
O=gpurun_out/ab_rot; mkdir -p $O
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset MAL_HIP_LIB; else export MAL_HIP_LIB=$PWD/mal_amd/lib/$v.so; fi
    for mode in distil step; do
      python bench.py --mode $mode --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$mode', 'ms/step', round(d['ms_per_step'],4), 'teacher us (events)', round(d['roofline']['kernel_ms']*1e3,1))" || exit 1
    done
  done
done | tee $O/ab.txt
unset MAL_HIP_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --mode distil --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_step -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 > $GRAFT_REPO_ROOT/$O/stats_step.log 2>&1
