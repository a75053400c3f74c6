#!/bin/bash
# same-box per-kernel A/B of mal_set_option switches (rocprofv3 averages of graph-replayed steps)
# usage: ab_kernels.sh OUTDIR "opt=0" "opt=1" ...     (MODES="distil step")
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for o in "$@"; do
  for mode in ${MODES:-distil step}; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${o}_$mode -o s -- python3 $GRAFT_REPO_ROOT/bench.py --mode $mode --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 --opt $o > $O/${o}_$mode.log 2>&1 || exit 1
    python3 - "$O/${o}_$mode/s_kernel_stats.csv" "$o $mode" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    if int(r["Calls"]) > 100: print(sys.argv[2], r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
  done
done | tee $O/ab_kernels.txt
