#!/bin/bash
# same-box A/B of the fused sweep's dispatch order and task height (mal_set_option("syn_queue" / "syn_rows")), headline step
O=gpurun_out/ab_syn; mkdir -p $O
for round in 1 2; do
  for cfg in "syn_queue=0 syn_rows=4" "syn_queue=1 syn_rows=4" "syn_queue=1 syn_rows=6" "syn_queue=1 syn_rows=3" "syn_queue=1 syn_rows=2"; do
    opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
    python bench.py --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 $opts 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'ms/step', round(d['ms_per_step'],4))" || exit 1
  done
done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1 || exit 1
