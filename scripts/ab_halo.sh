#!/bin/bash
# same-box A/B of the one-row halo (mal_set_option("march_halo1", 0|1)), --distil and the --temporal headline, two rounds;
# rocprofv3 kernel stats of one run each
O=gpurun_out/ab_halo; mkdir -p $O
for round in 1 2; do
  for h in 0 1; do
    for mode in distil step; do
      python bench.py --mode $mode --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 --opt march_halo1=$h 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('halo1=$h', '$mode', 'ms/step', round(d['ms_per_step'],4), 'teacher kernel us (events, eager)', round(d['roofline']['kernel_ms']*1e3,1), 'temporal sweep us', round(d.get('roofline_temporal',{}).get('kernel_ms',0)*1e3,1))" || exit 1
    done
  done
done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for h in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_h$h -o s -- python3 $GRAFT_REPO_ROOT/bench.py --mode distil --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 --opt march_halo1=$h > $GRAFT_REPO_ROOT/$O/stats_h$h.log 2>&1 || exit 1
done
