#!/bin/bash
# same-box A/B of library variants (scripts/build_variant.py; "default" = the in-tree library) in BOTH regimes: ms per step
# (cold = rotating over six batches, warm = one batch replayed) and the north-star kernel replayed alone (cold: eight batches)
#   ab_cold.sh [-m "distil step"] name1 name2 ...   (two rounds each; default modes: step)
modes="step"
if [ "$1" = "-m" ]; then modes="$2"; shift 2; fi
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset MAL_HIP_LIB; else export MAL_HIP_LIB=$PWD/mal_amd/lib/$v.so; fi
    for mode in $modes; do
      python bench.py --mode $mode --steps 300 --warmup 30 --no-cpu-baseline --train-steps 0 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', '$mode', 'ms/step cold', round(d['cold_ms_per_step'],4), 'warm', round(d['warm_ms_per_step'],4), '| teacher us cold', round(r['cold_kernel_ms']*1e3,2), 'warm', round(r['warm_kernel_ms']*1e3,2), '| channels_last cold', round(d['channels_last']['ms_per_step'],4))" || exit 1
    done
  done
done
