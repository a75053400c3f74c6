#!/bin/bash
# same-box A/B of library variants (scripts/build_variant.py; "default" = the in-tree library) on the headline step
# usage: ab_step.sh OUTDIR variant...      (MODES="step distil" to change the bench modes)
O=gpurun_out/$1; shift; mkdir -p $O
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset MAL_HIP_LIB; else export MAL_HIP_LIB=$PWD/mal_amd/lib/$v.so; fi
    for mode in ${MODES:-step}; do
      timeout -k 10 200 python bench.py --mode $mode --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$mode', 'ms/step', round(d['ms_per_step'],4), 'eager', round(d.get('eager_ms_per_step') or 0,4))" || exit 1
    done
  done
done | tee $O/ab.txt
