#!/bin/bash
# same-box A/B of a mal_set_option switch on the bench modes: ab_opt3.sh OUTDIR "opt=0" "opt=1" ...  (MODES="distil step")
O=gpurun_out/$1; shift; mkdir -p $O
for round in 1 2 3; do
  for o in "$@"; do
    for mode in ${MODES:-distil step}; do
      timeout -k 10 200 python bench.py --mode $mode --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 --opt $o 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', '$mode', 'ms/step', round(d['ms_per_step'],4), 'teacher us (events)', round(d['roofline']['kernel_ms']*1e3,1))" || exit 1
    done
  done
done | tee $O/ab.txt
