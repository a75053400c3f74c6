#!/bin/bash
# same-box A/B of a runtime option: ab_opt.sh NAME V1 V2 ...   (two rounds each)
name=$1; shift
for round in 1 2; do for v in "$@"; do
  python bench.py --steps 400 --warmup 50 --no-cpu-baseline --opt $name=$v 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name=$v', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms']*1e3,1))" || exit 1
done; done
