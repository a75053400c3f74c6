#!/bin/bash
# same-box A/B of library variants built by scripts/build_variant.py ("default" = the in-tree library):
#   ab.sh name1 name2 ...   (two rounds each)
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset MAL_HIP_LIB; else export MAL_HIP_LIB=$PWD/mal_amd/lib/$v.so; fi
    python bench.py --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms']*1e3,1))" || exit 1
  done
done
