#!/bin/bash
# same-box A/B: "name[:opt=val[,opt=val]]" entries; name = library variant ("default" = in-tree), opts = mal_set_option
for round in 1 2; do
  for e in "$@"; do
    v=${e%%:*}; o=""
    if [[ "$e" == *:* ]]; then for kv in ${e#*:}; do o="$o --opt $kv"; done; o=${o//,/ --opt }; fi
    if [ "$v" = default ]; then unset MAL_HIP_LIB; else export MAL_HIP_LIB=$PWD/mal_amd/lib/$v.so; fi
    python bench.py --steps 400 --warmup 50 --no-cpu-baseline --train-steps 0 $o 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms']*1e3,1))" || exit 1
  done
done
