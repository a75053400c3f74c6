#!/bin/bash
# Round 5: what would a third wave per SIMD buy the teacher's gradient pass (the north-star kernel)?  Same-box A/B of the
# shipped library against the -DMAL_PROBE_W3 build (168-VGPR cap -> scratch spills, LDS ring folded to 13.3 KB; WRONG results
# on purpose: only the time counts).  Three waves per SIMD need >= 3072 tasks to exist at all, i.e. 9-row tasks (2904) instead
# of 13-row ones (1980): march_rows is set accordingly.  `--regime warm` = the kernel replayed alone on one batch.
#   scripts/build_variant.py w3 -DMAL_PROBE_W3 && scripts/w3_probe.sh
run() {  # name lib rows
  if [ "$2" = default ]; then unset MAL_HIP_LIB; else export MAL_HIP_LIB=$PWD/mal_amd/lib/$2.so; fi
  python bench.py --mode distil --regime warm --steps 200 --warmup 20 --no-cpu-baseline --train-steps 0 --opt march_rows=$3 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 rows=$3', 'teacher replayed us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4), '--distil ms/step', round(d['ms_per_step'],4))" || exit 1
}
for round in 1 2; do
  run "shipped(2 waves/SIMD)" default 13
  run "shipped(2 waves/SIMD)" default 9
  run "w3(3 waves/SIMD,spills)" w3 13
  run "w3(3 waves/SIMD,spills)" w3 11
  run "w3(3 waves/SIMD,spills)" w3 9
done
