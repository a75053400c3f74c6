#!/bin/bash
# Round 5 probe (results WRONG on purpose; timing only): what does ONE vector-memory instruction per row cost a marching pass?
# Variants built with -DMAL_PROBE_NOLOADS=<bits> (1 ident, 2 the six gc loads, 4 forced_arg, 8 the student's mono+cost, 16 fin_gn,
# 32 the min_reproj store); per-kernel averages of the warm headline bench by rocprofv3.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in default nl1 nl2 nl4 nl8 nl16 nl32 nl63 default; do
  if [ "$v" = default ]; then unset MAL_HIP_LIB; else export MAL_HIP_LIB=$R/mal_amd/lib/$v.so; fi
  rm -rf $O/s_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$v -o s -- python3 $R/bench.py --regime warm --no-cpu-baseline --train-steps 0 --steps 300 --warmup 20 > $O/$v.log 2>&1 || { echo "$v failed"; tail -3 $O/$v.log; continue; }
  python3 - $O/s_$v/s_kernel_stats.csv $v <<'PY'
import csv,sys
rows={r["Name"]:r for r in csv.DictReader(open(sys.argv[1]))}
def avg(sub):
    for n,r in rows.items():
        if sub in n: return float(r["AverageNs"])/1e3
    return float("nan")
print("%-8s teacher<false> %.2f  teacher<true> %.2f  student_noepi %.2f  fwdwarp %.2f  ensemble %.2f  student<619> %.2f" % (
    sys.argv[2], avg("march_teacher_kernel<false, false>"), avg("march_teacher_kernel<true, false>"), avg("march_student_noepi_kernel"),
    avg("march_kernel<false, true, false, false, false, false>"), avg("march_kernel<false, false, false, false, false, false>"), avg("march_student_kernel<619>")))
PY
done
