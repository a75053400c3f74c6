#!/bin/bash
# (the formulations this script compares lost: their code is in commit 1f05471 only -- check that commit out to re-run)
# same-box A/B of the forward-only passes' formulations: library variants pipe0 (-DMAL_FWD_PIPE=0, the round-4 loop), pipe1 (taps of
# row r+1 issued together after row r's blend), default (pipe 2: spread between the stages) x option fwd_lean (0: generic instantiation)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05e; mkdir -p $O
out=$O/ab_fwd_pipe.txt; : > $out
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for spec in pipe0:fwd_lean=0 pipe1:fwd_lean=0 pipe1:fwd_lean=1 default:fwd_lean=1; do
  v=${spec%%:*}; opt="--opt ${spec#*:}"
  if [ "$v" = default ]; then unset MAL_HIP_LIB; else export MAL_HIP_LIB=$R/mal_amd/lib/$v.so; fi
  for regime in cold warm; do
  rm -rf $O/s_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$v -o s -- python3 $R/bench.py --mode step --regime $regime --no-cpu-baseline --train-steps 0 --steps 300 --warmup 20 $opt > $O/$v.log 2>&1 || { echo "$v failed" >> $out; tail -3 $O/$v.log >> $out; continue; }
  python3 - $O/s_$v/s_kernel_stats.csv "$spec $regime" $O/$v.log >> $out <<'PY'
import csv,sys,json
rows={r["Name"]:r for r in csv.DictReader(open(sys.argv[1]))}
def avg(*subs):
    for sub in subs:
        for n,r in rows.items():
            if sub in n: return float(r["AverageNs"])/1e3
    return float("nan")
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print("%-28s ms/step %.4f | pack %.2f fwdwarp %.2f ens %.2f student %.2f sweep %.2f teacherT %.2f" % (
    sys.argv[2], d["ms_per_step"], avg("pack_identity_kernel<false, false>"), avg("march_forward_kernel<true>", "march_kernel<false, true, false, false, false, false>"),
    avg("march_forward_kernel<false>", "march_kernel<false, false, false, false, false, false>"), avg("march_student_noepi_kernel"), avg("photo_march_bwd_kernel<true>"),
    avg("march_teacher_kernel<true, false>")))
PY
  rm -rf $O/s_$v
  done
done
done
cat $out
