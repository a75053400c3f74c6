#!/bin/bash
# same-box per-kernel A/B of library variants: rocprofv3 per-dispatch averages of the headline bench in one regime
#   ab_kernels_r05.sh regime "name[:opt=val] ..."     ("default" = the in-tree library)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/abk; mkdir -p $O
regime=$1; shift
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}; opt=""; [ "$spec" != "$v" ] && opt="--opt ${spec#*:}"
  if [ "$v" = default ]; then unset MAL_HIP_LIB; else export MAL_HIP_LIB=$R/mal_amd/lib/$v.so; fi
  rm -rf $O/s_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$v -o s -- python3 $R/bench.py --regime $regime --no-cpu-baseline --train-steps 0 --steps 300 --warmup 20 $opt > $O/$v.log 2>&1 || { echo "$v failed"; tail -3 $O/$v.log; continue; }
  python3 - $O/s_$v/s_kernel_stats.csv "$spec" $O/$v.log <<'PY'
import csv,sys,json
rows={r["Name"]:r for r in csv.DictReader(open(sys.argv[1]))}
def avg(sub):
    for n,r in rows.items():
        if sub in n: return float(r["AverageNs"])/1e3
    return float("nan")
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print("%-16s ms/step %.4f | pack %.2f fwdwarp %.2f ens %.2f student %.2f sweep %.2f epi %.2f final %.2f teacherT %.2f assemble %.2f | teacher<false> %.2f" % (
    sys.argv[2], d["ms_per_step"], avg("pack_identity_kernel<false, false>"), avg("march_kernel<false, true, false, false, false, false>"),
    avg("march_kernel<false, false, false, false, false, false>"), avg("march_student_noepi_kernel"), avg("photo_march_bwd_kernel<true>"),
    avg("step_epilogue_kernel"), avg("step_final_kernel"), avg("march_teacher_kernel<true, false>"), avg("step_assemble_kernel"),
    avg("march_teacher_kernel<false, false>")))
PY
done
done
