#!/bin/bash
# Round 5, on the GPU box: FETCH_SIZE / WRITE_SIZE of the north-star kernel and of the step's kernels in both regimes
# (separate --pmc passes, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/r05m/.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for regime in cold warm; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${regime}_$c -o p -- python3 $R/scripts/gpu_pmc_regime_target.py $regime > $O/${regime}_$c.log 2>&1 || { echo "$regime $c failed"; tail -3 $O/${regime}_$c.log; }
    echo "$regime $c done" >> $O/progress.log
  done
done
cd $R && for regime in cold warm; do for c in FETCH_SIZE WRITE_SIZE; do echo "== $regime $c"; python scripts/pmc_summary.py $O/${regime}_$c/p_counter_collection.csv "march_teacher_kernel<false" "march_teacher_kernel<true" "pack_identity" "march_kernel<false, true" "march_kernel<false, false" "march_student" "photo_march" "step_epilogue" "step_assemble"; done; done
