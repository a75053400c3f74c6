#!/bin/bash
# SQ issue/stall counters of the whole loss step (three steps), one rocprofv3 pass per counter group.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/stalls; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o p -- python3 $R/scripts/gpu_step_target.py > $O/g$i.log 2>&1 || { echo "group $i ($grp) failed"; tail -3 $O/g$i.log; }
  echo "group $i done" >> $O/progress.log
done
cd $R && for f in $O/g*/p_counter_collection.csv; do python scripts/pmc_summary.py $f "march_teacher_kernel<false"; done
