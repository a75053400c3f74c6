#!/bin/bash
# Texture-addresser / vector-L1 counters of the marching kernels (is the gather rate what bounds the forward-only passes?)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/ta; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o p -- python3 $R/scripts/gpu_step_target.py > $O/g$i.log 2>&1 || { echo "group $i ($grp) failed"; tail -2 $O/g$i.log; }
done
cd $R && for f in $O/g*/p_counter_collection.csv; do python scripts/pmc_summary.py $f "march_teacher_kernel<false" "march_kernel<false, false" "march_student" "pack_identity"; done
