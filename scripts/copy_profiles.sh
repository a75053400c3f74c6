#!/bin/bash
# gpurun_out/refresh/ (scripts/refresh_profiles.sh on the GPU box) -> profiles/rNN_* (tracked).  usage: copy_profiles.sh r02
set -e
R=${1:-r02}
O=gpurun_out/refresh
P=profiles
cp $O/bench.json $P/${R}_bench.json
[ -f $O/bench_driver.json ] && cp $O/bench_driver.json $P/${R}_bench_driver_command.json
[ -f mal_amd/lib/valu_cost.json ] && cp mal_amd/lib/valu_cost.json $P/${R}_valu_cost.json
cp $O/bench_distil.json $P/${R}_bench_distil.json
cp $O/bench_multiscale.json $P/${R}_bench_multiscale.json
cp $O/stats/s_kernel_stats.csv $P/${R}_kernel_stats.csv
cp $O/stats_distil/s_kernel_stats.csv $P/${R}_kernel_stats_distil.csv
cp $O/stats_multiscale/s_kernel_stats.csv $P/${R}_kernel_stats_multiscale.csv
for c in fetch write; do  # the library's kernels only (the capture also holds torch's fills / copies outside the step)
  { head -1 $O/$c/p_counter_collection.csv; grep "mal::" $O/$c/p_counter_collection.csv; } > $P/${R}_pmc_${c}_size.csv
done
cp $O/traffic.json $P/traffic.json
ls -la $P | grep ${R}_
