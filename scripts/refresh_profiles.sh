#!/bin/bash
# On the GPU box (gpurun): bench line, rocprofv3 kernel stats of the same command, HBM-traffic PMC passes.
# Outputs land in gpurun_out/refresh/; copy what should be judged into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refresh
mkdir -p $O
# the driver's own command, default flags (round 2 lost its headline because this exact line was never run after an edit)
cd $R && python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || exit 1
python bench.py --loss-blc > $O/bench.json 2> $O/bench.err || exit 1
python bench.py --mode distil --train-steps 0 > $O/bench_distil.json 2> $O/bench_distil.err || exit 1
python bench.py --mode multiscale --train-steps 0 --no-cpu-baseline > $O/bench_multiscale.json 2> $O/bench_multiscale.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-cpu-baseline --train-steps 0 > $O/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_distil -o s -- python3 $R/bench.py --mode distil --no-cpu-baseline --train-steps 0 > $O/stats_distil.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_multiscale -o s -- python3 $R/bench.py --mode multiscale --no-cpu-baseline --train-steps 0 > $O/stats_multiscale.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- python3 $R/scripts/gpu_step_target.py > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- python3 $R/scripts/gpu_step_target.py > $O/write.log 2>&1 || exit 1
cd $R && python scripts/make_traffic.py $O/fetch/p_counter_collection.csv $O/write/p_counter_collection.csv $O/traffic.json $O/bench.json > /dev/null
# bench.py read profiles/traffic.json of the PREVIOUS refresh: put this run's counter value into this run's line
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads(open(o + "/bench.json").read().strip().splitlines()[-1])
d["roofline"]["traffic"] = json.load(open(o + "/traffic.json"))["pass_kernel_teacher_bytes_per_launch"]
json.dump(d, open(o + "/bench.json", "w"))
PY
cat $O/bench.json
