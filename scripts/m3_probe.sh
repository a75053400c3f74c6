#!/bin/bash
# march3 role-elimination probe: teacher-kernel time (HIP events, eager launches) with roles switched off (debug bits 256 = A,
# 512 = S, 1024 = G; results are then wrong, only the time means something) and with longer tasks (march_rows)
O=gpurun_out/${1:-m3probe}; mkdir -p $O
run() {
  timeout -k 10 200 python bench.py --mode distil --steps 100 --warmup 20 --no-cpu-baseline --train-steps 0 "$@" 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms/step', round(d['ms_per_step'],4), 'teacher us', round(d['roofline']['kernel_ms']*1e3,1))"
}
{
run --opt march3=0
run
run --opt debug=256
run --opt debug=512
run --opt debug=1024
run --opt debug=768
run --opt debug=1280
run --opt debug=1536
run --opt debug=1792
run --opt march_rows=26
run --opt march_rows=26 --opt march3=0
run --opt march_rows=20
} | tee $O/probe.txt
