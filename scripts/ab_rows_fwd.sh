for r in 0 9 10 11 0 9; do
  python bench.py --mode step --opt march_rows_fwd=$r --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step rows_fwd=$r', round(d['ms_per_step'],4))"
done
for r in 0 9 10; do
  python bench.py --mode distil --opt march_rows_fwd=$r --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('distil rows_fwd=$r', round(d['ms_per_step'],4))"
done
