# prove latency of small circuits: usage tools/small_sweep.sh [extra bench.py flags]
for lr in ${SWEEP_LOG_ROWS:-8 10 12 14 16}; do
  v=$(timeout -k 10 120 python bench.py --log-rows $lr --steps 30 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  gpu total %.3f' % (d['ms_per_step'], d['stage_ms_last_step']['total_ms']))") || exit 1
  echo "log_rows=$lr $* => $v"
done
