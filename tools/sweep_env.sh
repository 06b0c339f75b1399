# Run the headline of bench.py once per environment setting given as arguments ("VAR=value [VAR2=value]"), on the GPU
# box; BENCH_ARGS adds bench.py flags (e.g. "--log-rows 17").
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/sweep
i=0
for s in "$@"; do
  i=$((i+1))
  env $s timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 $BENCH_ARGS > gpurun_out/sweep/e_$i.json 2> gpurun_out/sweep/e_$i.err || exit 1
  echo "$s: $(python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sweep/e_$i.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3),'ms')")"
done
