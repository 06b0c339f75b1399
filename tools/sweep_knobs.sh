#!/bin/bash
# usage: tools/sweep_knobs.sh "<ENV=VAL ...>" ... ; one bench.py run per argument, prints ms per proof
for cfg in "$@"; do
  v=$(env $cfg timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timers 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
  echo "$cfg => $v ms" | tee -a gpurun_out/sweep_knobs.log
done
