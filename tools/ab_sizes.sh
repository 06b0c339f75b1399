#!/bin/bash
# Same-box A/B of two builds (zinc_amd/lib/ab_A.so, ab_B.so) at several circuit sizes: tools/ab_sizes.sh "10 14 17 20"
L=zinc_amd/lib
cp $L/libzinc_groth16_hip.so $L/ab_keep.so
for lr in ${1:-10 14 17 20}; do
  for v in A B A B; do
    cp $L/ab_$v.so $L/libzinc_groth16_hip.so
    timeout -k 10 400 python bench.py $AB_EXTRA --log-rows $lr --steps ${AB_STEPS:-30} --warmup ${AB_WARMUP:-4} --no-secondary --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { cp $L/ab_keep.so $L/libzinc_groth16_hip.so; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); print('log_rows $lr', '$v', round(d['ms_per_step'],3), flush=True)"
  done
done
cp $L/ab_keep.so $L/libzinc_groth16_hip.so
