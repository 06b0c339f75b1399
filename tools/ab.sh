#!/bin/bash
# Same-box A/B of two builds of the library (zinc_amd/lib/ab_A.so, ab_B.so): alternates them under bench.py.
# usage (on the GPU box): tools/ab.sh [bench.py arguments]
L=zinc_amd/lib
cp $L/libzinc_groth16_hip.so $L/ab_keep.so
for v in A B A B A B; do
  cp $L/ab_$v.so $L/libzinc_groth16_hip.so
  timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { cp $L/ab_keep.so $L/libzinc_groth16_hip.so; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3))"
done
cp $L/ab_keep.so $L/libzinc_groth16_hip.so
