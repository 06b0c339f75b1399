"""Proof latency with the witness in host memory (pageable / page-locked / resident), as bench.py's
`prove_*_host_witness` entries measure it: python tools/host_witness.py [log_rows] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.set_device(0)
import bench
import zinc_amd

log_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = bench.Dist(0, 1, "nccl", 0)
ctx = zinc_amd.Context(0)
case = bench.ProveCase(ctx, d, log_rows, False, None)
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in case.host_witness(reps).items() if k != "workload"}))
