#!/usr/bin/env python3
"""Stand-alone G1 multiexp over resident window tables at several sizes: seconds and points per second.
usage: msm_sweep.py [log_n ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import zinc_amd
from zinc_amd import workload
from zinc_amd.groth16 import ResidentBases
from bench import rand_fr

ctx = zinc_amd.Context(0)
g1gen, _ = workload._generators()
for log_n in [int(a) for a in sys.argv[1:]] or [20, 22, 24, 26]:
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    bases = workload.fixed_base_g1(ctx, g1gen, rand_fr(rng, n))
    res = ResidentBases(ctx, 1, bases, window_tables=True)
    del bases
    sc = torch.from_numpy(rand_fr(rng, n).view(np.int64)).to("cuda:0")
    ref = res.msm(device_ptr=sc.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5 if log_n <= 24 else 3
    for _ in range(reps):
        out = res.msm(device_ptr=sc.data_ptr())
    dt = (time.perf_counter() - t0) / reps
    assert np.array_equal(out, ref)
    print("2^%d  %.3f ms  %.1f M points/s" % (log_n, dt * 1e3, n / dt / 1e6), flush=True)
    res.free()
    del sc
