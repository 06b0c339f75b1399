#!/usr/bin/env python3
"""Proves the factorial.zn stand-in circuit in a loop (for a kernel trace of a small proof, or with
ZG16_HOST_TRACE=1 for the host's side of it).  usage: small_trace.py [proofs]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import zinc_amd
from zinc_amd import standin, workload
from zinc_amd.fields import ints_to_mont, R_MOD
from zinc_amd.groth16 import prove_recorded

ctx = zinc_amd.Context(0)
pa = standin.record(standin.FactorialCircuit(6))
r1cs = pa.r1cs()
params = workload.generate_parameters(ctx, r1cs, dict(k1=5, k2=7, alpha=11, beta=13, gamma=17, delta=19, tau=23))
key, mat = params.upload(ctx), r1cs.upload(ctx)
zi, za = pa.witness()
rs = ints_to_mont([pow(7, 123456789, R_MOD), pow(5, 987654321, R_MOD)], R_MOD)  # full-size, as Fr::rand gives
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(5):
    prove_recorded(ctx, key, mat, zi, za, rs[0], rs[1])
t0 = time.perf_counter()
for _ in range(n):
    prove_recorded(ctx, key, mat, zi, za, rs[0], rs[1])
print("%.1f us per proof, %d rows" % ((time.perf_counter() - t0) / n * 1e6, r1cs.n_rows))
