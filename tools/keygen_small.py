"""Key generation of the factorial.zn stand-in, three times on one context (the second and third find the fixed-base
tables of the generators in the context): host time per call.  usage: keygen_small.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import zinc_amd
from zinc_amd import standin, workload

ctx = zinc_amd.Context(0)
pa = standin.record(standin.FactorialCircuit(6))
r1cs = pa.r1cs()
for rep in range(3):
    t0 = time.perf_counter()
    params = workload.generate_parameters(ctx, r1cs, dict(k1=5, k2=7, alpha=11, beta=13, gamma=17, delta=19, tau=23 + rep))
    print("keygen %d: %.1f ms" % (rep, 1e3 * (time.perf_counter() - t0)), flush=True)
