"""Developer tool: time zg16_pk_upload_bytes (GPU decode + checks of a proving-key file) on a 2^18-row key."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zinc_amd
from zinc_amd import workload, codec
from zinc_amd.fields import mont_to_g1, mont_to_g2
from zinc_amd.groth16 import DeviceKey
ctx = zinc_amd.Context(0)
circ = workload.SyntheticCircuit(1 << 18, n_pub=4)
params = workload.generate_parameters(ctx, circ, dict(k1=3, k2=5, alpha=7, beta=11, gamma=13, delta=17, tau=19))
k = codec.KeyPoints()
k.alpha_g1, k.beta_g1, k.delta_g1 = mont_to_g1(params.alpha_g1)[0], mont_to_g1(params.beta_g1)[0], mont_to_g1(params.delta_g1)[0]
k.beta_g2, k.delta_g2 = mont_to_g2(params.beta_g2)[0], mont_to_g2(params.delta_g2)[0]
k.gamma_g2 = k.beta_g2
k.ic = []
k.h, k.l, k.a, k.b_g1, k.b_g2 = mont_to_g1(params.h), mont_to_g1(params.l), mont_to_g1(params.a), mont_to_g1(params.b_g1), mont_to_g2(params.b_g2)
data = codec.params_write(k)
print("key bytes", len(data), "points", len(k.h), len(k.l), len(k.a), len(k.b_g1), len(k.b_g2), flush=True)
for checked in (False, True):
    t0 = time.time(); d = DeviceKey.from_bytes(ctx, data, checked=checked); t1 = time.time(); d.free()
    print("checked", checked, "%.3f s" % (t1 - t0), flush=True)
t0 = time.time(); d = params.upload(ctx); print("upload from Montgomery arrays %.3f s" % (time.time() - t0)); d.free()
