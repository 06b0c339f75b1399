import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import zinc_amd
from zinc_amd import workload, standin
from zinc_amd.groth16 import prove_recorded
from zinc_amd.fields import ints_to_mont, R_MOD
TOXIC = dict(k1=0x1234567, k2=0x7654321, alpha=0xA11CE5, beta=0xBEEF77, gamma=0x6A33A, delta=0xDE17A, tau=0x7A0F00D)
ctx = zinc_amd.Context(0)
rs = ints_to_mont([3, 5], R_MOD)
def cold(params, r1cs, zi, za, tag):
    for rep in range(2):
        c = zinc_amd.Context(0)
        t0 = time.perf_counter(); key = params.upload(c); t1 = time.perf_counter(); mat = r1cs.upload(c); t2 = time.perf_counter()
        prove_recorded(c, key, mat, zi, za, rs[0], rs[1]); t3 = time.perf_counter()
        prove_recorded(c, key, mat, zi, za, rs[0], rs[1]); t4 = time.perf_counter()
        print(tag, "rep", rep, "key %.1f ms  matrices %.1f ms  first proof %.1f ms  second %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3), flush=True)
        key.free(); mat.free(); c.close()
circ = standin.FactorialCircuit(6); pa = standin.record(circ); r1cs = pa.r1cs(); fact_r1cs = r1cs
params = workload.generate_parameters(ctx, r1cs, TOXIC); zi, za = pa.witness(); fact_params = params
cold(params, r1cs, zi, za, "factorial(1.8k rows)")
for lg in (12, 14, 16):
    c = workload.SyntheticCircuit(1 << lg); r = c.r1cs(); p = workload.generate_parameters(ctx, r, TOXIC); zi, za = c.witness()
    cold(p, r, zi, za, "synthetic 2^%d" % lg)

# the same key read from its file bytes (bellman Parameters::read, checked): zg16_pk_upload_bytes
from zinc_amd import codec
from zinc_amd.fields import mont_to_g1, mont_to_g2
from zinc_amd.groth16 import DeviceKey
p = fact_params
k = codec.KeyPoints()
k.alpha_g1, k.beta_g1, k.delta_g1 = mont_to_g1(p.alpha_g1)[0], mont_to_g1(p.beta_g1)[0], mont_to_g1(p.delta_g1)[0]
k.beta_g2, k.gamma_g2, k.delta_g2 = mont_to_g2(p.beta_g2)[0], mont_to_g2(p.gamma_g2)[0], mont_to_g2(p.delta_g2)[0]
k.ic = mont_to_g1(p.ic)
k.h, k.l, k.a, k.b_g1, k.b_g2 = mont_to_g1(p.h), mont_to_g1(p.l), mont_to_g1(p.a), mont_to_g1(p.b_g1), mont_to_g2(p.b_g2)
blob = codec.params_write(k)
for rep in range(2):
    c = zinc_amd.Context(0)
    t0 = time.perf_counter()
    dk = DeviceKey.from_bytes(c, blob, checked=True)
    print("factorial(1.8k rows) key file of %d bytes decoded, checked and expanded: %.1f ms" % (len(blob), (time.perf_counter() - t0) * 1e3), flush=True)
    dk.free()
    c.close()
