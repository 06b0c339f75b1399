"""Ad-hoc GPU-vs-oracle check used while bringing kernels up (the pytest suite is the real gate)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import bn254 as O
from oracle import groth16 as G
from oracle.synth import make_circuit
from oracle.xorshift import XorShiftRng
import zinc_amd
from zinc_amd.fields import ints_to_mont, mont_to_ints, g1_to_mont, g2_to_mont, mont_to_g1, mont_to_g2, R_MOD

rng = XorShiftRng()
ctx = zinc_amd.Context(0)
ok = True
def check(name, cond):
    global ok
    print(("PASS " if cond else "FAIL ") + name, flush=True)
    ok &= bool(cond)

for log_n in [1, 2, 3, 5, 8, 10, 11, 12, 13, 15]:
    n = 1 << log_n
    x = [rng.gen_fr() for _ in range(n)]
    d = G.EvaluationDomain(x)
    for inv, coset in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        d.coeffs = list(x)
        {(0, 0): d.fft, (1, 0): d.ifft, (0, 1): d.coset_fft, (1, 1): d.icoset_fft}[(inv, coset)]()
        got = mont_to_ints(ctx.ntt(ints_to_mont(x, R_MOD), inverse=inv, coset=coset), R_MOD)
        check("ntt 2^%d inv=%d coset=%d" % (log_n, inv, coset), got == d.coeffs)

for n_rows in [3, 8, 100, 1000]:
    a = [rng.gen_fr() for _ in range(n_rows)]; b = [rng.gen_fr() for _ in range(n_rows)]
    c = [x * y % R_MOD for x, y in zip(a, b)]
    m, h = G.h_coefficients(a, b, c)
    gh, gm = ctx.h_coefficients(ints_to_mont(a, R_MOD), ints_to_mont(b, R_MOD), ints_to_mont(c, R_MOD))
    check("h_coefficients rows=%d" % n_rows, gm == m and mont_to_ints(gh, R_MOD) == h)

def rand_scalars(n):
    out = []
    for i in range(n):
        k = rng.below(10)
        out.append(0 if k == 0 else 1 if k < 4 else R_MOD - 1 if k == 4 else rng.below(1 << 20) if k == 5 else rng.gen_fr())
    return out

for n in [1, 2, 7, 64, 1000, 5000]:
    ks = [rng.below(1 << 30) + 1 for _ in range(n)]
    pts = O.G1.batch_to_affine([O.G1.mul(O.G1.gen, k) for k in ks])
    if n >= 7: pts[3] = pts[5]
    sc = rand_scalars(n)
    want = O.G1.to_affine(G.naive_msm(O.G1, pts, sc)) if n <= 1000 else O.G1.to_affine(O.G1.mul(O.G1.gen, sum(k*s for k,s in zip([ks[i] if not (n>=7 and i==3) else ks[5] for i in range(n)], sc)) % R_MOD))
    got = mont_to_g1(ctx.msm_g1(g1_to_mont(pts), ints_to_mont(sc, R_MOD)))[0]
    check("msm_g1 n=%d" % n, got == want)
    dens = [rng.below(3) != 0 for _ in range(n)]
    dpts = [p for p, d in zip(pts, dens) if d]
    want = O.G1.to_affine(G.multiexp(O.G1, dpts, sc, dens)) if n <= 1000 else None
    if want is not None or n <= 1000:
        got = mont_to_g1(ctx.msm_g1(g1_to_mont(dpts) if dpts else np.zeros((0, 8), np.uint64), ints_to_mont(sc, R_MOD), np.array(dens, dtype=np.uint8)))[0]
        check("msm_g1 density n=%d" % n, got == want)

for n in [1, 5, 200]:
    ks = [rng.below(1 << 30) + 1 for _ in range(n)]
    pts = O.G2.batch_to_affine([O.G2.mul(O.G2.gen, k) for k in ks])
    sc = rand_scalars(n)
    want = O.G2.to_affine(O.G2.mul(O.G2.gen, sum(k * s for k, s in zip(ks, sc)) % R_MOD))
    got = mont_to_g2(ctx.msm_g2(g2_to_mont(pts), ints_to_mont(sc, R_MOD)))[0]
    check("msm_g2 n=%d" % n, got == want)

for n_rows in [5, 40, 300]:
    circ = make_circuit(n_rows, 2, rng)
    g1 = O.G1.mul(O.G1.gen, rng.gen_fr()); g2 = O.G2.mul(O.G2.gen, rng.gen_fr())
    tox = [rng.gen_fr() for _ in range(5)]
    pk, toxic = G.generate_parameters(circ, g1, g2, *tox)
    r, s = rng.gen_fr(), rng.gen_fr()
    pa = G.prepare_assignment(circ)
    want = G.closed_form_proof(pa, toxic, r, s)
    t = time.time()
    proof = zinc_amd.create_proof(circ, zinc_amd.Parameters.from_points(pk), r, s, ctx)
    check("prove rows=%d (%.1f ms)" % (n_rows, 1e3 * (time.time() - t)), proof.as_tuple() == want)
    print(ctx.timings())
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
