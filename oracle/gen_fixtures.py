"""Generate the small golden fixtures under tests/golden/ from the big-integer oracle.

    python -m oracle.gen_fixtures            # rewrites tests/golden/*.json

TEST INFRASTRUCTURE ONLY.  The reference ships no vectors for the prove path (its only check is
`verify_proof == true`, zinc-tester/src/runners/proof_check.rs:243-264), so these are produced by
this repository's own oracle ("parity unpinned" against bellman bytes) and every prove fixture
records that the Groth16 pairing equation holds and that the closed-form derivation from the
toxic waste gives the same points.  All inputs are drawn from rand 0.4's XorShiftRng with the
seed the reference's tests use (zinc-vm/src/gadgets/contract/storage.rs:219).
Numbers are canonical (non-Montgomery) integers written as hex strings.
"""
import json
import os

from . import bn254 as O
from . import groth16 as G
from .codec import proof_bytes
from .synth import make_circuit
from .xorshift import XorShiftRng, REFERENCE_SEED

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
H = lambda x: "%x" % x


def hx(v):
    if v is None:
        return None
    if isinstance(v, int):
        return H(v)
    return [hx(x) for x in v]


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name)


def field_ops(p, gen, rng_draw):
    R = O.MONT_R % p
    edge = [0, 1, p - 1, R, R * R % p, 2, p - 2, (p + 1) // 2]
    vals = edge + [rng_draw() for _ in range(64 - len(edge))]
    cases = []
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        cases.append(dict(a=H(a), b=H(b), add=H((a + b) % p), sub=H((a - b) % p), mul=H(a * b % p),
                          neg=H(-a % p), sqr=H(a * a % p), inv=H(pow(a, -1, p)) if a else None))
    return dict(modulus=H(p), cases=cases)


def group_ops(Gp, rng, scalar_rng):
    g = Gp.gen
    pts = [Gp.mul(g, scalar_rng()) for _ in range(6)]
    aff = Gp.batch_to_affine(pts)
    cases = []
    for i, P in enumerate(pts):
        Q = pts[(i + 1) % len(pts)]
        cases.append(dict(p=hx(aff[i]), q=hx(Gp.to_affine(Q)),
                          dbl=hx(Gp.to_affine(Gp.double(P))), add=hx(Gp.to_affine(Gp.add(P, Q))),
                          add_self=hx(Gp.to_affine(Gp.add(P, P))), add_neg=hx(Gp.to_affine(Gp.add(P, Gp.neg(P)))),
                          add_identity=hx(Gp.to_affine(Gp.add(P, Gp.identity)))))
    k = scalar_rng()
    smul = [dict(k=H(x), out=hx(Gp.to_affine(Gp.mul(pts[0], x)))) for x in (0, 1, 2, O.R_MOD - 1, k)]
    return dict(generator=hx(Gp.gen_affine), cases=cases, scalar_mul=dict(p=hx(aff[0]), cases=smul))


def ntt_fixture(log_n, rng):
    n = 1 << log_n
    x = [rng.gen_fr() for _ in range(n)]
    out = dict(log_n=log_n, input=hx(x))
    d = G.EvaluationDomain(x)
    out["omega"] = H(d.omega)
    if log_n <= 5:
        assert G.naive_dft(x, d.omega) == G.serial_fft(x, d.omega, log_n)
    for name in ("fft", "ifft", "coset_fft", "icoset_fft"):
        d.coeffs = list(x)
        getattr(d, name)()
        out[name] = hx(d.coeffs)
    return out


def scalars_mix(rng, n):
    out = []
    for _ in range(n):
        k = rng.below(10)
        out.append(0 if k == 0 else 1 if k < 4 else O.R_MOD - 1 if k == 4 else rng.below(1 << 16) if k == 5 else rng.gen_fr())
    return out


def msm_fixture(Gp, log_n, rng):
    n = 1 << log_n
    logs = [rng.below(1 << 40) + 1 for _ in range(n)]
    logs[n // 2] = logs[1]  # a repeated base
    pts = Gp.batch_to_affine([Gp.mul(Gp.gen, k) for k in logs])
    sc = scalars_mix(rng, n)
    dens = [rng.below(4) != 0 for _ in range(n)]
    full = Gp.to_affine(Gp.mul(Gp.gen, sum(k * s for k, s in zip(logs, sc)) % O.R_MOD))
    dsum = Gp.to_affine(Gp.mul(Gp.gen, sum(k * s for k, s, d in zip(logs, sc, dens) if d) % O.R_MOD))
    if log_n <= 8:  # the bellman-style windowed path agrees with the discrete-log shortcut
        assert Gp.to_affine(G.multiexp(Gp, pts, sc)) == full
        assert Gp.to_affine(G.multiexp(Gp, [p for p, d in zip(pts, dens) if d], sc, dens)) == dsum
    return dict(bases=hx(pts), scalars=hx(sc), density=[int(d) for d in dens], result=hx(full), result_density=hx(dsum))


def lc_json(rows):
    return [[[col, H(c)] for col, c in lc] for lc in rows]


def prove_fixture(n_rows, n_pub, rng):
    circ = make_circuit(n_rows, n_pub, rng)
    k1, k2 = rng.gen_fr(), rng.gen_fr()
    tox = dict(alpha=rng.gen_fr(), beta=rng.gen_fr(), gamma=rng.gen_fr(), delta=rng.gen_fr(), tau=rng.gen_fr())
    pk, toxic = G.generate_parameters(circ, O.G1.mul(O.G1.gen, k1), O.G2.mul(O.G2.gen, k2), tox["alpha"], tox["beta"],
                                      tox["gamma"], tox["delta"], tox["tau"])
    r, s = rng.gen_fr(), rng.gen_fr()
    pa = G.prepare_assignment(circ)
    proof = G.create_proof_from_assignment(pa, pk, r, s)
    assert proof == G.closed_form_proof(pa, toxic, r, s)
    assert proof == G.create_proof_from_assignment(pa, pk, r, s, window=4)
    assert G.verify_proof(pk, proof, circ.z_in[1:])
    m, h = G.h_coefficients(pa.a, pa.b, pa.c)
    return dict(
        n_inputs=circ.n_in, n_aux=circ.n_aux,
        rows_a=lc_json(circ.rows_a), rows_b=lc_json(circ.rows_b), rows_c=lc_json(circ.rows_c),
        z_in=hx(circ.z_in), z_aux=hx(circ.z_aux),
        eval_a=hx(pa.a), eval_b=hx(pa.b), eval_c=hx(pa.c),
        a_aux_density=[int(x) for x in pa.a_aux_density], b_input_density=[int(x) for x in pa.b_input_density],
        b_aux_density=[int(x) for x in pa.b_aux_density],
        m=m, h=hx(h),
        toxic=dict(k1=H(k1), k2=H(k2), **{k: H(v) for k, v in tox.items()}),
        key=dict(alpha_g1=hx(pk.alpha_g1), beta_g1=hx(pk.beta_g1), beta_g2=hx(pk.beta_g2), gamma_g2=hx(pk.gamma_g2),
                 delta_g1=hx(pk.delta_g1), delta_g2=hx(pk.delta_g2), ic=hx(pk.ic), h=hx(pk.h), l=hx(pk.l), a=hx(pk.a),
                 b_g1=hx(pk.b_g1), b_g2=hx(pk.b_g2)),
        r=H(r), s=H(s), proof=dict(a=hx(proof[0]), b=hx(proof[1]), c=hx(proof[2])),
        proof_bytes=proof_bytes(proof).hex(), public_inputs=hx(circ.z_in[1:]), pairing_check=True,
        closed_form_agrees=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = XorShiftRng()
    r2 = XorShiftRng()
    dump("rng_xorshift.json", dict(seed=list(REFERENCE_SEED), u32=[r2.next_u32() for _ in range(64)],
                                   fr_raw=[H(XorShiftRng().gen_fr_raw())] + [], fr_first8=(lambda g: [H(g.gen_fr_raw()) for _ in range(8)])(XorShiftRng())))
    dump("fr_ops.json", field_ops(O.R_MOD, 7, rng.gen_fr))
    dump("fq_ops.json", field_ops(O.Q_MOD, 3, rng.gen_fq))
    vals = [(rng.gen_fq(), rng.gen_fq()) for _ in range(24)] + [(0, 0), (1, 0), (0, 1), (O.Q_MOD - 1, O.Q_MOD - 1)]
    dump("fq2_ops.json", dict(cases=[dict(a=hx(a), b=hx(vals[(i * 5 + 1) % len(vals)]),
                                          add=hx(O.fq2_add(a, vals[(i * 5 + 1) % len(vals)])),
                                          sub=hx(O.fq2_sub(a, vals[(i * 5 + 1) % len(vals)])),
                                          mul=hx(O.fq2_mul(a, vals[(i * 5 + 1) % len(vals)])),
                                          sqr=hx(O.fq2_sqr(a)), neg=hx(O.fq2_neg(a)),
                                          mul_nonresidue=hx(O.fq2_mul(a, (9, 1))),
                                          inv=hx(O.fq2_inv(a)) if a != (0, 0) else None) for i, a in enumerate(vals)]))
    dump("g1_ops.json", group_ops(O.G1, rng, rng.gen_fr))
    dump("g2_ops.json", group_ops(O.G2, rng, rng.gen_fr))
    for log_n in (4, 5, 6, 7, 8, 9, 10):
        dump("ntt_%d.json" % log_n, ntt_fixture(log_n, rng))
    for n_rows in (3, 6):
        a = [rng.gen_fr() for _ in range(n_rows)]
        b = [rng.gen_fr() for _ in range(n_rows)]
        c = [x * y % O.R_MOD for x, y in zip(a, b)]
        m, h = G.h_coefficients(a, b, c)
        dump("h_poly_%d.json" % n_rows, dict(a=hx(a), b=hx(b), c=hx(c), m=m, h=hx(h)))
    for log_n in (4, 8, 12):
        dump("msm_g1_%d.json" % log_n, msm_fixture(O.G1, log_n, rng))
    for log_n in (4, 8):
        dump("msm_g2_%d.json" % log_n, msm_fixture(O.G2, log_n, rng))
    for i, (rows, pub) in enumerate(((4, 1), (24, 2), (70, 3)), 1):
        dump("prove_tiny_%d.json" % i, prove_fixture(rows, pub, rng))


if __name__ == "__main__":
    main()
