"""CPU: the oracles against public BN254 constants, each other, and the committed golden vectors."""
import numpy as np
import pytest

from conftest import golden, unhex, FixtureCircuit, FixtureKey, fixture_proof
from oracle import bn254 as O
from oracle import groth16 as G
from oracle import cpu_binding as cb
from oracle.xorshift import XorShiftRng
from zinc_amd.fields import (R_MOD, Q_MOD, ints_to_mont, mont_to_ints, g1_to_mont, g2_to_mont, mont_to_g1, mont_to_g2)
from zinc_amd.groth16 import ProvingAssignment, Parameters


def test_public_constants():
    assert O.R_MOD.bit_length() == 254 and O.Q_MOD.bit_length() == 254
    assert O.FR_ROOT_OF_UNITY == 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
    assert pow(O.FR_ROOT_OF_UNITY, 1 << 28, O.R_MOD) == 1 != pow(O.FR_ROOT_OF_UNITY, 1 << 27, O.R_MOD)
    assert O.FR_INV64 == 0xc2e1f593efffffff and O.FQ_INV64 == 0x87d20782e4866389
    assert O.G1.on_curve_affine(O.G1_GEN) and O.G2.on_curve_affine(O.G2_GEN)
    assert O.G1.is_identity(O.G1.mul(O.G1.gen, O.R_MOD)) and O.G2.is_identity(O.G2.mul(O.G2.gen, O.R_MOD))


def test_pairing_is_bilinear_and_nondegenerate():
    a, b = 0x1234567, 0x89abcd
    e1 = O.pairing(O.G1.to_affine(O.G1.mul(O.G1.gen, a)), O.G2.to_affine(O.G2.mul(O.G2.gen, b)))
    e2 = O._f12_pow(O.pairing(O.G1_GEN, O.G2_GEN), a * b)
    assert e1 == e2 and e1 != O.F12_ONE


def test_xorshift_fixture():
    fx = golden("rng_xorshift.json")
    rng = XorShiftRng(tuple(fx["seed"]))
    assert [rng.next_u32() for _ in range(64)] == fx["u32"]
    rng = XorShiftRng(tuple(fx["seed"]))
    assert [rng.gen_fr_raw() for _ in range(8)] == [int(x, 16) for x in fx["fr_first8"]]
    assert all(int(x, 16) < R_MOD for x in fx["fr_first8"])


@pytest.mark.parametrize("log_n", [4, 5, 6, 7, 8, 9, 10])
def test_fft_golden_python_and_cpp(log_n):
    fx = golden("ntt_%d.json" % log_n)
    x = [int(v, 16) for v in fx["input"]]
    d = G.EvaluationDomain(x)
    assert d.omega == int(fx["omega"], 16)
    for name, inv, coset in (("fft", 0, 0), ("ifft", 1, 0), ("coset_fft", 0, 1), ("icoset_fft", 1, 1)):
        want = [int(v, 16) for v in fx[name]]
        d.coeffs = list(x)
        getattr(d, name)()
        assert d.coeffs == want
        for threads in (1, 4):
            assert mont_to_ints(cb.fft(ints_to_mont(x, R_MOD), inv, coset, threads), R_MOD) == want


def test_naive_dft_agrees():
    rng = XorShiftRng()
    x = [rng.gen_fr() for _ in range(16)]
    d = G.EvaluationDomain(x)
    assert G.naive_dft(x, d.omega) == G.serial_fft(x, d.omega, 4)


@pytest.mark.parametrize("rows", [3, 6])
def test_h_poly_golden(rows):
    fx = golden("h_poly_%d.json" % rows)
    a, b, c = ([int(v, 16) for v in fx[k]] for k in "abc")
    want = [int(v, 16) for v in fx["h"]]
    assert G.h_coefficients(a, b, c) == (fx["m"], want)
    h, m = cb.h_coefficients(*(ints_to_mont(v, R_MOD) for v in (a, b, c)), threads=3)
    assert m == fx["m"] and mont_to_ints(h, R_MOD) == want


@pytest.mark.parametrize("name,group", [("msm_g1_4", "g1"), ("msm_g1_8", "g1"), ("msm_g1_12", "g1"),
                                        ("msm_g2_4", "g2"), ("msm_g2_8", "g2")])
def test_multiexp_golden_cpp(name, group):
    fx = golden(name + ".json")
    to_m, from_m = (g1_to_mont, mont_to_g1) if group == "g1" else (g2_to_mont, mont_to_g2)
    bases = [unhex(p) for p in fx["bases"]]
    sc = ints_to_mont([int(v, 16) for v in fx["scalars"]], R_MOD)
    st, out = cb.multiexp(group, to_m(bases), sc, None, threads=4)
    assert st == 0 and from_m(out)[0] == unhex(fx["result"])
    dens = np.array(fx["density"], dtype=np.uint8)
    st, out = cb.multiexp(group, to_m([p for p, d in zip(bases, dens) if d]), sc, dens, threads=4)
    assert st == 0 and from_m(out)[0] == unhex(fx["result_density"])


def test_multiexp_identity_base_is_an_error():
    pts = g1_to_mont([O.G1_GEN, None, O.G1_GEN])
    st, _ = cb.multiexp("g1", pts, ints_to_mont([5, 7, 9], R_MOD))
    assert st == 2  # UnexpectedIdentity
    st, out = cb.multiexp("g1", pts, ints_to_mont([5, 0, 9], R_MOD))  # a zero scalar never touches its base
    assert st == 0 and mont_to_g1(out)[0] == O.G1.to_affine(O.G1.mul(O.G1.gen, 14))
    with pytest.raises(G.UnexpectedIdentity):
        G.multiexp(O.G1, [O.G1_GEN, None], [3, 4])


@pytest.mark.parametrize("idx", [1, 2, 3])
def test_prove_golden_python_and_cpp(idx):
    fx = golden("prove_tiny_%d.json" % idx)
    circ, key, want = FixtureCircuit(fx), FixtureKey(fx), fixture_proof(fx)
    r, s = int(fx["r"], 16), int(fx["s"], 16)
    # python oracle
    pa = G.ProvingAssignment()
    pa.alloc_input(1)
    circ.synthesize(pa)
    pa.finalize_inputs()
    assert pa.a == [int(v, 16) for v in fx["eval_a"]] and pa.b == [int(v, 16) for v in fx["eval_b"]]
    assert [int(x) for x in pa.a_aux_density] == fx["a_aux_density"]
    assert [int(x) for x in pa.b_input_density] == fx["b_input_density"]
    assert G.create_proof_from_assignment(pa, key, r, s) == want
    if idx == 1:
        assert G.verify_proof(key, want, [int(v, 16) for v in fx["public_inputs"]])
    # C++ restatement, through the product's recorder so the CSR layout is the ABI's
    rec = ProvingAssignment()
    rec.alloc_input(1)
    circ.synthesize(rec)
    rec.finalize_inputs()
    zi, za = rec.witness()
    (ea, eb, ec), (da, dbi, dba) = cb.eval_rows(rec.r1cs(), zi, za)
    assert mont_to_ints(ea, R_MOD) == pa.a and mont_to_ints(ec, R_MOD) == pa.c
    assert list(da) == fx["a_aux_density"] and list(dbi) == fx["b_input_density"] and list(dba) == fx["b_aux_density"]
    rs = ints_to_mont([r, s], R_MOD)
    for threads in (1, 5):
        st, (a, b, c), _ = cb.prove(Parameters.from_points(key), rec.r1cs(), zi, za, rs[0], rs[1], threads)
        assert st == 0
        assert (mont_to_g1(a)[0], mont_to_g2(b)[0], mont_to_g1(c)[0]) == want


def test_closed_form_matches_fft_multiexp_prover():
    from oracle.synth import make_circuit
    rng = XorShiftRng((1, 2, 3, 4))
    circ = make_circuit(12, 2, rng)
    pk, toxic = G.generate_parameters(circ, O.G1.mul(O.G1.gen, 11), O.G2.mul(O.G2.gen, 13), *[rng.gen_fr() for _ in range(5)])
    pa = G.prepare_assignment(circ)
    r, s = rng.gen_fr(), rng.gen_fr()
    proof = G.create_proof_from_assignment(pa, pk, r, s)
    assert proof == G.closed_form_proof(pa, toxic, r, s)
    assert proof == G.create_proof_from_assignment(pa, pk, r, s, window=7)
    assert G.verify_proof(pk, proof, circ.z_in[1:])
    assert not G.verify_proof(pk, proof, [(circ.z_in[1] + 1) % R_MOD] + circ.z_in[2:])


def test_domain_too_large():
    with pytest.raises(G.PolynomialDegreeTooLarge):
        class Huge(list):
            def __len__(self):
                return (1 << 27) + 1
        G.EvaluationDomain(Huge())
