"""CPU: host-side logic -- ABI surface, recorder, codecs, workload generator, sharding, gloo gather."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, unhex, FixtureCircuit, FixtureKey, fixture_proof
from zinc_amd import codec, lib as zlib
from zinc_amd.fields import R_MOD, ints_to_mont, mont_to_ints
from zinc_amd.groth16 import ProvingAssignment, AUX, random_fr
from zinc_amd.distributed import shard_range


def header_symbols():
    text = open(os.path.join(ROOT, "include", "zinc_groth16.h")).read()
    return sorted(set(re.findall(r"\b(zg16_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = header_symbols()
    assert len(names) >= 20
    lib = zlib.load()
    for n in names:
        assert hasattr(lib, n), "include/zinc_groth16.h declares %s but the library does not export it" % n
    bound = {n for n, _, _ in zlib.SYMBOLS}
    assert bound == set(names), "ctypes table and header disagree: %s" % (bound ^ set(names))


def test_header_compiles_as_c():
    src = '#include "zinc_groth16.h"\nint main(void){zg16_proof p; (void)p; return sizeof(zg16_partial) == 640 ? 0 : 1;}\n'
    exe = os.path.join(ROOT, "build", "hdr_check")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe],
                   input=src.encode(), check=True)
    assert subprocess.run([exe]).returncode == 0
    assert C.sizeof(zlib.Partial) == 640 and C.sizeof(zlib.Proof) == 256


def test_recorder_builds_canonical_csr():
    fx = golden("prove_tiny_2.json")
    circ = FixtureCircuit(fx)
    pa = ProvingAssignment()
    assert pa.alloc_input(1) == 0 == ProvingAssignment.one()
    circ.synthesize(pa)
    pa.finalize_inputs()
    r = pa.r1cs()
    n_rows = len(circ.rows[0]) + circ.n_in
    assert r.n_rows == n_rows and r.n_inputs == circ.n_in and r.n_aux == circ.n_aux
    rp, col, co = r.mats[0]
    # the appended rows are `input_i * 0 = 0`: one A term with coefficient one, empty B and C
    for i in range(circ.n_in):
        row = len(circ.rows[0]) + i
        assert rp[row + 1] - rp[row] == 1 and col[rp[row]] == i
        assert mont_to_ints(co[rp[row]], R_MOD) == [1]
    assert r.mats[1][0][-1] == sum(len(lc) for lc in circ.rows[1])
    # aux columns carry bit 31
    cols = [int(c) for c in col]
    assert any(c & AUX for c in cols) and all((c & ~AUX) < max(circ.n_aux, circ.n_in) for c in cols)


def off_subgroup_twist_point():
    """A point of the twist y^2 = x^3 + 3 / (9 + u) whose order is not r (the twist's cofactor is 2q - r)."""
    x = (5, 1)
    while True:
        rhs = codec._fq2_mul(codec._fq2_mul(x, x), x)
        y = codec._fq2_sqrt(((rhs[0] + codec.B_G2[0]) % codec.Q_MOD, (rhs[1] + codec.B_G2[1]) % codec.Q_MOD))
        if y is not None and not codec._g2_times_order_is_identity((x, y)):
            return x, y
        x = (x[0] + 1, 1)


@pytest.mark.parametrize("idx", [1, 2, 3])
def test_proof_codec_golden(idx):
    fx = golden("prove_tiny_%d.json" % idx)
    proof = fixture_proof(fx)
    data = codec.proof_write(proof)
    assert len(data) == 128 and data.hex() == fx["proof_bytes"]
    assert codec.proof_read(data) == proof
    with pytest.raises(codec.DecodeError):
        codec.proof_read(data[:-1])


def test_params_codec_roundtrip_and_checks():
    key = FixtureKey(golden("prove_tiny_1.json"))
    data = codec.params_write(key)
    back = codec.params_read(data, checked=True)
    for name in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g1", "ic", "h", "l", "a", "b_g1", "b_g2"):
        assert getattr(back, name) == getattr(key, name), name
    assert data[:len(codec.vk_write(key))] == codec.vk_write(key)
    # the product codec writes what the oracle's restatement of Parameters::write / VerifyingKey::write writes
    from oracle import codec as ocodec
    for idx in (1, 2, 3):
        k = FixtureKey(golden("prove_tiny_%d.json" % idx))
        assert codec.params_write(k) == ocodec.params_bytes(k) and codec.vk_write(k) == ocodec.vk_bytes(k)
    assert codec.params_read(ocodec.params_bytes(key), checked=True).b_g2 == key.b_g2
    # verifying-key points are validated whatever `checked` says, and ic may not hold the identity
    ba = bytearray(data)
    ba[63] ^= 1
    with pytest.raises(codec.DecodeError):
        codec.params_read(bytes(ba), checked=False)
    ic0 = 64 * 3 + 128 * 3 + 4
    ba = bytearray(data)
    ba[ic0:ic0 + 64] = bytes([0x40]) + bytes(63)
    with pytest.raises(codec.DecodeError):
        codec.params_read(bytes(ba), checked=False)
    # a twist point outside the prime-order subgroup is refused, in the uncompressed form of a key file and in the
    # compressed form of a proof's B (pairing_ce's into_affine() checks both)
    x, y = off_subgroup_twist_point()
    with pytest.raises(codec.DecodeError):
        codec.g2_read(codec.g2_write((x, y), False), False, True)
    assert codec.g2_read(codec.g2_write((x, y), False), False, False) == (x, y)
    with pytest.raises(codec.DecodeError):
        codec.g2_read(codec.g2_write((x, y), True), True, True)
    a, _, c = fixture_proof(golden("prove_tiny_1.json"))
    with pytest.raises(codec.DecodeError):
        codec.proof_read(codec.proof_write((a, (x, y), c)))
    # bellman's Proof::read refuses the point at infinity in any of the three places
    with pytest.raises(codec.DecodeError):
        codec.proof_read(codec.proof_write((a, None, c)))
    # infinity in a query vector is rejected when checked, accepted otherwise
    key.h[0] = None
    bad = codec.params_write(key)
    with pytest.raises(codec.DecodeError):
        codec.params_read(bad, checked=True)
    assert codec.params_read(bad, checked=False).h[0] is None
    # an off-curve point
    ba = bytearray(data)
    ba[63] ^= 1
    with pytest.raises(codec.DecodeError):
        codec.params_read(bytes(ba), checked=True)
    assert codec.g1_read(codec.g1_write(None, True), True) is None
    assert codec.g2_read(codec.g2_write(None, False), False) is None


def test_random_fr_follows_reference_rng():
    from oracle.xorshift import XorShiftRng
    fx = golden("rng_xorshift.json")
    rng = XorShiftRng(tuple(fx["seed"]))
    want = [int(x, 16) * pow(1 << 256, -1, R_MOD) % R_MOD for x in fx["fr_first8"]]
    assert [random_fr(rng) for _ in range(8)] == want
    assert 0 <= random_fr() < R_MOD


def test_synthetic_workload_is_satisfied_and_canonical():
    from zinc_amd.workload import SyntheticCircuit
    for dense in (False, True):
        c = SyntheticCircuit(1 << 12, n_pub=16, dense=dense)
        assert c.n_rows == 4096 and c.n_inputs == 17 and c.satisfied()
        r = c.r1cs()
        for rp, col, co in r.mats:
            assert rp[0] == 0 and np.all(np.diff(rp.astype(np.int64)) >= 0) and rp[-1] == col.shape[0]
            assert co.any(axis=1).all(), "zero coefficient emitted"
            for row in range(0, r.n_rows, 97):  # inputs first then aux, strictly ascending, no repeats
                cols = col[rp[row]:rp[row + 1]].astype(np.int64)
                assert np.all(np.diff(cols) > 0)
        zi, za = c.witness()
        assert mont_to_ints(zi[:1], R_MOD) == [1]
        bits = sum(1 for v in mont_to_ints(za[:2000], R_MOD) if v in (0, 1))
        assert (bits < 50) if dense else (bits > 1200)
    # same seed, same circuit
    a, b = SyntheticCircuit(1 << 10), SyntheticCircuit(1 << 10)
    assert np.array_equal(a.witness()[1], b.witness()[1])
    assert not np.array_equal(a.witness()[1], SyntheticCircuit(1 << 10, seed=(9, 8, 7, 6)).witness()[1])


def test_shard_ranges_partition_every_query():
    for n in (0, 1, 7, 1000, (1 << 20) - 1):
        for cnt in (1, 2, 3, 8):
            spans = [shard_range(n, i, cnt) for i in range(cnt)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(cnt - 1))


def test_two_rank_gloo_gather_and_finish():
    """world_size 2 over gloo: each rank sums its half of every query (on the CPU here), the partials
    are all-gathered and rank 0 assembles the proof -- must equal the single-process proof."""
    script = os.path.join(ROOT, "tests", "gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "PROOF_MATCH" in outs[0], outs[0]


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_sharded_transform_schedule(world):
    """The one-exchange sharded transform over gloo ranks on the CPU: the product's schedule, all-to-all
    and index maps around CPU stand-ins for the per-rank kernels equal the oracle's whole-vector FFT."""
    script = os.path.join(ROOT, "tests", "gloo_ntt_worker.py")
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(29560 + world), "8"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "NTT_SHARDED_MATCH" in outs[0], outs[0]


def test_fp29_bounds_hold():
    """The lazy 29-bit-limb arithmetic of the multiexp kernels never overflows a column sum, never
    borrows in a limb-wise subtraction and keeps every product below 2p (worst-case propagation)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_fp29_bounds.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all bounds hold" in out.stdout


def test_fp29_constants_in_sync(tmp_path):
    """fp29_consts.inc is what tools/gen_fp29_consts.py generates, and the constants are what they claim."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fresh = tmp_path / "consts.inc"
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_fp29_consts.py"), str(fresh)], check=True)
    committed = open(os.path.join(root, "zinc_amd", "csrc", "fp29_consts.inc")).read()
    assert fresh.read_text() == committed
    import re
    q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    block = committed.split("struct Fr29Params")[0]
    arrays = {m.group(1): [int(x.rstrip("u"), 16) for x in m.group(2).split(",")]
              for m in re.finditer(r"uint32_t (\w+)\[9\] = \{([^}]*)\}", block)}
    val = lambda l: sum(v << (29 * i) for i, v in enumerate(l))
    assert val(arrays["MOD"]) == q
    assert val(arrays["ONE"]) == (1 << 261) % q
    for name, l in arrays.items():
        if name.startswith("K"):
            k = int(name.lstrip("KS"))
            assert val(l) == k * q, name
            lend = 30 if name.startswith("KS") else 31
            assert all(v >= (1 << lend) - (1 << (lend - 29)) for v in l[:8]) and all(0 <= v < 1 << 32 for v in l)


def test_proof_encode_matches_codec_and_fixtures():
    """zg16_proof_encode (C, host-only) == the Python codec == the proof bytes recorded in the fixtures."""
    from zinc_amd.fields import g1_to_mont, g2_to_mont
    L = zlib.load()
    for idx in (1, 2, 3):
        fx = golden("prove_tiny_%d.json" % idx)
        a, b, c = fixture_proof(fx)
        raw = zlib.Proof()
        C.memmove(raw.a, g1_to_mont([a]).ctypes.data, 64)
        C.memmove(raw.b, g2_to_mont([b]).ctypes.data, 128)
        C.memmove(raw.c, g1_to_mont([c]).ctypes.data, 64)
        out = (C.c_uint8 * 128)()
        assert L.zg16_proof_encode(C.byref(raw), out) == 0
        assert bytes(out).hex() == fx["proof_bytes"] == codec.proof_write((a, b, c)).hex()
    # negated points flip the sign bit; the identity sets bit 6
    fx = golden("prove_tiny_1.json")
    a, b, c = fixture_proof(fx)
    from zinc_amd.fields import Q_MOD
    na = (a[0], (-a[1]) % Q_MOD)
    nb = (b[0], ((-b[1][0]) % Q_MOD, (-b[1][1]) % Q_MOD))
    raw = zlib.Proof()
    C.memmove(raw.a, g1_to_mont([na]).ctypes.data, 64)
    C.memmove(raw.b, g2_to_mont([nb]).ctypes.data, 128)
    out = (C.c_uint8 * 128)()
    assert L.zg16_proof_encode(C.byref(raw), out) == 0
    assert bytes(out).hex() == codec.proof_write((na, nb, None)).hex()


@pytest.mark.parametrize("n", [5, 6, 7, 0, 200])
def test_factorial_stand_in_rows(n):
    """The hand-recorded factorial.zn circuit: the output is the reference's pinned one
    (zinc-tester/tests/algorithm/factorial.zn: 5 -> 120, 6 -> 720, 7 -> 5040), every row is satisfied, and every
    linear combination is in zinc-vm's canonical form (constraint_systems/dedup.rs:30-69)."""
    from zinc_amd import standin
    from zinc_amd.fields import R_MOD
    import math
    circ = standin.FactorialCircuit(n)
    pa = standin.record(circ)
    want = standin.FactorialCircuit.CASES.get(n, math.factorial(min(n, 7)))
    assert circ.result == want and pa.z_in == [1, want]
    z = lambda v: pa.z_aux[v & 0x7fffffff] if v & (1 << 31) else pa.z_in[v]
    n_rows = len(pa._rp[0]) - 1
    assert 1500 < n_rows < 2048    # a factorial.zn-sized circuit: domain 2^11
    for r in range(n_rows):
        vals = []
        for k in range(3):
            lo, hi = pa._rp[k][r], pa._rp[k][r + 1]
            cols = pa._col[k][lo:hi]
            key = [((c >> 31), c & 0x7fffffff) for c in cols]
            assert key == sorted(set(key)), "row %d matrix %d is not canonical" % (r, k)
            assert all(pa._coeff[k][lo:hi])
            vals.append(sum(co * z(c) for c, co in zip(cols, pa._coeff[k][lo:hi])) % R_MOD)
        assert vals[0] * vals[1] % R_MOD == vals[2], "row %d is not satisfied" % r
    # every aux variable occurs in some row (bellman's setup refuses the circuit otherwise)
    seen = {c for k in range(3) for c in pa._col[k] if c & (1 << 31)}
    assert len(seen) == len(pa.z_aux)
    # the matrices do not depend on the input: that is what lets a prover keep them resident per (bytecode, method)
    other = standin.record(standin.FactorialCircuit(3))
    assert other._rp == pa._rp and other._col == pa._col and other._coeff == pa._coeff


@pytest.mark.parametrize("idx", [1, 2, 3])
def test_verify_proof_golden(idx):
    """zg16_verify_proof (host pairing arithmetic in the library) accepts the fixture proofs the oracle's verifier
    accepted, and rejects a changed input, a changed proof point and a wrong number of inputs."""
    from conftest import golden, FixtureKey, fixture_proof
    from zinc_amd import lib as zlib
    from zinc_amd.fields import g1_to_mont, g2_to_mont
    from zinc_amd.groth16 import SynthesisError, verify_proof
    import time
    fx = golden("prove_tiny_%d.json" % idx)
    assert fx["pairing_check"] is True
    key = FixtureKey(fx)

    class VK:
        alpha_g1, beta_g2 = g1_to_mont([key.alpha_g1]), g2_to_mont([key.beta_g2])
        gamma_g2, delta_g2 = g2_to_mont([key.gamma_g2]), g2_to_mont([key.delta_g2])
        ic = g1_to_mont(key.ic)

    a, b, c = fixture_proof(fx)
    raw = zlib.Proof()
    raw.a[:] = [int(x) for x in g1_to_mont([a])[0]]
    raw.b[:] = [int(x) for x in g2_to_mont([b])[0]]
    raw.c[:] = [int(x) for x in g1_to_mont([c])[0]]
    public = [int(x, 16) for x in fx["public_inputs"]]
    t0 = time.time()
    assert verify_proof(VK, raw, public)
    assert time.time() - t0 < 2.0
    assert not verify_proof(VK, raw, [public[0] + 1] + public[1:])
    bad = zlib.Proof.from_buffer_copy(bytes(raw))
    bad.a[:], bad.c[:] = list(raw.c), list(raw.a)     # two valid curve points in the wrong places
    assert not verify_proof(VK, bad, public)
    bad = zlib.Proof.from_buffer_copy(bytes(raw))
    bad.c[0] ^= 1                                      # not on the curve
    assert not verify_proof(VK, bad, public)
    with pytest.raises(SynthesisError) as e:
        verify_proof(VK, raw, public + [1])
    assert e.value.status_name == "MalformedKey"
    # B is the one untrusted G2 point: on the twist but outside the prime-order subgroup, off the twist, and limbs
    # that are not a residue below q (in A, B or C) are all "not a proof" -- bellman's Proof::read refuses them
    bad = zlib.Proof.from_buffer_copy(bytes(raw))
    bad.b[:] = [int(x) for x in g2_to_mont([off_subgroup_twist_point()])[0]]
    assert not verify_proof(VK, bad, public)
    bad = zlib.Proof.from_buffer_copy(bytes(raw))
    bad.b[5] ^= 1
    assert not verify_proof(VK, bad, public)
    from zinc_amd.fields import Q_MOD
    for field, limb0 in (("a", 0), ("a", 4), ("b", 8), ("c", 4)):
        bad = zlib.Proof.from_buffer_copy(bytes(raw))
        arr = getattr(bad, field)
        val = sum(int(arr[limb0 + i]) << (64 * i) for i in range(4)) + Q_MOD   # the same residue, not canonical
        if val >> 256:
            continue
        for i in range(4):
            arr[limb0 + i] = (val >> (64 * i)) & ((1 << 64) - 1)
        assert not verify_proof(VK, bad, public)
    # the identity in B (an in-memory proof may hold it; e(A, O) = 1): handled, not a crash
    bad = zlib.Proof.from_buffer_copy(bytes(raw))
    bad.b[:] = [0] * 16
    assert not verify_proof(VK, bad, public)


def test_verify_fast_pairing_equals_plain_pairing():
    """The pairing that runs (twist arithmetic, one shared Miller loop, Frobenius constants, x-power final
    exponentiation) against the plain one kept beside it (E(Fq12) arithmetic, power by (q^12 - 1) / r):
    ZG16_VERIFY_CHECK=1 makes zg16_verify_proof compute both and fail unless the Miller products and the
    exponentiations agree element for element -- on the accepted fixture proofs and on every rejected variant of
    test_verify_proof_golden.  The switch is read once per process, hence the child process."""
    import subprocess
    import sys
    env = dict(os.environ, ZG16_VERIFY_CHECK="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__), "-k",
                        "test_verify_proof_golden"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "3 passed" in r.stdout


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_sharded_h_schedule(world):
    """The three-round schedule that keeps a, b, c sharded through the whole H computation, over gloo ranks on the
    CPU: the product's stage / exchange loop around big-integer stand-ins for the stages gives every rank its
    cyclic slice of the oracle's H coefficients."""
    script = os.path.join(ROOT, "tests", "gloo_hshard_worker.py")
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(29570 + world), "6"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "HSHARD_MATCH" in outs[0], outs[0]


def test_key_file_encoder_matches_codec():
    """workload.params_file_bytes (the threaded encoder benchmarks make 0.4 .. 6 GB key files with) writes what the
    point-by-point codec -- and the oracle's restatement of Parameters::write -- writes."""
    from oracle import codec as ocodec
    from zinc_amd import workload
    from zinc_amd.fields import g1_to_mont, g2_to_mont
    from zinc_amd.groth16 import Parameters
    for idx in (1, 2, 3):
        k = FixtureKey(golden("prove_tiny_%d.json" % idx))
        p = Parameters.from_points(k)
        p.gamma_g2, p.ic = g2_to_mont([k.gamma_g2]), g1_to_mont(k.ic)
        data = bytes(workload.params_file_bytes(p, threads=3))
        assert data == codec.params_write(k) == ocodec.params_bytes(k)
    k.h[1] = None   # the identity inside a vector: 0x40 and zeros
    p = Parameters.from_points(k)
    p.gamma_g2, p.ic = g2_to_mont([k.gamma_g2]), g1_to_mont(k.ic)
    assert bytes(workload.params_file_bytes(p)) == codec.params_write(k)



@pytest.mark.parametrize("name", ["fibonacci", "division", "uniswap", "factorial_contract"])
def test_more_stand_in_rows(name):
    """The other hand-recorded programs (zinc-tester's fibonacci.zn and division_u8.zn, the uniswap example's
    get_eth_to_token_input_price, factorial.zn wrapped as a contract method): the outputs are the reference's pinned
    ones, every row is satisfied and canonical (constraint_systems/dedup.rs:30-69), every aux variable occurs in a
    row, and the matrices do not depend on the inputs."""
    from zinc_amd import standin
    from zinc_amd.fields import R_MOD
    S = standin
    cases = {
        "fibonacci": [(S.FibonacciCircuit(n), [w]) for n, w in S.FibonacciCircuit.CASES.items()],
        "division": [(S.DivisionU8Circuit(a, b), [w]) for (a, b), w in S.DivisionU8Circuit.CASES.items()],
        "uniswap": [(S.UniswapInputPriceCircuit(e, b, t, 77), [S.UniswapInputPriceCircuit.price(e, b, t), 77])
                    for e, b, t in ((10 ** 18, 5 * 10 ** 21, 7 * 10 ** 23), (1, 1, 1), (3 * 10 ** 30, 10 ** 9, 10 ** 40))],
        "factorial_contract": [(S.FactorialCircuit(6, storage_root=5), [720, 5])],
    }[name]
    shapes = set()
    for circ, want in cases:
        pa = S.record(circ)
        assert circ.public_inputs() == want and pa.z_in == [1] + want
        z = lambda v: pa.z_aux[v & 0x7fffffff] if v & (1 << 31) else pa.z_in[v]
        n_rows = len(pa._rp[0]) - 1
        for r in range(n_rows):
            vals = []
            for k in range(3):
                lo, hi = pa._rp[k][r], pa._rp[k][r + 1]
                cols = pa._col[k][lo:hi]
                key = [((c >> 31), c & 0x7fffffff) for c in cols]
                assert key == sorted(set(key)), "row %d matrix %d is not canonical" % (r, k)
                assert all(pa._coeff[k][lo:hi])
                vals.append(sum(co * z(c) for c, co in zip(cols, pa._coeff[k][lo:hi])) % R_MOD)
            assert vals[0] * vals[1] % R_MOD == vals[2], "row %d is not satisfied" % r
        seen = {c for k in range(3) for c in pa._col[k] if c & (1 << 31)}
        assert len(seen) == len(pa.z_aux)
        shapes.add((tuple(map(tuple, pa._rp)), tuple(map(tuple, pa._col)), tuple(map(tuple, pa._coeff))))
    assert len(shapes) == 1, "the matrices depend on the inputs"
    # division by zero and a failed require are refused at synthesis, as the reference's VM refuses them
    if name == "division":
        with pytest.raises(AssertionError):
            S.record(S.DivisionU8Circuit(42, 0))
    if name == "uniswap":
        with pytest.raises(AssertionError):
            S.record(S.UniswapInputPriceCircuit(0, 1, 1, 77))


def test_prove_finish_assembly_against_oracle_for_edge_randomness():
    """zg16_prove_finish (host arithmetic only) on made-up query sums, against the big-integer oracle's
    A = alpha + r delta + a, B = beta + s delta + b (G2), C = s A + r (beta_g1 + b_g1) + hl -- for r, s at the edges of
    what the scalar split by the curve's endomorphism must handle (0, 1, r - 1, lambda and its negative, values
    with a split half at its largest) and random full-size ones."""
    import random
    from oracle.bn254 import G1, G2
    from zinc_amd.fields import Q_MOD, g1_to_mont, g2_to_mont, mont_to_g1, mont_to_g2
    rng = random.Random(2718)
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd

    def g1(k):
        return G1.to_affine(G1.mul(G1.gen, k))

    def g2(k):
        return G2.to_affine(G2.mul(G2.gen, k))

    def xyzz(aff, width):
        out = np.zeros(2 * width, dtype=np.uint64)
        if aff.any():
            out[:width] = aff
            one = ints_to_mont([1], Q_MOD)[0]
            out[width:width + 4] = one
            out[width + width // 2: width + width // 2 + 4] = one
        return out

    key = {"alpha_g1": g1(11), "beta_g1": g1(13), "delta_g1": g1(17), "beta_g2": g2(13), "delta_g2": g2(17)}
    bufs = {n: (g1_to_mont([p])[0] if n.endswith("g1") else g2_to_mont([p])[0]) for n, p in key.items()}
    view = zlib.PkView()
    for n, b in bufs.items():
        setattr(view, n, zlib.as_u64p(b))
    sums = {"hl": g1(rng.randrange(R_MOD)), "a": g1(rng.randrange(R_MOD)), "b1": g1(rng.randrange(R_MOD)), "b2": g2(rng.randrange(R_MOD))}
    part = zlib.Partial()
    for name, width in (("hl", 8), ("a", 8), ("b1", 8), ("b2", 16)):
        aff = g1_to_mont([sums[name]])[0] if width == 8 else g2_to_mont([sums[name]])[0]
        C.memmove(getattr(part, name), xyzz(aff, width).ctypes.data, 16 * width)
    parts = (zlib.Partial * 1)(part)
    edge = [0, 1, 2, R_MOD - 1, lam, R_MOD - lam, lam + 1, (1 << 127) - 1, 1 << 127, (1 << 128) - 1, (lam << 1) % R_MOD,
            (lam * ((1 << 126) + 12345) + (1 << 126) + 999) % R_MOD]
    cases = [(r, s) for r in edge[:6] for s in edge[:6]] + [(edge[i], edge[-1 - i]) for i in range(len(edge))]
    cases += [(rng.randrange(R_MOD), rng.randrange(R_MOD)) for _ in range(24)]
    L = zlib.load()
    for r, s in cases:
        rs = ints_to_mont([r, s], R_MOD)
        proof = zlib.Proof()
        assert L.zg16_prove_finish(C.byref(view), parts, 1, zlib.as_u64p(rs[0]), zlib.as_u64p(rs[1]), C.byref(proof)) == 0
        A = G1.add(G1.add(G1.from_affine(key["alpha_g1"]), G1.mul(G1.from_affine(key["delta_g1"]), r)), G1.from_affine(sums["a"]))
        B = G2.add(G2.add(G2.from_affine(key["beta_g2"]), G2.mul(G2.from_affine(key["delta_g2"]), s)), G2.from_affine(sums["b2"]))
        b1 = G1.add(G1.from_affine(key["beta_g1"]), G1.from_affine(sums["b1"]))
        Cp = G1.add(G1.add(G1.mul(A, s), G1.mul(b1, r)), G1.from_affine(sums["hl"]))
        got = (mont_to_g1(np.frombuffer(bytes(proof.a), dtype=np.uint64))[0],
               mont_to_g2(np.frombuffer(bytes(proof.b), dtype=np.uint64))[0],
               mont_to_g1(np.frombuffer(bytes(proof.c), dtype=np.uint64))[0])
        assert got == (G1.to_affine(A), G2.to_affine(B), G1.to_affine(Cp)), "assembly differs for r = %x, s = %x" % (r, s)
