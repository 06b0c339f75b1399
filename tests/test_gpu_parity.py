"""GPU (-m gpu): the HIP path, called through the C ABI, against golden vectors, the oracles on
seeded inputs, and size-independent properties at the benchmark's full sizes.  Bar: bit-exact
(all arithmetic is integer / modular; there is no floating point on this path)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import golden, unhex, FixtureCircuit, FixtureKey, fixture_proof
from zinc_amd import lib as zlib
from zinc_amd.fields import (R_MOD, Q_MOD, ints_to_mont, mont_to_ints, g1_to_mont, g2_to_mont, mont_to_g1, mont_to_g2)
from zinc_amd.groth16 import (ProvingAssignment, Parameters, SynthesisError, create_proof, prove_recorded, R1CS)

pytestmark = pytest.mark.gpu


def rand_fr(rng, n):
    """n canonical field elements as Montgomery limbs (top limb below 2^61 < the modulus' top limb)."""
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    x[:, 3] >>= np.uint64(3)
    return x


# ------------------------------------------------------------------------------ primitives
def _field(ctx, field, op, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros_like(a)
    n = a.shape[0]
    st = ctx._lib.zg16_selftest_field(ctx.handle, field, op, zlib.as_u64p(a), zlib.as_u64p(b), n, zlib.as_u64p(out))
    assert st == 0, ctx._lib.zg16_last_error(ctx.handle)
    return out


@pytest.mark.parametrize("name,field,mod", [("fr_ops", 0, R_MOD), ("fq_ops", 1, Q_MOD)])
def test_field_ops_golden(gpu_ctx, name, field, mod):
    fx = golden(name + ".json")
    a = ints_to_mont([int(c["a"], 16) for c in fx["cases"]], mod)
    b = ints_to_mont([int(c["b"], 16) for c in fx["cases"]], mod)
    for op, key in enumerate(("add", "sub", "mul", "sqr", "neg", "inv")):
        want = [int(c[key], 16) if c[key] is not None else 0 for c in fx["cases"]]
        assert mont_to_ints(_field(gpu_ctx, field, op, a, b), mod) == want, key


def test_fq2_ops_golden(gpu_ctx):
    fx = golden("fq2_ops.json")
    flat = lambda key: ints_to_mont([int(v, 16) for c in fx["cases"] for v in c[key]], Q_MOD).reshape(-1, 8)
    a, b = flat("a"), flat("b")
    for op, key in enumerate(("add", "sub", "mul", "sqr", "neg", "inv")):
        want = [tuple(int(v, 16) for v in c[key]) if c[key] is not None else (0, 0) for c in fx["cases"]]
        got = mont_to_ints(_field(gpu_ctx, 2, op, a, b).reshape(-1, 4), Q_MOD)
        assert list(zip(got[0::2], got[1::2])) == want, key
    nine_u = np.tile(ints_to_mont([9, 1], Q_MOD).reshape(1, 8), (a.shape[0], 1))
    got = mont_to_ints(_field(gpu_ctx, 2, 2, a, nine_u).reshape(-1, 4), Q_MOD)
    assert list(zip(got[0::2], got[1::2])) == [tuple(int(v, 16) for v in c["mul_nonresidue"]) for c in fx["cases"]]


# groups 1, 2: the saturated-limb code (key tables, assembly); 3, 4: the multiexp kernels' arithmetic
# (29-bit limbs, G2 on lane pairs) -- both pinned by the same vectors, exceptional cases included
@pytest.mark.parametrize("name,group", [("g1_ops", 1), ("g2_ops", 2), ("g1_ops", 3), ("g2_ops", 4)])
def test_group_ops_golden(gpu_ctx, name, group):
    fx = golden(name + ".json")
    to_m, from_m, w = (g1_to_mont, mont_to_g1, 8) if group in (1, 3) else (g2_to_mont, mont_to_g2, 16)

    def run(op, p, q):
        p = np.ascontiguousarray(p, dtype=np.uint64)
        q = np.ascontiguousarray(q, dtype=np.uint64)
        out = np.zeros_like(p)
        st = gpu_ctx._lib.zg16_selftest_group(gpu_ctx.handle, group, op, zlib.as_u64p(p), zlib.as_u64p(q), p.shape[0],
                                              zlib.as_u64p(out))
        assert st == 0
        return from_m(out)

    P = [unhex(c["p"]) for c in fx["cases"]]
    Q = [unhex(c["q"]) for c in fx["cases"]]
    neg = lambda p: (p[0], (-p[1]) % Q_MOD) if group in (1, 3) else (p[0], ((-p[1][0]) % Q_MOD, (-p[1][1]) % Q_MOD))
    assert run(0, to_m(P), to_m(P)) == [unhex(c["dbl"]) for c in fx["cases"]]
    # mixed and projective addition, including every exceptional case; 7: the addition spread over lanes that
    # the closing sums of a multiexp use
    for op in (1, 2) + ((7,) if group in (3, 4) else ()):
        assert run(op, to_m(P), to_m(Q)) == [unhex(c["add"]) for c in fx["cases"]]
        assert run(op, to_m(P), to_m(P)) == [unhex(c["add_self"]) for c in fx["cases"]]
        assert run(op, to_m(P), to_m([neg(p) for p in P])) == [None] * len(P)
        assert run(op, to_m(P), to_m([None] * len(P))) == P
        assert run(op, to_m([None] * len(P)), to_m(Q)) == Q
    if group in (3, 4):  # op 8: 2p + 2q through three chained additions of the spread form
        twice = lambda pts: run(0, to_m(pts), to_m(pts))
        for q in (Q, P, [neg(p) for p in P]):
            assert run(8, to_m(P), to_m(q)) == run(2, to_m(twice(P)), to_m(twice(q)))
    sm = fx["scalar_mul"]
    ks = ints_to_mont([int(c["k"], 16) for c in sm["cases"]], R_MOD)
    pts = to_m([unhex(sm["p"])] * len(sm["cases"]))
    assert run(3, pts, ks) == [unhex(c["out"]) for c in sm["cases"]]


def test_g2_subgroup_test_sides(gpu_ctx):
    """The subgroup test the key-file decoder runs on every b_g2 point, psi(P) = [6 x^2] P (psi the
    untwist-Frobenius-twist map, x the BN parameter), side by side in the lane-pair arithmetic: both sides equal the
    oracle's big-integer values on points of G2 (where they agree with each other) and on a twist point outside
    G2 (where they differ) -- the accept set is that of r P = O."""
    from oracle import bn254 as B
    from zinc_amd import codec
    fx = golden("g2_ops.json")
    pts = [unhex(c["p"]) for c in fx["cases"] if unhex(c["p"]) is not None]
    pts += [p for i in (1, 2, 3) for p in FixtureKey(golden("prove_tiny_%d.json" % i)).b_g2]
    x = (5, 1)
    while True:     # a point of the twist whose order is not r
        rhs = codec._fq2_mul(codec._fq2_mul(x, x), x)
        y = codec._fq2_sqrt(((rhs[0] + codec.B_G2[0]) % Q_MOD, (rhs[1] + codec.B_G2[1]) % Q_MOD))
        if y is not None and not codec._g2_times_order_is_identity((x, y)):
            break
        x = (x[0] + 1, 1)
    off = (x, y)
    pts.append(off)
    bn_x = 4965661367192848881
    k = 6 * bn_x * bn_x
    assert k == Q_MOD - R_MOD
    xi = (9, 1)
    kx, ky = codec._fq2_pow(xi, (Q_MOD - 1) // 3), codec._fq2_pow(xi, (Q_MOD - 1) // 2)
    conj = lambda a: (a[0], (-a[1]) % Q_MOD)
    psi = lambda p: (codec._fq2_mul(conj(p[0]), kx), codec._fq2_mul(conj(p[1]), ky))

    def run(op):
        arr = np.ascontiguousarray(g2_to_mont(pts), dtype=np.uint64)
        out = np.zeros_like(arr)
        st = gpu_ctx._lib.zg16_selftest_group(gpu_ctx.handle, 4, op, zlib.as_u64p(arr), None, arr.shape[0], zlib.as_u64p(out))
        assert st == 0
        return mont_to_g2(out)

    arr = np.ascontiguousarray(g2_to_mont(pts), dtype=np.uint64)
    verdict = np.zeros_like(arr)
    assert gpu_ctx._lib.zg16_selftest_group(gpu_ctx.handle, 4, 6, zlib.as_u64p(arr), None, arr.shape[0], zlib.as_u64p(verdict)) == 0
    assert [int(v) for v in verdict[:, 0]] == [1] * (len(pts) - 1) + [0]
    times, frob = run(4), run(5)
    assert frob == [psi(p) for p in pts]
    assert times == [B.G2.to_affine(B.G2.mul(B.G2.from_affine(p), k)) for p in pts]
    assert times[:-1] == frob[:-1] and times[-1] != frob[-1]


# ------------------------------------------------------------------------------ NTT / H
@pytest.mark.parametrize("log_n", [4, 5, 6, 7, 8, 9, 10])
def test_ntt_golden(gpu_ctx, log_n):
    fx = golden("ntt_%d.json" % log_n)
    x = ints_to_mont([int(v, 16) for v in fx["input"]], R_MOD)
    for name, inv, coset in (("fft", 0, 0), ("ifft", 1, 0), ("coset_fft", 0, 1), ("icoset_fft", 1, 1)):
        assert mont_to_ints(gpu_ctx.ntt(x, inv, coset), R_MOD) == [int(v, 16) for v in fx[name]], name


# 21: the smallest domain that takes three passes (pass B and its twiddle table)
@pytest.mark.parametrize("log_n", [1, 2, 3, 11, 12, 13, 16, 18, 21])
def test_ntt_vs_cpu_oracle(gpu_ctx, log_n):
    from oracle import cpu_binding as cb
    rng = np.random.default_rng(log_n)
    x = rand_fr(rng, 1 << log_n)
    for inv, coset in ((0, 0), (1, 0), (0, 1), (1, 1)):
        assert np.array_equal(gpu_ctx.ntt(x, inv, coset), cb.fft(x, inv, coset, threads=8)), (inv, coset)


@pytest.mark.parametrize("log_n", [20, 22, 24, 26])
def test_ntt_full_size_properties(gpu_ctx, log_n):
    """Round trip, linearity and the transform of a spike at benchmark sizes, up to the 2^26 of the stand-alone
    configuration (no oracle run needed: the spike's spectrum is known in closed form)."""
    rng = np.random.default_rng(7)
    n = 1 << log_n
    x = rand_fr(rng, n)
    fx = gpu_ctx.ntt(x)
    assert np.array_equal(gpu_ctx.ntt(fx, inverse=True), x)
    assert np.array_equal(gpu_ctx.ntt(gpu_ctx.ntt(x, coset=True), inverse=True, coset=True), x)
    # a single spike s at position j transforms to X[k] = s * omega^(j k): checked with big integers at positions
    # spread over the passes' digits
    j = 12345 % n
    e = np.zeros((n, 4), dtype=np.uint64)
    e[j] = x[j]
    fe = gpu_ctx.ntt(e)
    assert np.array_equal(fe[0], x[j])
    omega = pow(pow(7, (R_MOD - 1) >> 28, R_MOD), 1 << (28 - log_n), R_MOD)
    spike = mont_to_ints(x[j:j + 1], R_MOD)[0]
    ks = [1, 2, 3, 1023, 1024, 1025, n // 2 - 1, n // 2, n - 1, 777, (1 << (log_n - 3)) + 5, 0x2F3A7 % n]
    assert mont_to_ints(fe[ks], R_MOD) == [spike * pow(omega, j * k % n, R_MOD) % R_MOD for k in ks]
    y = x.copy()
    y[12345 % n] = 0
    fy = gpu_ctx.ntt(y)
    # linearity: F(x) = F(y) + F(e), checked on a sample of positions with big integers
    idx = [0, 1, 2, n // 2, n - 1, 777]
    got = mont_to_ints(fx[idx], R_MOD)
    want = [(a + b) % R_MOD for a, b in zip(mont_to_ints(fy[idx], R_MOD), mont_to_ints(fe[idx], R_MOD))]
    assert got == want


@pytest.mark.parametrize("log_n,world", [(2, 2), (6, 2), (8, 4), (10, 8), (13, 2), (16, 4), (20, 8)])
def test_ntt_sharded_steps(gpu_ctx, log_n, world):
    """The transform spread over `world` shards with one exchange: every shard's kernels run back to back on
    this GPU through the C ABI and numpy plays the all-to-all.  Equals the whole-vector transform (the CPU
    oracle's up to 2^16, zg16_ntt's above), forward and inverse."""
    from oracle import cpu_binding as cb
    from zinc_amd.distributed import ntt_shard_input, ntt_shard_output_index
    L = gpu_ctx._lib
    n = 1 << log_n
    log_g = world.bit_length() - 1
    rng = np.random.default_rng(log_n * 10 + world)
    x = rand_fr(rng, n)
    whole = cb.fft(x, False, False, threads=8) if log_n <= 16 else gpu_ctx.ntt(x)
    chunk = n // world // world

    def run(parts, inverse):
        bufs = [np.ascontiguousarray(p, dtype=np.uint64) for p in parts]
        local = lambda t: L.zg16_ntt(gpu_ctx.handle, t.ctypes.data_as(C.c_void_p), log_n - log_g, inverse, 0, 0)
        cross = lambda t, g: L.zg16_ntt_cross(gpu_ctx.handle, t.ctypes.data_as(C.c_void_p), log_n, log_g, g, inverse, 0)

        def exchange(ts):
            host = [t.reshape(world, chunk, 4) for t in ts]
            return [np.ascontiguousarray(np.stack([host[g][h] for g in range(world)]).reshape(-1, 4)) for h in range(world)]
        if not inverse:
            assert all(local(t) == 0 for t in bufs)
            bufs = exchange(bufs)
            assert all(cross(t, g) == 0 for g, t in enumerate(bufs))
        else:
            assert all(cross(t, g) == 0 for g, t in enumerate(bufs))
            bufs = exchange(bufs)
            assert all(local(t) == 0 for t in bufs)
        return bufs

    got = run([ntt_shard_input(x, g, world) for g in range(world)], 0)
    for g in range(world):
        assert np.array_equal(got[g], whole[ntt_shard_output_index(n, g, world)]), ("forward", g)
    back = run(got, 1)
    for g in range(world):
        assert np.array_equal(back[g], ntt_shard_input(x, g, world)), ("inverse", g)
    # a shard count the split does not allow
    t = np.zeros((8, 4), dtype=np.uint64)
    assert L.zg16_ntt_cross(gpu_ctx.handle, t.ctypes.data_as(C.c_void_p), 2, 2, 0, 0, 0) == 5
    assert L.zg16_ntt_cross(gpu_ctx.handle, t.ctypes.data_as(C.c_void_p), 4, 1, 2, 0, 0) == 5


@pytest.mark.parametrize("rows", [3, 6])
def test_h_poly_golden(gpu_ctx, rows):
    fx = golden("h_poly_%d.json" % rows)
    a, b, c = (ints_to_mont([int(v, 16) for v in fx[k]], R_MOD) for k in "abc")
    h, m = gpu_ctx.h_coefficients(a, b, c)
    assert m == fx["m"] and mont_to_ints(h, R_MOD) == [int(v, 16) for v in fx["h"]]


@pytest.mark.parametrize("rows", [1, 2, 3, 5, 17, 100, 1000, 1025, 2048, 2049, 4096, 4097, 5000, (1 << 15) + 3])
def test_h_poly_vs_cpu_oracle(gpu_ctx, monkeypatch, rows):
    """Domains of 2^0 .. 2^15: up to 2^12 the H polynomial comes from the two small-domain launches -- every transform
    cut in two halves on two workgroups (k_small_chains_split, k_small_finish_split) from 2^2 on, whole on one
    workgroup (k_small_chains, k_small_finish; ZG16_SMALL_SPLIT=0) below that --, above 2^12 from the general
    passes; all equal the CPU restatement of bellman, and the general passes give the same at the small sizes too
    (ZG16_SMALL_H=0)."""
    from oracle import cpu_binding as cb
    rng = np.random.default_rng(rows)
    a, b, c = (rand_fr(rng, rows) for _ in range(3))
    h, m = gpu_ctx.h_coefficients(a, b, c)
    h2, m2 = cb.h_coefficients(a, b, c, threads=8)
    assert m == m2 and np.array_equal(h, h2)
    if rows <= 4096:
        monkeypatch.setenv("ZG16_SMALL_SPLIT", "0")
        h4, m4 = gpu_ctx.h_coefficients(a, b, c)
        assert m4 == m2 and np.array_equal(h4, h2)
        monkeypatch.setenv("ZG16_SMALL_H", "0")
        h3, m3 = gpu_ctx.h_coefficients(a, b, c)
        assert m3 == m2 and np.array_equal(h3, h2)


# ------------------------------------------------------------------------------ multiexp
@pytest.mark.parametrize("name,group", [("msm_g1_4", "g1"), ("msm_g1_8", "g1"), ("msm_g1_12", "g1"),
                                        ("msm_g2_4", "g2"), ("msm_g2_8", "g2")])
def test_msm_golden(gpu_ctx, name, group):
    fx = golden(name + ".json")
    to_m, from_m, fn = (g1_to_mont, mont_to_g1, gpu_ctx.msm_g1) if group == "g1" else (g2_to_mont, mont_to_g2, gpu_ctx.msm_g2)
    bases = [unhex(p) for p in fx["bases"]]
    sc = ints_to_mont([int(v, 16) for v in fx["scalars"]], R_MOD)
    assert from_m(fn(to_m(bases), sc))[0] == unhex(fx["result"])
    dens = np.array(fx["density"], dtype=np.uint8)
    assert from_m(fn(to_m([p for p, d in zip(bases, dens) if d]), sc, dens))[0] == unhex(fx["result_density"])


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_closing_sums_at_group_boundaries(gpu_ctx, group):
    """The sums that end a multiexp run with a point spread over 4 (G1) / 8 (G2) lanes: lane groups stride over the
    points, fold inside a wave and across the four waves of a workgroup.  Term counts on both sides of every one of
    those boundaries (1 .. 257 unit scalars: only the unit-scalar sums run; then small scalars: heavy buckets)
    against the CPU oracle."""
    from oracle import cpu_binding as cb
    from zinc_amd import workload
    rng = np.random.default_rng(77)
    g1gen, g2gen = workload._generators()
    one = ints_to_mont([1], R_MOD)[0]
    fn = gpu_ctx.msm_g1 if group == "g1" else gpu_ctx.msm_g2
    sizes = (1, 2, 3, 4, 5, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 255, 256, 257, 700) if group == "g1" else (1, 2, 3, 7, 8, 9, 31, 32, 33, 127, 128, 129, 300)
    top = max(sizes)
    ks = rand_fr(rng, top)
    all_bases = workload.fixed_base_g1(gpu_ctx, g1gen, ks) if group == "g1" else workload.fixed_base_g2(gpu_ctx, g2gen, ks)
    for n in sizes:
        bases = np.ascontiguousarray(all_bases[:n])
        sc = np.tile(one, (n, 1))
        st, want = cb.multiexp(group, bases, sc, None, threads=4)
        assert st == 0 and np.array_equal(fn(bases, sc), want), "unit scalars, n = %d" % n
    small = ints_to_mont([int(v) for v in rng.integers(2, 6, size=top)], R_MOD)
    st, want = cb.multiexp(group, all_bases, small, None, threads=4)
    assert st == 0 and np.array_equal(fn(all_bases, small), want)


def test_msm_edge_cases(gpu_ctx):
    from oracle import bn254 as O
    G = O.G1_GEN
    empty8, empty4 = np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64)
    assert mont_to_g1(gpu_ctx.msm_g1(empty8, empty4))[0] is None                                   # empty
    assert mont_to_g1(gpu_ctx.msm_g1(g1_to_mont([G] * 5), ints_to_mont([0] * 5, R_MOD)))[0] is None  # all zero
    want = O.G1.to_affine(O.G1.mul(O.G1.gen, 700))
    assert mont_to_g1(gpu_ctx.msm_g1(g1_to_mont([G] * 700), ints_to_mont([1] * 700, R_MOD)))[0] == want  # all ones
    assert mont_to_g1(gpu_ctx.msm_g1(g1_to_mont([G, G]), ints_to_mont([5, R_MOD - 5], R_MOD)))[0] is None  # cancels
    want = O.G1.to_affine(O.G1.mul(O.G1.gen, (R_MOD - 1) * 3 % R_MOD))
    assert mont_to_g1(gpu_ctx.msm_g1(g1_to_mont([G] * 3), ints_to_mont([R_MOD - 1] * 3, R_MOD)))[0] == want  # max scalar
    # a cleared density bit skips the scalar without consuming a base
    dens = np.array([1, 0, 1], dtype=np.uint8)
    got = mont_to_g1(gpu_ctx.msm_g1(g1_to_mont([G, G]), ints_to_mont([2, 1000, 3], R_MOD), dens))[0]
    assert got == O.G1.to_affine(O.G1.mul(O.G1.gen, 5))
    # an identity base consumed by a non-zero scalar is bellman's UnexpectedIdentity; by a zero scalar it is not
    with pytest.raises(SynthesisError) as e:
        gpu_ctx.msm_g1(g1_to_mont([G, None]), ints_to_mont([2, 3], R_MOD))
    assert e.value.status_name == "UnexpectedIdentity"
    with pytest.raises(SynthesisError):
        gpu_ctx.msm_g1(g1_to_mont([G, None]), ints_to_mont([2, 1], R_MOD))
    assert mont_to_g1(gpu_ctx.msm_g1(g1_to_mont([G, None]), ints_to_mont([2, 0], R_MOD)))[0] == O.G1.to_affine(O.G1.mul(O.G1.gen, 2))
    with pytest.raises(SynthesisError) as e:
        gpu_ctx.msm_g1(g1_to_mont([G]), ints_to_mont([2, 3], R_MOD))
    assert e.value.status_name == "InvalidArgument"


@pytest.mark.parametrize("group,n", [("g1", 1 << 14), ("g1", 50000), ("g2", 1 << 12)])
def test_msm_vs_cpu_oracle(gpu_ctx, group, n):
    """Seeded bases k_i*G (built by the GPU fixed-base multiplier), skewed scalars (zinc-like)."""
    from oracle import cpu_binding as cb
    from zinc_amd import workload
    rng = np.random.default_rng(n)
    g1gen, g2gen = workload._generators()
    ks = rand_fr(rng, n)
    bases = workload.fixed_base_g1(gpu_ctx, g1gen, ks) if group == "g1" else workload.fixed_base_g2(gpu_ctx, g2gen, ks)
    sc = rand_fr(rng, n)
    one = ints_to_mont([1], R_MOD)[0]
    kind = rng.integers(0, 10, size=n)
    sc[kind < 2] = 0
    sc[(kind >= 2) & (kind < 6)] = one
    small = ints_to_mont([int(v) for v in rng.integers(2, 300, size=64)], R_MOD)
    sc[kind == 6] = small[rng.integers(0, 64, size=int((kind == 6).sum()))]
    dens = (rng.integers(0, 4, size=n) != 0).astype(np.uint8)
    fn = gpu_ctx.msm_g1 if group == "g1" else gpu_ctx.msm_g2
    st, want = cb.multiexp(group, bases, sc, None, threads=8)
    assert st == 0 and np.array_equal(fn(bases, sc), want)
    st, want = cb.multiexp(group, bases[dens != 0], sc, dens, threads=8)
    assert st == 0 and np.array_equal(fn(bases[dens != 0], sc, dens), want)


@pytest.mark.parametrize("group,n,tables", [("g1", 20000, 1), ("g1", 20000, 0), ("g2", 3000, 1), ("g1", 1, 1), ("g1", 0, 1)])
def test_msm_resident_bases(gpu_ctx, group, n, tables):
    """zg16_bases_upload / zg16_msm_resident (window tables or not) against the CPU oracle.  Device-resident
    scalars are exercised by tools/bench_standalone.py (torch has to own the HIP runtime before this library
    loads, which a pytest session cannot guarantee)."""
    from oracle import cpu_binding as cb
    from zinc_amd import workload
    from zinc_amd.groth16 import ResidentBases
    rng = np.random.default_rng(7 * n + tables)
    g1gen, g2gen = workload._generators()
    width = 8 if group == "g1" else 16
    if n:
        ks = rand_fr(rng, n)
        bases = workload.fixed_base_g1(gpu_ctx, g1gen, ks) if group == "g1" else workload.fixed_base_g2(gpu_ctx, g2gen, ks)
    else:
        bases = np.zeros((0, width), dtype=np.uint64)
    sc = rand_fr(rng, n) if n else np.zeros((0, 4), dtype=np.uint64)
    if n > 10:
        sc[::3] = ints_to_mont([1], R_MOD)[0]
        sc[1::7] = 0
    res = ResidentBases(gpu_ctx, 1 if group == "g1" else 2, bases, window_tables=bool(tables))
    if n:
        st, want = cb.multiexp(group, bases, sc, None, threads=8)
        assert st == 0
    else:
        want = np.zeros(width, dtype=np.uint64)
    assert np.array_equal(res.msm(sc), want)
    if n:
        assert np.array_equal(res.msm(sc), want)  # the handle is reusable
    res.free()


@pytest.mark.parametrize("group,n", [("g1", 5000), ("g2", 700)])
def test_msm_direct_tables_vs_buckets_and_oracle(gpu_ctx, group, n):
    """A small resident base set is summed through its direct table (every digit multiple looked up, no
    buckets); the same set uploaded with direct tables switched off goes through the bucket pipeline.  Both equal
    the CPU oracle on scalars chosen to hit the digit recoding's corners: 0, 1, 2, 15, 16, 17, 31, 32, r-1, r-2,
    (r-1)/2, all-ones, every window 15 / 16 / 17, single high bits, and an identity base under a zero scalar."""
    from oracle import cpu_binding as cb
    from zinc_amd import workload
    from zinc_amd.groth16 import ResidentBases
    rng = np.random.default_rng(n)
    g1gen, g2gen = workload._generators()
    ks = rand_fr(rng, n)
    bases = workload.fixed_base_g1(gpu_ctx, g1gen, ks) if group == "g1" else workload.fixed_base_g2(gpu_ctx, g2gen, ks)
    every = lambda d: sum(d << (5 * w) for w in range(50))
    edge = [0, 1, 2, 15, 16, 17, 31, 32, 33, R_MOD - 1, R_MOD - 2, (R_MOD - 1) // 2, (1 << 253) - 1, every(15), every(16),
            every(17), every(31), 1 << 250, 1 << 253, (1 << 253) + (1 << 252), 16 << 250, 15 << 250]
    sc = rand_fr(rng, n)
    sc[: len(edge)] = ints_to_mont(edge, R_MOD)
    sc[100:140:2] = ints_to_mont([1], R_MOD)[0]
    bases[0] = 0          # the point at infinity, under a zero scalar: consumed by nobody
    st, want = cb.multiexp(group, bases, sc, None, threads=8)
    assert st == 0
    gid = 1 if group == "g1" else 2
    direct = ResidentBases(gpu_ctx, gid, bases, window_tables=True)
    os.environ["ZG16_DIRECT_MAX_LOG"] = "0"
    try:
        buckets = ResidentBases(gpu_ctx, gid, bases, window_tables=True)
    finally:
        del os.environ["ZG16_DIRECT_MAX_LOG"]
    assert np.array_equal(direct.msm(sc), want)
    assert np.array_equal(buckets.msm(sc), want)
    # an identity base under a non-zero scalar is bellman's UnexpectedIdentity on both paths
    sc[0] = ints_to_mont([5], R_MOD)[0]
    for res in (direct, buckets):
        with pytest.raises(SynthesisError) as e:
            res.msm(sc)
        assert e.value.status_name == "UnexpectedIdentity"
    direct.free()
    buckets.free()


def test_msm_full_size_linearity(gpu_ctx):
    """2^20 points: msm(s) + msm(t) == msm(s + t) and msm(k*s) == k*msm(s), via 3-term multiexps."""
    from zinc_amd import workload
    rng = np.random.default_rng(99)
    n = 1 << 20
    g1gen, _ = workload._generators()
    bases = workload.fixed_base_g1(gpu_ctx, g1gen, rand_fr(rng, n))
    s = rand_fr(rng, n)
    t = rand_fr(rng, n)
    s_plus_t = _field(gpu_ctx, 0, 0, s, t)
    ps, pt, pst = gpu_ctx.msm_g1(bases, s), gpu_ctx.msm_g1(bases, t), gpu_ctx.msm_g1(bases, s_plus_t)
    one = ints_to_mont([1, 1], R_MOD)
    assert np.array_equal(gpu_ctx.msm_g1(np.stack([ps, pt]), one), pst)
    k = ints_to_mont([0xABCDEF12345], R_MOD)
    ks = _field(gpu_ctx, 0, 2, s, np.tile(k, (n, 1)))
    assert np.array_equal(gpu_ctx.msm_g1(bases, ks), gpu_ctx.msm_g1(ps.reshape(1, 8), k))


def test_msm_resident_2p26_linearity(gpu_ctx):
    """The stand-alone configuration's size, 2^26 points, through zg16_bases_upload / zg16_msm_resident (window
    tables): msm(s) + msm(t) == msm(s + t), msm(k s) == k msm(s), and a sparse scalar vector equals the small
    multiexp over the bases it selects (which test_msm_vs_cpu_oracle pins to the oracle)."""
    from zinc_amd import workload
    from zinc_amd.groth16 import ResidentBases
    rng = np.random.default_rng(2026)
    n = 1 << 26
    g1gen, _ = workload._generators()
    bases = workload.fixed_base_g1(gpu_ctx, g1gen, rand_fr(rng, n))
    res = ResidentBases(gpu_ctx, 1, bases, window_tables=True)
    s = rand_fr(rng, n)
    t = rand_fr(rng, n)
    ps, pt = res.msm(s), res.msm(t)
    step = 1 << 24  # the element-wise self-test hook takes at most 2^24 elements per call
    chunked = lambda op, a, b: np.concatenate([_field(gpu_ctx, 0, op, a[i:i + step], b[i:i + step]) for i in range(0, n, step)])
    s_plus_t = chunked(0, s, t)
    pst = res.msm(s_plus_t)
    one = ints_to_mont([1, 1], R_MOD)
    assert np.array_equal(gpu_ctx.msm_g1(np.stack([ps, pt]), one), pst)
    del s_plus_t, t
    k = ints_to_mont([0xABCDEF12345], R_MOD)
    ks = chunked(2, s, np.tile(k, (n, 1)))
    assert np.array_equal(res.msm(ks), gpu_ctx.msm_g1(ps.reshape(1, 8), k))
    del ks
    # 3000 non-zero scalars scattered over the whole range (first, last, a run across the middle)
    sparse = np.zeros((n, 4), dtype=np.uint64)
    idx = np.unique(np.concatenate([[0, 1, n - 1, n - 2], rng.integers(0, n, size=2900), np.arange(n // 2 - 48, n // 2 + 48)]))
    sparse[idx] = s[idx]
    sparse[idx[::5]] = one[0]
    assert np.array_equal(res.msm(sparse), gpu_ctx.msm_g1(bases[idx], sparse[idx]))
    res.free()


# ------------------------------------------------------------------------------ SpMV / prove
@pytest.mark.parametrize("idx", [1, 2, 3])
def test_prove_golden(gpu_ctx, idx):
    fx = golden("prove_tiny_%d.json" % idx)
    circ, key = FixtureCircuit(fx), FixtureKey(fx)
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circ.synthesize(pa)
    pa.finalize_inputs()
    mat = pa.r1cs().upload(gpu_ctx)
    zi, za = pa.witness()
    a, b, c = mat.spmv(zi, za)
    assert mont_to_ints(a, R_MOD) == [int(v, 16) for v in fx["eval_a"]]
    assert mont_to_ints(b, R_MOD) == [int(v, 16) for v in fx["eval_b"]]
    assert mont_to_ints(c, R_MOD) == [int(v, 16) for v in fx["eval_c"]]
    assert mat.density() == (sum(fx["a_aux_density"]), sum(fx["b_input_density"]), sum(fx["b_aux_density"]))
    mat.free()
    proof = create_proof(circ, Parameters.from_points(key), int(fx["r"], 16), int(fx["s"], 16), gpu_ctx)
    assert proof.as_tuple() == fixture_proof(fx)
    from zinc_amd import codec
    assert codec.proof_write(proof.as_tuple()).hex() == fx["proof_bytes"]


@pytest.mark.parametrize("idx", [1, 3])
def test_key_file_decoded_on_gpu(gpu_ctx, idx):
    """`Parameters::read(file, checked = true)` on the GPU: the key bytes the codec writes decode to a key
    whose proof is the golden one; damaged files are rejected the way upstream's reader rejects them."""
    from zinc_amd import codec
    from zinc_amd.groth16 import DeviceKey
    from oracle import codec as ocodec
    fx = golden("prove_tiny_%d.json" % idx)
    circ, key = FixtureCircuit(fx), FixtureKey(fx)
    data = ocodec.params_bytes(key)      # the oracle's bytes, not the product codec's
    assert codec.params_write(key) == data and codec.vk_write(key) == ocodec.vk_bytes(key)
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circ.synthesize(pa)
    pa.finalize_inputs()
    mat = pa.r1cs().upload(gpu_ctx)
    zi, za = pa.witness()
    rs = ints_to_mont([int(fx["r"], 16), int(fx["s"], 16)], R_MOD)
    # checked = True: subgroup membership of every b_g2 point by psi(P) = [6x^2] P; 2: by the plain r P = O (the
    # slower form kept as its checker) -- the two accept the same points
    for checked in (True, 2, False):
        dev = DeviceKey.from_bytes(gpu_ctx, data, checked=checked)
        assert prove_recorded(gpu_ctx, dev, mat, zi, za, rs[0], rs[1]).as_tuple() == fixture_proof(fx)
        dev.free()
    mat.free()

    def rejected(blob, checked=True):
        with pytest.raises(SynthesisError) as e:
            DeviceKey.from_bytes(gpu_ctx, blob, checked=checked)
        return e.value.status_name == "MalformedKey"

    vk_len = len(codec.vk_write(key))
    h0 = vk_len + 4  # first point of h
    bad = bytearray(data); bad[h0 + 63] ^= 1                      # y of h[0] off the curve
    assert rejected(bad)
    DeviceKey.from_bytes(gpu_ctx, bytes(bad), checked=False).free()   # unchecked accepts it, as upstream does
    bad = bytearray(data); bad[h0:h0 + 64] = bytes([0x40]) + bytes(63)  # infinity inside a query vector
    assert rejected(bad)
    bad = bytearray(data); bad[h0] |= 0x80                        # compression flag on an uncompressed point
    assert rejected(bad)
    bad = bytearray(data); bad[h0:h0 + 32] = (codec.Q_MOD).to_bytes(32, "big")   # x = q is not a field element
    assert rejected(bad)
    assert rejected(data[:-1]) and rejected(data[:vk_len + 2]) and rejected(data[:10])
    # a point on the twist that is not in the prime-order subgroup, in place of b_g2[0]
    x = (5, 1)
    while True:
        rhs = codec._fq2_mul(codec._fq2_mul(x, x), x)
        y = codec._fq2_sqrt(((rhs[0] + codec.B_G2[0]) % Q_MOD, (rhs[1] + codec.B_G2[1]) % Q_MOD))
        if y is not None:
            break
        x = (x[0] + 1, 1)
    off = len(data) - 128 * len(key.b_g2)
    assert not codec._g2_times_order_is_identity((x, y))
    bad = bytearray(data); bad[off:off + 128] = codec.g2_write((x, y), False)
    assert rejected(bad) and rejected(bad, checked=2)
    DeviceKey.from_bytes(gpu_ctx, bytes(bad), checked=False).free()
    # the same point as gamma_g2: verifying-key points are checked on the host, with the same two tests
    g0 = 64 * 2 + 128
    bad = bytearray(data); bad[g0:g0 + 128] = codec.g2_write((x, y), False)
    assert rejected(bad) and rejected(bad, checked=2) and rejected(bad, checked=False)
    # alpha_g1 (a verifying-key point) may be anything on the curve, but not off it -- whatever `checked` says:
    # upstream's VerifyingKey::read validates its points unconditionally
    bad = bytearray(data); bad[63] ^= 1
    assert rejected(bad) and rejected(bad, checked=False)
    # the identity inside ic is refused (checked or not); the identity as alpha_g1 is a valid encoding
    ic0 = 64 * 3 + 128 * 3 + 4
    bad = bytearray(data); bad[ic0:ic0 + 64] = bytes([0x40]) + bytes(63)
    assert rejected(bad) and rejected(bad, checked=False)
    ok = bytearray(data); ok[0:64] = bytes([0x40]) + bytes(63)
    DeviceKey.from_bytes(gpu_ctx, bytes(ok), checked=True).free()


@pytest.mark.parametrize("log_rows,dense", [(10, False), (12, True), (14, False), (16, False)])
def test_prove_vs_cpu_oracle(gpu_ctx, log_rows, dense):
    """Seeded zinc-shaped circuits: GPU proof == proof of the CPU restatement of bellman, bit for bit."""
    from oracle import cpu_binding as cb
    from zinc_amd import workload
    circ = workload.SyntheticCircuit(1 << log_rows, dense=dense, seed=(log_rows, 2, 3, 4))
    params = workload.generate_parameters(gpu_ctx, circ, dict(k1=5, k2=7, alpha=11, beta=13, gamma=17, delta=19, tau=23 + log_rows))
    r1cs = circ.r1cs()
    zi, za = circ.witness()
    rs = ints_to_mont([0x123456789, 0x987654321], R_MOD)
    st, want, _ = cb.prove(params, r1cs, zi, za, rs[0], rs[1], threads=8)
    assert st == 0
    key, mat = params.upload(gpu_ctx), r1cs.upload(gpu_ctx)
    (ea, eb, ec), _ = cb.eval_rows(r1cs, zi, za)
    ga, gb, gc = mat.spmv(zi, za)
    assert np.array_equal(ga, ea) and np.array_equal(gb, eb) and np.array_equal(gc, ec)
    got = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
    for g, w in zip((got.raw.a, got.raw.b, got.raw.c), want):
        assert np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), w)
    # a host that keeps bellman's row evaluation hands a, b, c over instead: same proof
    from zinc_amd.groth16 import prove_from_evals
    ev = prove_from_evals(gpu_ctx, key, mat, ea, eb, ec, zi, za, rs[0], rs[1])
    assert bytes(ev.raw) == bytes(got.raw)
    # the two-shard path gives the same proof (what two GPUs would compute, run back to back here)
    from zinc_amd.distributed import finish
    parts = (zlib.Partial * 2)()
    for i in range(2):
        shard = params.upload(gpu_ctx, i, 2)
        st = gpu_ctx._lib.zg16_prove_partial(gpu_ctx.handle, shard.handle, mat.handle, zi.ctypes.data_as(C.c_void_p),
                                             za.ctypes.data_as(C.c_void_p), 0, C.byref(parts[i]))
        assert st == 0
        shard.free()
    pv = params.view()
    p2 = finish(pv, parts, 2, rs[0], rs[1])
    assert bytes(p2) == bytes(got.raw)
    key.free()
    mat.free()


@pytest.mark.parametrize("rows,n_pub,dense,pack", [
    (37, 1, False, 0), (300, 3, True, 31), (1025, 16, False, 64), (4095, 2, False, 248), (4097, 5, True, 100),
    (9001, 16, False, 248), (20000, 7, False, 17), (40000, 1, True, 248)])
def test_prove_ragged_sizes_vs_cpu_oracle(gpu_ctx, rows, n_pub, dense, pack):
    """Row counts that are not powers of two (domains padded with zero rows, both sides of the one-workgroup
    transform limit 2^12 and of the direct-table limit 2^14), few and many public inputs, short and long packing
    rows, r and s of full size: GPU proof == proof of the CPU restatement of bellman, and the library's verifier
    accepts it against the circuit's public inputs."""
    from oracle import cpu_binding as cb
    from zinc_amd import workload
    from zinc_amd.groth16 import verify_proof
    from zinc_amd.fields import mont_to_ints
    circ = workload.SyntheticCircuit(rows, n_pub=n_pub, dense=dense, pack_width=pack, seed=(rows, n_pub, 11, 13))
    assert circ.satisfied()
    params = workload.generate_parameters(gpu_ctx, circ, dict(k1=3, k2=5, alpha=rows + 7, beta=11, gamma=13, delta=17, tau=19 + n_pub))
    r1cs = circ.r1cs()
    zi, za = circ.witness()
    rs = ints_to_mont([pow(7, rows, R_MOD), pow(5, rows + 1, R_MOD)], R_MOD)
    st, want, _ = cb.prove(params, r1cs, zi, za, rs[0], rs[1], threads=8)
    assert st == 0
    key, mat = params.upload(gpu_ctx), r1cs.upload(gpu_ctx)
    try:
        got = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
        for g, w in zip((got.raw.a, got.raw.b, got.raw.c), want):
            assert np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), w)
        public = mont_to_ints(zi[1:], R_MOD)
        assert verify_proof(params, got, public)
        if public:
            assert not verify_proof(params, got, [(public[0] + 1) % R_MOD] + public[1:])
    finally:
        key.free()
        mat.free()


@pytest.mark.parametrize("kind", ["zeros", "ones", "minus_ones", "uniform", "sparse"])
@pytest.mark.parametrize("rows", [600, 1 << 15])
def test_prove_degenerate_witnesses_vs_cpu_oracle(gpu_ctx, kind, rows):
    """Witness vectors a synthesizer would never produce -- every aux value zero (every multiexp sums nothing, H is
    zero), one (everything goes down the unit-scalar path), r - 1, uniform, or a few non-zero values in a sea of zeros.
    They do not satisfy the circuit; bellman's create_proof does not look (the division by Z on the coset is then
    simply not exact) and neither does this library: the proof must still equal the CPU restatement's bit for bit."""
    from oracle import cpu_binding as cb
    from zinc_amd import workload
    circ = workload.SyntheticCircuit(rows, n_pub=2, seed=(rows, 5, 6, 7))
    params = workload.generate_parameters(gpu_ctx, circ, dict(k1=2, k2=3, alpha=5, beta=7, gamma=11, delta=13, tau=17))
    r1cs = circ.r1cs()
    zi, za = circ.witness()
    rng = np.random.default_rng(rows)
    n = za.shape[0]
    if kind == "zeros":
        za = np.zeros_like(za)
    elif kind == "ones":
        za = np.ascontiguousarray(np.repeat(ints_to_mont([1], R_MOD), n, axis=0))
    elif kind == "minus_ones":
        za = np.ascontiguousarray(np.repeat(ints_to_mont([R_MOD - 1], R_MOD), n, axis=0))
    elif kind == "uniform":
        za = ints_to_mont([pow(int(x), 5, R_MOD) for x in rng.integers(2, 1 << 62, size=n)], R_MOD)
    else:
        za = np.zeros_like(za)
        idx = rng.choice(n, size=7, replace=False)
        za[idx] = ints_to_mont([pow(3, int(i) + 2, R_MOD) for i in idx], R_MOD)
    za = np.ascontiguousarray(za, dtype=np.uint64)
    rs = ints_to_mont([R_MOD - 2, 1], R_MOD)
    st, want, _ = cb.prove(params, r1cs, zi, za, rs[0], rs[1], threads=8)
    assert st == 0
    key, mat = params.upload(gpu_ctx), r1cs.upload(gpu_ctx)
    try:
        got = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
        for g, w in zip((got.raw.a, got.raw.b, got.raw.c), want):
            assert np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), w)
    finally:
        key.free()
        mat.free()


@pytest.mark.parametrize("log_rows", [8, 13])
def test_graph_replay_gives_the_same_proofs(gpu_ctx, log_rows):
    """A proof replayed as a captured HIP graph equals the plainly launched one: first call plain, second captured,
    later ones replayed -- with a different witness and different r, s each time (the witness is staged outside the
    graph); a larger proof in between moves the scratch arenas and forces a new capture."""
    from zinc_amd import workload
    toxic = dict(k1=3, k2=5, alpha=7, beta=11, gamma=13, delta=17, tau=29 + log_rows)
    circs = [workload.SyntheticCircuit(1 << log_rows, n_pub=4, seed=(log_rows, 1, 2, 3))]
    params = workload.generate_parameters(gpu_ctx, circs[0], toxic)
    key, mat = params.upload(gpu_ctx), circs[0].r1cs().upload(gpu_ctx)
    zi, za = circs[0].witness()
    # other satisfying witnesses of the same matrices do not exist for free; other (r, s) pairs and a perturbed
    # witness (the prover does not check satisfaction) exercise the same data paths
    zb = za.copy()
    zb[: len(zb) // 2] = za[len(za) // 2: len(za) // 2 + len(zb) // 2]
    cases = [(za, 11, 12), (za, 13, 14), (zb, 15, 16), (za, 17, 18), (zb, 19, 20), (za, 21, 22)]
    gpu_ctx.graph(False)
    want = [prove_recorded(gpu_ctx, key, mat, zi, w, *ints_to_mont([r, s], R_MOD)).as_tuple() for w, r, s in cases]
    gpu_ctx.graph(True)
    c0, r0, f0 = gpu_ctx.graph_counts()
    try:
        got = []
        for k, (w, r, s) in enumerate(cases):
            if k == 4:  # grow the scratch arenas: the captured graph must not be replayed on stale pointers
                big = workload.SyntheticCircuit(1 << (log_rows + 2), n_pub=4, seed=(9, 9, 9, 9))
                bp = workload.generate_parameters(gpu_ctx, big, toxic)
                bk, bm = bp.upload(gpu_ctx), big.r1cs().upload(gpu_ctx)
                bzi, bza = big.witness()
                prove_recorded(gpu_ctx, bk, bm, bzi, bza, *ints_to_mont([1, 2], R_MOD))
                bk.free()
                bm.free()
            got.append(prove_recorded(gpu_ctx, key, mat, zi, w, *ints_to_mont([r, s], R_MOD)).as_tuple())
        assert got == want
        c1, r1, f1 = gpu_ctx.graph_counts()
        # plain, capture + replay, replay, replay | (a second capture if the larger proof moved an arena) replay, replay
        assert c1 - c0 in (1, 2) and (r1 - r0, f1 - f0) == (5, 0)
    finally:
        gpu_ctx.graph(None)
    key.free()
    mat.free()


def test_prove_full_size_verifies(gpu_ctx):
    """2^20 rows (the benchmark workload): the proof satisfies the Groth16 pairing equation, is
    reproducible, and changes with r."""
    from oracle import bn254 as O, groth16 as G
    from zinc_amd import workload
    circ = workload.SyntheticCircuit(1 << 20)
    params = workload.generate_parameters(gpu_ctx, circ, dict(k1=3, k2=5, alpha=7, beta=11, gamma=13, delta=17, tau=0xC0FFEE))
    key, mat = params.upload(gpu_ctx), circ.r1cs().upload(gpu_ctx)
    zi, za = circ.witness()
    rs = ints_to_mont([0xDEADBEEF, 0xFEEDFACE, 0xDEADBEF0], R_MOD)
    p1 = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
    p2 = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
    p3 = prove_recorded(gpu_ctx, key, mat, zi, za, rs[2], rs[1])
    assert p1.as_tuple() == p2.as_tuple() != p3.as_tuple()

    class VK:
        pass
    vk = VK()
    vk.alpha_g1, vk.delta_g1 = mont_to_g1(params.alpha_g1)[0], mont_to_g1(params.delta_g1)[0]
    vk.beta_g2, vk.delta_g2, vk.gamma_g2 = (mont_to_g2(x)[0] for x in (params.beta_g2, params.delta_g2, params.gamma_g2))
    vk.ic = mont_to_g1(params.ic)
    public = mont_to_ints(zi[1:], R_MOD)
    assert G.verify_proof(vk, p1.as_tuple(), public)
    assert G.verify_proof(vk, p3.as_tuple(), public)
    assert not G.verify_proof(vk, p1.as_tuple(), [(public[0] + 1) % R_MOD] + public[1:])
    key.free()
    mat.free()


def _vk_of(params):
    class VK:
        pass
    vk = VK()
    vk.alpha_g1, vk.delta_g1 = mont_to_g1(params.alpha_g1)[0], mont_to_g1(params.delta_g1)[0]
    vk.beta_g2, vk.delta_g2, vk.gamma_g2 = (mont_to_g2(x)[0] for x in (params.beta_g2, params.delta_g2, params.gamma_g2))
    vk.ic = mont_to_g1(params.ic)
    return vk


def test_prove_2p24_equals_cpu_restatement_and_verifies(gpu_ctx):
    """2^24 rows (the large synthetic configuration; three-pass transforms, ~70 GB of key tables): the GPU proof
    is bit-identical to the CPU restatement of bellman's prover and satisfies the Groth16 pairing equation -- the
    reference's own bar at this boundary is verify == true (zinc-tester/src/runners/proof_check.rs:243-264)."""
    from oracle import cpu_binding as cb, groth16 as G
    from zinc_amd import workload
    circ = workload.SyntheticCircuit(1 << 24)
    params = workload.generate_parameters(gpu_ctx, circ, dict(k1=3, k2=5, alpha=7, beta=11, gamma=13, delta=17, tau=0xFACADE))
    r1cs = circ.r1cs()
    zi, za = circ.witness()
    rs = ints_to_mont([0xDEADBEEF, 0xFEEDFACE], R_MOD)
    key, mat = params.upload(gpu_ctx), r1cs.upload(gpu_ctx)
    got = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
    key.free()
    mat.free()
    st, want, _ = cb.prove(params, r1cs, zi, za, rs[0], rs[1], threads=min(cb.hardware_threads() or 1, 16))
    assert st == 0
    for g, w in zip((got.raw.a, got.raw.b, got.raw.c), want):
        assert np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), w)
    public = mont_to_ints(zi[1:], R_MOD)
    assert G.verify_proof(_vk_of(params), got.as_tuple(), public)


# ------------------------------------------------------------------------------ error behaviour
def test_error_statuses(gpu_ctx):
    lib = gpu_ctx._lib
    buf = np.zeros((2, 4), dtype=np.uint64)
    assert lib.zg16_ntt(gpu_ctx.handle, buf.ctypes.data_as(C.c_void_p), 28, 0, 0, 0) == 1  # PolynomialDegreeTooLarge
    assert b"2^27" in lib.zg16_last_error(gpu_ctx.handle)
    fx = golden("prove_tiny_1.json")
    circ, key = FixtureCircuit(fx), FixtureKey(fx)
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circ.synthesize(pa)
    pa.finalize_inputs()
    zi, za = pa.witness()
    rs = ints_to_mont([1, 2], R_MOD)
    mat = pa.r1cs().upload(gpu_ctx)
    # a key whose l query is one point short
    short = Parameters.from_points(key)
    short.l = short.l[:-1].copy()
    dk = short.upload(gpu_ctx)
    with pytest.raises(SynthesisError) as e:
        prove_recorded(gpu_ctx, dk, mat, zi, za, rs[0], rs[1])
    assert e.value.status_name == "MalformedKey"
    dk.free()
    # delta at infinity
    bad = Parameters.from_points(key)
    bad.delta_g1 = np.zeros((1, 8), dtype=np.uint64)
    dk = bad.upload(gpu_ctx)
    with pytest.raises(SynthesisError) as e:
        prove_recorded(gpu_ctx, dk, mat, zi, za, rs[0], rs[1])
    assert e.value.status_name == "UnexpectedIdentity"
    dk.free()
    # missing witness
    dk = Parameters.from_points(key).upload(gpu_ctx)
    proof = zlib.Proof()
    st = lib.zg16_prove(gpu_ctx.handle, dk.handle, mat.handle, None, None, 0, zlib.as_u64p(rs[0]), zlib.as_u64p(rs[1]),
                        C.byref(proof))
    assert st == 3  # AssignmentMissing
    dk.free()
    mat.free()
    # a column that names a variable the circuit does not have
    r = pa.r1cs()
    rp, col, co = r.mats[0]
    col = col.copy()
    col[0] = 0x80000000 | 5000
    with pytest.raises(SynthesisError) as e:
        R1CS(r.n_inputs, r.n_aux, [(rp, col, co), r.mats[1], r.mats[2]]).upload(gpu_ctx)
    assert e.value.status_name == "InvalidArgument"


def test_cpp_host_mirror_driver(gpu_ctx, tmp_path):
    """The C++ mirror of the bellman interface (zinc_amd/host/zinc_groth16.hpp), driven like zinc-vm drives bellman."""
    import os
    import struct
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "bin", "host_driver")
    assert os.path.exists(exe), "bin/host_driver missing: run __graft_entry__.build()"
    fx = golden("prove_tiny_3.json")
    circ, key = FixtureCircuit(fx), FixtureKey(fx)
    params = Parameters.from_points(key)
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circ.synthesize(pa)          # recorded WITHOUT the input rows: the C++ side must append them itself
    r1cs = pa.r1cs()
    zi, za = pa.witness()
    want = fixture_proof(fx)
    blob = b"".join(x.tobytes() for x in (params.alpha_g1, params.beta_g1, params.beta_g2, params.delta_g1, params.delta_g2))
    for arr in (params.h, params.l, params.a, params.b_g1, params.b_g2, zi, za):
        blob += struct.pack("<Q", arr.shape[0]) + arr.tobytes()
    for rp, col, co in r1cs.mats:
        blob += struct.pack("<Q", rp.shape[0]) + rp.tobytes() + struct.pack("<Q", col.shape[0]) + col.tobytes() + co.tobytes()
    blob += ints_to_mont([int(fx["r"], 16), int(fx["s"], 16)], R_MOD).tobytes()
    blob += g1_to_mont([want[0]]).tobytes() + g2_to_mont([want[1]]).tobytes() + g1_to_mont([want[2]]).tobytes()
    path = tmp_path / "case.bin"
    path.write_bytes(blob)
    res = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr


# ------------------------------------------------------------------------------ multi-process
def _run_ranks(script, world, port, *args):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", script), str(r), str(world), str(port)] +
                              [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return outs


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_transform_over_ranks(world):
    """`world` processes on one GPU (gloo stands in for RCCL): zinc_amd.distributed.ntt_sharded forward and
    inverse, equal to zg16_ntt on the whole vector."""
    outs = _run_ranks("gpu_ntt_worker.py", world, 29550 + world, 14)
    assert "sharded transform" in outs[0], outs[0]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_prove_with_split_transforms(world):
    """`world` processes on one GPU (gloo stands in for RCCL): key sharded by point range, the A/B/C
    transform chains on different ranks, H broadcast, partials gathered -- bit-identical to zg16_prove."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "gpu_split_worker.py")
    env = dict(os.environ, PYTHONPATH=root)
    port = 29540 + world
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port), "12"], env=env,
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
    assert "bit-identical" in outs[0], outs[0]


@pytest.mark.gpu
def test_spmv_coefficient_codes(gpu_ctx):
    """Row evaluation with every kind of coefficient code: one, minus one, +-2^k for every k, a few thousand other
    values, and more distinct values than there are codes (those terms fall back to the full coefficient array)."""
    rng = np.random.default_rng(5)
    pa = ProvingAssignment()
    pa.alloc_input(1)
    zs = [int.from_bytes(rng.bytes(32), "little") % R_MOD for _ in range(64)]
    vs = [pa.alloc(v) for v in zs]
    coeffs = [1, R_MOD - 1] + [1 << k for k in range(1, 254)] + [R_MOD - (1 << k) for k in range(1, 254)]
    coeffs += [3, 5, R_MOD - 3, (1 << 200) + 1]
    coeffs += [int.from_bytes(rng.bytes(32), "little") % R_MOD or 1 for _ in range(70000)]
    want = []
    for i, c in enumerate(coeffs):
        v, w = i % 64, (7 * i + 1) % 64
        if v == w:
            w = (w + 1) % 64
        pa.enforce([(vs[v], c)], [(0, 1), (vs[w], c)], [(vs[w], 1), (vs[v], R_MOD - 1)])
        want.append((c * zs[v] % R_MOD, (1 + c * zs[w]) % R_MOD, (zs[w] - zs[v]) % R_MOD))
    # one long row (a wavefront's worth of terms) mixing the kinds
    long_lc = [(vs[i % 64] if i < 64 else (i - 63), coeffs[(5 * i) % 600]) for i in range(64)]
    pa.enforce(long_lc, [], [])
    want.append((sum(c * (zs[i]) for i, (_, c) in enumerate(long_lc)) % R_MOD, 0, 0))
    pa.finalize_inputs()
    mat = pa.r1cs().upload(gpu_ctx)
    zi, za = pa.witness()
    a, b, c = (mont_to_ints(x, R_MOD) for x in mat.spmv(zi, za))
    n = len(want)
    assert a[:n] == [w[0] for w in want]
    assert b[:n] == [w[1] for w in want]
    assert c[:n] == [w[2] for w in want]
    mat.free()


@pytest.mark.gpu
@pytest.mark.parametrize("world,log_rows", [(2, 12), (4, 12), (2, 16), (4, 20)])
def test_sharded_transforms_over_ranks_prove(world, log_rows):
    """`world` processes on one GPU (gloo stands in for RCCL): a, b, c stay sharded through the whole H computation
    (zg16_hshard_*: one destination-major exchange per stage), h query cyclic -- bit-identical to zg16_prove, up to
    the headline size."""
    outs = _run_ranks("gpu_hshard_worker.py", world, 29600 + world + log_rows, log_rows)
    assert "bit-identical" in outs[0], outs[0]


@pytest.mark.gpu
@pytest.mark.parametrize("devices,sharded,log_rows", [([0, 0], True, 12), ([0, 0, 0, 0], True, 12), ([0, 0], True, 16),
                                                       ([0, 0, 0, 0], True, 16), ([0] * 8, True, 13), ([0, 0, 0], False, 12),
                                                       ([0] * 8, True, 20), ([0, 0], None, 20)])
def test_multi_gpu_prover_in_one_process(gpu_ctx, devices, sharded, log_rows):
    """zg16_multi_*: one process, a context per listed device (all on GPU 0 here), peer copies for the exchanges.
    Sharded transforms at 2, 4 and 8 shards and the replicated schedule at 3: the proof equals zg16_prove's."""
    from zinc_amd import workload
    from zinc_amd.groth16 import MultiProver
    circ = workload.SyntheticCircuit(1 << log_rows, n_pub=4, seed=(log_rows, 9, 9, 9))
    params = workload.generate_parameters(gpu_ctx, circ, dict(k1=3, k2=5, alpha=7, beta=11, gamma=13, delta=17, tau=31))
    r1cs = circ.r1cs()
    zi, za = circ.witness()
    rs = ints_to_mont([0xABCDEF, 0x13579B], R_MOD)
    key, mat = params.upload(gpu_ctx), r1cs.upload(gpu_ctx)
    want = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
    key.free()
    mat.free()
    mp = MultiProver(devices)
    # sharded None: the library chooses (sharded from 2^20 rows when the device count allows it)
    assert mp.load(params, r1cs, sharded_transforms=sharded) == (True if sharded is None else sharded)
    for _ in range(2):
        got = mp.prove(zi, za, rs[0], rs[1])
        assert got.as_tuple() == want.as_tuple()
    mp.close()


@pytest.mark.gpu
def test_multi_gpu_prover_errors(gpu_ctx):
    from zinc_amd.groth16 import MultiProver
    fx = golden("prove_tiny_1.json")
    circ, key = FixtureCircuit(fx), FixtureKey(fx)
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circ.synthesize(pa)
    pa.finalize_inputs()
    mp = MultiProver([0, 0, 0])
    with pytest.raises(SynthesisError) as e:      # three GPUs cannot shard the transforms
        mp.load(Parameters.from_points(key), pa.r1cs(), sharded_transforms=True)
    assert e.value.status_name == "InvalidArgument"
    assert mp.load(Parameters.from_points(key), pa.r1cs()) is False
    zi, za = pa.witness()
    rs = ints_to_mont([int(fx["r"], 16), int(fx["s"], 16)], R_MOD)
    assert mp.prove(zi, za, rs[0], rs[1]).as_tuple() == fixture_proof(fx)
    bad = Parameters.from_points(key)
    bad.delta_g2 = np.zeros((1, 16), dtype=np.uint64)
    mp.load(bad, pa.r1cs())
    with pytest.raises(SynthesisError) as e:
        mp.prove(zi, za, rs[0], rs[1])
    assert e.value.status_name == "UnexpectedIdentity"
    mp.close()


# ------------------------------------------------------------------------------ key generation
@pytest.mark.gpu
@pytest.mark.parametrize("idx", [1, 2, 3])
def test_generate_parameters_golden(gpu_ctx, idx):
    """zg16_generate_parameters from the fixture's toxic waste gives the fixture's key, point for point
    (bellman `generate_parameters`: Lagrange values, per-variable sums, identity points dropped from a / b)."""
    from zinc_amd import workload
    fx = golden("prove_tiny_%d.json" % idx)
    circ, key = FixtureCircuit(fx), FixtureKey(fx)
    pa = ProvingAssignment()
    pa.alloc_input(1)
    circ.synthesize(pa)
    pa.finalize_inputs()
    got = workload.generate_parameters(gpu_ctx, pa.r1cs(), {k: int(v, 16) for k, v in fx["toxic"].items()})
    want = Parameters.from_points(key)
    for name in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2", "h", "l", "a", "b_g1", "b_g2"):
        assert np.array_equal(getattr(got, name), getattr(want, name)), name
    assert np.array_equal(got.ic, g1_to_mont(key.ic)) and np.array_equal(got.gamma_g2, g2_to_mont([key.gamma_g2]))


@pytest.mark.gpu
def test_generate_parameters_vs_host_and_unconstrained(gpu_ctx):
    from zinc_amd import workload
    # (2^15 rows: ONE occurs in more rows than one segment of the transposed product holds)
    circ = workload.SyntheticCircuit(1 << 15, n_pub=5, seed=(4, 3, 2, 1))
    toxic = dict(k1=5, k2=7, alpha=11, beta=13, gamma=17, delta=19, tau=0x1234567)
    got = workload.generate_parameters(gpu_ctx, circ, toxic)
    want = workload.generate_parameters_host(gpu_ctx, circ, toxic)
    for name in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2", "gamma_g2", "ic", "h", "l", "a", "b_g1", "b_g2"):
        assert np.array_equal(getattr(got, name), getattr(want, name)), name
    # an aux variable that no row mentions: bellman's UnconstrainedVariable
    pa = ProvingAssignment()
    pa.alloc_input(1)
    x, y = pa.alloc(3), pa.alloc(4)
    pa.enforce([(x, 1)], [(x, 1)], [(0, 9)])
    pa.finalize_inputs()
    with pytest.raises(SynthesisError) as e:
        workload.generate_parameters(gpu_ctx, pa.r1cs(), toxic)
    assert e.value.status_name == "UnconstrainedVariable"


@pytest.mark.gpu
@pytest.mark.parametrize("n,storage_root", [(5, None), (6, None), (7, None), (6, 0x1234567890abcdef << 100)])
def test_factorial_stand_in_proves_and_verifies(gpu_ctx, n, storage_root):
    """A zinc-tester-sized circuit (hand-recorded factorial.zn; with a storage root appended it has the shape of a
    contract method's): key from zg16_generate_parameters, proof bit-identical to the CPU restatement, and it
    verifies against the public inputs the reference's test pins (proof_check.rs:243-264: verify == true)."""
    from oracle import cpu_binding as cb
    from oracle import groth16 as G
    from zinc_amd import standin, workload
    circ = standin.FactorialCircuit(n, storage_root)
    pa = standin.record(circ)
    assert circ.result == standin.FactorialCircuit.CASES[n]
    r1cs = pa.r1cs()
    params = workload.generate_parameters(gpu_ctx, r1cs, dict(k1=3, k2=5, alpha=7, beta=11, gamma=13, delta=17, tau=0xFAC70))
    zi, za = pa.witness()
    rs = ints_to_mont([0x5EED, 0xFEED], R_MOD)
    st, want, _ = cb.prove(params, r1cs, zi, za, rs[0], rs[1], threads=4)
    assert st == 0
    key, hit = params.cached(gpu_ctx)
    mat, _ = r1cs.cached(gpu_ctx, b"factorial.zn/main" + (b"+root" if storage_root is not None else b""))
    got = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
    for g, w in zip((got.raw.a, got.raw.b, got.raw.c), want):
        assert np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), w)
    public = circ.public_inputs()
    assert G.verify_proof(_vk_of(params), got.as_tuple(), public)
    assert not G.verify_proof(_vk_of(params), got.as_tuple(), [public[0] + 1] + public[1:])
    # the library's own verifier agrees with the oracle's
    from zinc_amd.groth16 import verify_proof
    assert verify_proof(params, got, public) and not verify_proof(params, got, [public[0] + 1] + public[1:])



def _stand_in_programs():
    from zinc_amd import standin
    return [("fibonacci.zn/12", lambda: standin.FibonacciCircuit(12), [89]),
            ("fibonacci.zn/16", lambda: standin.FibonacciCircuit(16), [610]),
            ("division_u8.zn/48/7", lambda: standin.DivisionU8Circuit(48, 7), [6]),
            ("division_u8.zn/255/255", lambda: standin.DivisionU8Circuit(255, 255), [1]),
            ("uniswap/get_eth_to_token_input_price",
             lambda: standin.UniswapInputPriceCircuit(10 ** 18, 5 * 10 ** 21, 7 * 10 ** 23, 0xC0FFEE << 200), None)]


@pytest.mark.gpu
@pytest.mark.parametrize("which", range(5))
def test_more_stand_ins_prove_and_verify(gpu_ctx, which):
    """More hand-recorded programs: zinc-tester's fibonacci.zn and division_u8.zn (each verifies against the output
    its `//!` header pins: proof_check.rs:243-264 is `verify == true`) and the uniswap example's
    get_eth_to_token_input_price (zinc-examples/uniswap/src/main.zn:705-712 -- a contract method: its storage root
    is the last public input; the reference holds no expected output for it, the public input is the formula's
    value).  Proof bit-identical to the CPU restatement; both verifiers accept it and refuse a changed input."""
    from oracle import cpu_binding as cb
    from oracle import groth16 as G
    from zinc_amd import standin, workload
    from zinc_amd.groth16 import verify_proof
    name, make, pinned = _stand_in_programs()[which]
    circ = make()
    pa = standin.record(circ)
    public = circ.public_inputs()
    if pinned is not None:
        assert public == pinned
    else:
        assert public == [standin.UniswapInputPriceCircuit.price(10 ** 18, 5 * 10 ** 21, 7 * 10 ** 23), (0xC0FFEE << 200) % R_MOD]
    r1cs = pa.r1cs()
    params = workload.generate_parameters(gpu_ctx, r1cs, dict(k1=3, k2=5, alpha=7, beta=11, gamma=13, delta=17, tau=0xFAC71 + which))
    zi, za = pa.witness()
    rs = ints_to_mont([0x5EED + which, 0xFEED], R_MOD)
    st, want, _ = cb.prove(params, r1cs, zi, za, rs[0], rs[1], threads=4)
    assert st == 0
    key, mat = params.upload(gpu_ctx), r1cs.upload(gpu_ctx)
    got = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
    key.free()
    mat.free()
    for g, w in zip((got.raw.a, got.raw.b, got.raw.c), want):
        assert np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), w), name
    assert G.verify_proof(_vk_of(params), got.as_tuple(), public)
    assert verify_proof(params, got, public)
    for k in range(len(public)):
        bad = list(public)
        bad[k] = (bad[k] + 1) % R_MOD
        assert not verify_proof(params, got, bad)
