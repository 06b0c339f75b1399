"""Keys and circuits that stay resident between create_proof calls (zg16_default_ctx, zg16_pk_cached,
zg16_r1cs_cached): what a drop-in for zinc-tester's `create_random_proof(circuit, &params.clone(), rng)` loop
(zinc-tester/src/runners/proof_check.rs:122, :181) needs so that only the first case pays for the key upload."""
import os
import time

import numpy as np
import pytest

from conftest import golden, FixtureCircuit, FixtureKey, fixture_proof
from zinc_amd.fields import ints_to_mont, R_MOD

pytestmark = pytest.mark.gpu


class CountingCircuit(FixtureCircuit):
    """Counts the rows the constraint system was actually handed (a WitnessAssignment drops them)."""

    def synthesize(self, cs):
        self.cs_type = type(cs).__name__
        super().synthesize(cs)


def test_default_context_is_one_per_gpu():
    import zinc_amd
    a, b = zinc_amd.Context.default(0), zinc_amd.Context.default(0)
    assert a.handle.value == b.handle.value
    own = zinc_amd.Context(0)
    assert own.handle.value != a.handle.value
    own.close()
    a.close()   # does not destroy the shared context
    assert zinc_amd.Context.default(0).handle.value == b.handle.value


def test_create_proof_reuses_key_and_circuit():
    import zinc_amd
    from zinc_amd.groth16 import Parameters, create_proof
    ctx = zinc_amd.Context.default(0)
    ctx.cache_clear()
    fx1, fx2 = golden("prove_tiny_1.json"), golden("prove_tiny_2.json")
    p1, p2 = Parameters.from_points(FixtureKey(fx1)), Parameters.from_points(FixtureKey(fx2))
    for rnd in range(3):
        for fx, params, cid in ((fx1, p1, b"fixture-1/main"), (fx2, p2, b"fixture-2/main")):
            circ = CountingCircuit(fx)
            proof = create_proof(circ, params, int(fx["r"], 16), int(fx["s"], 16), circuit_id=cid)
            assert proof.as_tuple() == fixture_proof(fx)
            assert circ.cs_type == ("ProvingAssignment" if rnd == 0 else "WitnessAssignment")
    st = ctx.cache_stats()
    assert (st["keys"], st["key_misses"], st["key_hits"]) == (2, 2, 4)
    assert (st["circuits"], st["circuit_misses"]) == (2, 2) and st["circuit_hits"] >= 4 and st["evictions"] == 0
    assert st["key_bytes"] > 0
    # a copy of the same key in other host buffers (`params.clone()`) is the same key
    clone = Parameters.from_points(FixtureKey(fx1))
    k_a, hit_a = p1.cached(ctx)
    k_b, hit_b = clone.cached(ctx, full_digest=False)
    assert hit_a and hit_b and k_a.handle.value == k_b.handle.value
    # zg16_pk_free leaves a cached handle alone
    ctx._lib.zg16_pk_free(k_a.handle)
    circ = CountingCircuit(fx1)
    assert create_proof(circ, p1, int(fx1["r"], 16), int(fx1["s"], 16), circuit_id=b"fixture-1/main").as_tuple() == fixture_proof(fx1)
    # without a circuit id the matrices themselves are digested: recorded every time, uploaded once
    before = ctx.cache_stats()
    for _ in range(2):
        circ = CountingCircuit(fx2)
        assert create_proof(circ, p2, int(fx2["r"], 16), int(fx2["s"], 16)).as_tuple() == fixture_proof(fx2)
        assert circ.cs_type == "ProvingAssignment"
    after = ctx.cache_stats()
    assert after["circuit_misses"] == before["circuit_misses"] + 1 and after["circuit_hits"] == before["circuit_hits"] + 1
    ctx.cache_clear()
    assert ctx.cache_stats()["keys"] == 0


def test_threads_share_the_default_context(monkeypatch):
    """Several host threads proving through create_proof at once (zinc-tester runs its cases on a thread pool): the
    default context is one object for all of them, and with room for a single key every proof evicts the other
    thread's key -- each call still returns its own fixture's proof."""
    import threading
    import zinc_amd
    from zinc_amd.groth16 import Parameters, create_proof
    monkeypatch.setenv("ZG16_CACHE_KEYS", "1")
    zinc_amd.Context.default(0).cache_clear()
    fxs = [golden("prove_tiny_%d.json" % i) for i in (1, 2, 3)]
    params = [Parameters.from_points(FixtureKey(fx)) for fx in fxs]
    failures = []

    def worker(k):
        try:
            for _ in range(6):
                fx = fxs[k]
                got = create_proof(FixtureCircuit(fx), params[k], int(fx["r"], 16), int(fx["s"], 16),
                                   circuit_id=b"threads/%d" % k)
                if got.as_tuple() != fixture_proof(fx):
                    failures.append((k, "wrong proof"))
        except Exception as e:   # a failure in a thread must fail the test, not vanish
            failures.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures
    assert zinc_amd.Context.default(0).cache_stats()["evictions"] > 0
    zinc_amd.Context.default(0).cache_clear()


def test_cache_evicts_least_recently_used(monkeypatch):
    import zinc_amd
    from zinc_amd.groth16 import Parameters, create_proof
    ctx = zinc_amd.Context(0)
    monkeypatch.setenv("ZG16_CACHE_KEYS", "2")
    fxs = [golden("prove_tiny_%d.json" % i) for i in (1, 2, 3)]
    params = [Parameters.from_points(FixtureKey(fx)) for fx in fxs]
    for fx, p in zip(fxs, params):
        assert create_proof(FixtureCircuit(fx), p, int(fx["r"], 16), int(fx["s"], 16), ctx).as_tuple() == fixture_proof(fx)
    st = ctx.cache_stats()
    assert st["keys"] == 2 and st["evictions"] == 1
    _, hit = params[2].cached(ctx)
    assert hit
    _, hit = params[0].cached(ctx)     # the oldest was pushed out: uploaded again, and still proves correctly
    assert not hit
    fx = fxs[0]
    assert create_proof(FixtureCircuit(fx), params[0], int(fx["r"], 16), int(fx["s"], 16), ctx).as_tuple() == fixture_proof(fx)
    ctx.close()


def test_resident_key_saves_the_upload(gpu_ctx):
    """2^16 rows: the second create-proof-shaped call (cached key + circuit) costs a small fraction of the first."""
    from zinc_amd import workload
    from zinc_amd.groth16 import prove_recorded
    from oracle import cpu_binding as cb
    circ = workload.SyntheticCircuit(1 << 16, seed=(7, 7, 7, 7))
    params = workload.generate_parameters(gpu_ctx, circ, dict(k1=5, k2=7, alpha=11, beta=13, gamma=17, delta=19, tau=29))
    r1cs = circ.r1cs()
    zi, za = circ.witness()
    rs = ints_to_mont([0x1234, 0x5678], R_MOD)
    times, proofs = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        key, _ = params.cached(gpu_ctx)
        mat, _ = r1cs.cached(gpu_ctx, b"synthetic-2^16")
        proofs.append(prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1]))
        times.append(time.perf_counter() - t0)
    st, want, _ = cb.prove(params, r1cs, zi, za, rs[0], rs[1], threads=8)
    assert st == 0
    for got in proofs:
        for g, w in zip((got.raw.a, got.raw.b, got.raw.c), want):
            assert np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), w)
    assert min(times[1:]) < times[0] / 3, times
    gpu_ctx.cache_clear()


@pytest.mark.parametrize("budget_mb,want_step", [(None, (1,)), (100, (2,)), (45, (3, 4)), (1, None)])
def test_table_memory_budget_falls_back_to_fewer_windows(gpu_ctx, monkeypatch, budget_mb, want_step):
    """A key whose full window tables do not fit the budget keeps every step-th window and the multiexps use `step`
    bucket sets per result: the proof is the same.  (2^14 rows with the direct tables of small queries switched
    off, so that the bucket pipeline is what runs.)"""
    from zinc_amd import workload
    from zinc_amd.groth16 import prove_recorded, DeviceKey
    from oracle import cpu_binding as cb
    monkeypatch.setenv("ZG16_DIRECT_MAX_LOG", "0")
    circ = workload.SyntheticCircuit(1 << 14, seed=(1, 4, 1, 4))
    params = workload.generate_parameters(gpu_ctx, circ, dict(k1=5, k2=7, alpha=11, beta=13, gamma=17, delta=19, tau=37))
    r1cs = circ.r1cs()
    zi, za = circ.witness()
    rs = ints_to_mont([0xAAAA, 0xBBBB], R_MOD)
    st, want, _ = cb.prove(params, r1cs, zi, za, rs[0], rs[1], threads=8)
    assert st == 0
    if budget_mb is not None:
        monkeypatch.setenv("ZG16_TABLE_BUDGET_MB", str(budget_mb))
    key, mat = params.upload(gpu_ctx), r1cs.upload(gpu_ctx)
    info = key.table_info()
    if want_step is not None:
        assert info["step"] in want_step, info
    else:
        assert info["step"] >= 8, info      # nothing fits 1 MB: as few windows as there are
    if budget_mb is not None and want_step is not None:
        assert info["device_bytes"] <= budget_mb << 20
    got = prove_recorded(gpu_ctx, key, mat, zi, za, rs[0], rs[1])
    for g, w in zip((got.raw.a, got.raw.b, got.raw.c), want):
        assert np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), w)
    key.free()
    mat.free()


def test_stale_key_address_is_not_taken_for_a_resident_key(monkeypatch):
    """A host that remembers (address, serial number) of the handle a key got: after the cache has evicted the key
    the pair is refused, even when a newer key has been allocated at the same address -- the address alone would
    pick the wrong key's tables."""
    import zinc_amd
    from zinc_amd.groth16 import Parameters
    ctx = zinc_amd.Context(0)
    monkeypatch.setenv("ZG16_CACHE_KEYS", "1")
    L = ctx._lib
    fxs = [golden("prove_tiny_%d.json" % i) for i in (1, 2, 3)]
    params = [Parameters.from_points(FixtureKey(fx)) for fx in fxs]
    seen = []
    for p in params * 3:                       # every call evicts the previous key
        k, hit = p.cached(ctx)
        assert not hit
        kid = L.zg16_pk_id(k.handle)
        assert kid and all(kid != i for _, i in seen), "a serial number was handed out twice"
        assert L.zg16_pk_cache_holds(ctx.handle, k.handle, kid) == 1
        assert L.zg16_pk_cache_holds(ctx.handle, k.handle, kid + 1) == 0
        for addr, old in seen:                 # every earlier handle is gone, whatever lives at its address now
            assert L.zg16_pk_cache_holds(ctx.handle, addr, old) == 0
        seen.append((k.handle.value, kid))
    ctx.close()


def test_same_shape_keys_under_one_setup_are_told_apart(gpu_ctx):
    """Two circuits of the same shape that differ in a few rows, keys from the same toxic waste: vk and h are the
    same and most l / a / b points too, so a digest of sampled points can take one for the other.  The digest folds
    every word of every query: each key gets its own resident handle and its own (correct) proof."""
    from zinc_amd import workload
    from zinc_amd.groth16 import prove_recorded, R1CS
    from zinc_amd.fields import mont_to_ints
    from oracle import cpu_binding as cb
    toxic = dict(k1=5, k2=7, alpha=11, beta=13, gamma=17, delta=19, tau=41)
    circ = workload.SyntheticCircuit(1 << 12, seed=(2, 7, 1, 8))
    r1 = circ.r1cs()
    zi, za = circ.witness()
    # the twin: one coefficient of one A row doubled (and its C side is irrelevant to key generation)
    mats = [(rp.copy(), col.copy(), co.copy()) for rp, col, co in r1.mats]
    rp, col, co = mats[0]
    row = 2049
    while rp[row + 1] == rp[row]:
        row += 1
    e = int(rp[row])
    v = mont_to_ints(co[e:e + 1], R_MOD)[0]
    co[e] = ints_to_mont([2 * v % R_MOD], R_MOD)[0]
    r2 = R1CS(r1.n_inputs, r1.n_aux, mats)
    p1 = workload.generate_parameters(gpu_ctx, r1, toxic)
    p2 = workload.generate_parameters(gpu_ctx, r2, toxic)
    assert np.array_equal(p1.h, p2.h) and p1.a.shape == p2.a.shape and not np.array_equal(p1.a, p2.a)
    assert int((p1.a != p2.a).any(axis=1).sum()) <= 2
    gpu_ctx.cache_clear()
    k1, hit1 = p1.cached(gpu_ctx)
    k2, hit2 = p2.cached(gpu_ctx)
    assert not hit1 and not hit2 and k1.handle.value != k2.handle.value
    assert p1.cached(gpu_ctx)[1] and p2.cached(gpu_ctx)[1]
    # the first circuit's witness proves under its own key, bit for bit
    rs = ints_to_mont([0x77, 0x99], R_MOD)
    mat = r1.upload(gpu_ctx)
    got = prove_recorded(gpu_ctx, k1, mat, zi, za, rs[0], rs[1])
    st, want, _ = cb.prove(p1, r1, zi, za, rs[0], rs[1], threads=4)
    assert st == 0
    for g, w in zip((got.raw.a, got.raw.b, got.raw.c), want):
        assert np.array_equal(np.frombuffer(bytes(g), dtype=np.uint64), w)
    mat.free()
    gpu_ctx.cache_clear()


def test_circuit_id_reused_for_another_shape_is_refused():
    """A circuit id that is already resident, handed a circuit with another number of variables: the binding
    compares the witness lengths with zg16_r1cs_shape and reports AssignmentMissing instead of letting zg16_prove
    read past the witness buffers."""
    import zinc_amd
    from zinc_amd.groth16 import Parameters, create_proof, SynthesisError
    ctx = zinc_amd.Context(0)
    fx1, fx3 = golden("prove_tiny_1.json"), golden("prove_tiny_3.json")
    p1, p3 = Parameters.from_points(FixtureKey(fx1)), Parameters.from_points(FixtureKey(fx3))
    assert create_proof(FixtureCircuit(fx1), p1, 3, 5, ctx, circuit_id=b"one-name").as_tuple()
    c1, c3 = FixtureCircuit(fx1), FixtureCircuit(fx3)
    assert (len(c1.z_in), len(c1.z_aux)) != (len(c3.z_in), len(c3.z_aux))
    with pytest.raises(SynthesisError) as e:
        create_proof(c3, p3, 3, 5, ctx, circuit_id=b"one-name")
    assert e.value.status_name == "AssignmentMissing"
    ctx.close()
