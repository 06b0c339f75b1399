"""GPU (-m gpu): zg16_msm_resident_noalloc, the entry point that allocates nothing, against zg16_msm_resident on
the same base sets and scalars (bit-exact), two calls in a row with the second on a smaller input; then its
limits.  Collected last so that a failure here cannot hide the other tests under `pytest -x`."""
import numpy as np
import pytest

from zinc_amd.fields import R_MOD, ints_to_mont
from zinc_amd.groth16 import ResidentBases, SynthesisError

pytestmark = pytest.mark.gpu


def rand_fr(rng, n):
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    x[:, 3] >>= np.uint64(3)
    return x


@pytest.mark.parametrize("group", [1, 2])
def test_noalloc_equals_resident_twice_second_smaller(gpu_ctx, group):
    from zinc_amd import workload
    rng = np.random.default_rng(40 + group)
    g1gen, g2gen = workload._generators()
    fixed = workload.fixed_base_g1 if group == 1 else workload.fixed_base_g2
    gen = g1gen if group == 1 else g2gen
    sizes = (9000, 700) if group == 1 else (2500, 300)
    sets, scalars = [], []
    for n in sizes:
        sets.append(ResidentBases(gpu_ctx, group, fixed(gpu_ctx, gen, rand_fr(rng, n)), window_tables=True))
        sc = rand_fr(rng, n)
        sc[::3] = ints_to_mont([1], R_MOD)[0]
        sc[1::7] = 0
        sc[2] = ints_to_mont([R_MOD - 1], R_MOD)[0]
        scalars.append(sc)
    want = [b.msm(sc) for b, sc in zip(sets, scalars)]
    # two calls in a row through the entry point that allocates nothing, the second on the smaller input: the
    # shared workspace still holds the first call's partial sums
    got = [b.msm(sc, noalloc=True) for b, sc in zip(sets, scalars)]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(sets[0].msm(scalars[0], noalloc=True), want[0])  # and back to the larger one
    for b in sets:
        b.free()


def test_noalloc_limits(gpu_ctx):
    from zinc_amd import workload
    rng = np.random.default_rng(5)
    g1gen, _ = workload._generators()
    n = (1 << 14) + 1  # one point beyond the limit (and beyond what gets a direct table)
    big = ResidentBases(gpu_ctx, 1, workload.fixed_base_g1(gpu_ctx, g1gen, rand_fr(rng, n)), window_tables=True)
    sc = rand_fr(rng, n)
    with pytest.raises(SynthesisError) as e:
        big.msm(sc, noalloc=True)
    assert e.value.status_name == "InvalidArgument" and "at most 16384" in str(e.value)
    big.free()
    plain = ResidentBases(gpu_ctx, 1, workload.fixed_base_g1(gpu_ctx, g1gen, rand_fr(rng, 100)), window_tables=False)
    with pytest.raises(SynthesisError) as e:
        plain.msm(rand_fr(rng, 100), noalloc=True)
    assert e.value.status_name == "InvalidArgument"
    plain.free()
    empty = ResidentBases(gpu_ctx, 1, np.zeros((0, 8), dtype=np.uint64), window_tables=True)
    assert not empty.msm(np.zeros((0, 4), dtype=np.uint64), noalloc=True).any()
    empty.free()


def _hip_runtime():
    """The HIP runtime the library itself runs on (already mapped into this process), for hipMemGetInfo and a plain
    device buffer -- torch is left out: it cannot take over a runtime that another library initialised first."""
    import ctypes as C
    with open("/proc/self/maps") as f:
        paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert paths, "the HIP runtime is not loaded"
    hip = C.CDLL(sorted(paths)[0])
    for name in ("hipMemGetInfo", "hipMalloc", "hipMemcpy", "hipFree", "hipDeviceSynchronize"):
        getattr(hip, name).restype = C.c_int
    return hip


def _free_bytes(hip):
    import ctypes as C
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


@pytest.mark.parametrize("group", [1, 2])
def test_noalloc_leaves_gpu_memory_untouched(gpu_ctx, group):
    """The claim itself: across calls of zg16_msm_resident_noalloc the GPU's free memory (hipMemGetInfo) does not
    move, with scalars on the device and with scalars in host memory (staged through the context's page-locked
    block, not through the runtime's own staging of a pageable copy).  Control: a 4 MiB hipMalloc -- the order of
    the workspace such a call would need -- does move it, so the meter sees allocations of that size."""
    import ctypes as C
    import zinc_amd
    from zinc_amd import workload
    rng = np.random.default_rng(90 + group)
    g1gen, g2gen = workload._generators()
    fixed = workload.fixed_base_g1 if group == 1 else workload.fixed_base_g2
    n = 6000 if group == 1 else 1500
    ctx = zinc_amd.Context(0)        # fresh: its scratch arenas are empty
    bases = ResidentBases(ctx, group, fixed(ctx, g1gen if group == 1 else g2gen, rand_fr(rng, n)), window_tables=True)
    hip = _hip_runtime()
    sc = np.ascontiguousarray(rand_fr(rng, n))
    sc_dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(sc_dev), C.c_size_t(sc.nbytes)) == 0
    assert hip.hipMemcpy(sc_dev, C.c_void_p(sc.ctypes.data), C.c_size_t(sc.nbytes), 1) == 0   # host to device
    want = None
    free0 = _free_bytes(hip)
    for k in range(6):
        got = bases.msm(sc, noalloc=True) if k % 2 else bases.msm(device_ptr=sc_dev.value, noalloc=True)
        want = got if want is None else want
        assert np.array_equal(got, want)
    assert _free_bytes(hip) == free0, "zg16_msm_resident_noalloc changed the GPU's free memory"
    probe = C.c_void_p()
    assert hip.hipMalloc(C.byref(probe), C.c_size_t(4 << 20)) == 0
    assert _free_bytes(hip) < free0, "hipMemGetInfo does not see a 4 MiB allocation: the check above proves nothing"
    assert hip.hipFree(probe) == 0
    assert np.array_equal(bases.msm(sc), want)            # the allocating entry point: same sum
    assert hip.hipFree(sc_dev) == 0
    bases.free()
    ctx.close()
