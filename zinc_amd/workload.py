"""Synthetic zinc-shaped workloads and GPU-assisted key generation (benchmark / test support).

Circuits come from libzg16_tools.so (seeded, satisfied R1CS instances whose row mix follows what
zinc-vm's gadgets emit); keys come from the scalar half of bellman's `generate_parameters`
(zinc-vm/src/core/circuit/facade.rs:126 is the reference's call) computed on the host plus the
library's fixed-base multiplier on the GPU.  Nothing here runs while proving.
"""
import ctypes as C
import os

import numpy as np

from . import lib as _lib
from .fields import R_MOD, MONT_R, ints_to_mont
from .groth16 import R1CS, Parameters, _check

TOOLS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libzg16_tools.so")
_tools = None
u64p, u32p = _lib.u64p, _lib.u32p


def tools():
    global _tools
    if _tools is None:
        if not os.path.exists(TOOLS_PATH):
            raise RuntimeError("%s is missing: run `make -C zinc_amd/csrc`" % TOOLS_PATH)
        t = C.CDLL(TOOLS_PATH)
        t.zgt_circuit_generate.restype = C.c_void_p
        t.zgt_circuit_generate.argtypes = [C.c_uint64, C.c_uint32, u32p, C.c_int, C.c_uint32]
        t.zgt_circuit_free.argtypes = [C.c_void_p]
        for n in ("zgt_n_rows", "zgt_n_inputs", "zgt_n_aux"):
            getattr(t, n).restype = C.c_uint64
            getattr(t, n).argtypes = [C.c_void_p]
        t.zgt_nnz.restype = C.c_uint64
        t.zgt_nnz.argtypes = [C.c_void_p, C.c_int]
        for n, r in (("zgt_row_ptr", u64p), ("zgt_col", u32p), ("zgt_coeff", u64p)):
            getattr(t, n).restype = r
            getattr(t, n).argtypes = [C.c_void_p, C.c_int]
        for n in ("zgt_z_in", "zgt_z_aux"):
            getattr(t, n).restype = u64p
            getattr(t, n).argtypes = [C.c_void_p]
        t.zgt_circuit_satisfied.restype = C.c_int
        t.zgt_circuit_satisfied.argtypes = [C.c_void_p]
        t.zgt_powers.argtypes = [u64p, u64p, C.c_uint64, u64p]
        t.zgt_points_encode.argtypes = [u64p, C.c_uint64, C.c_int, C.c_void_p, C.c_int]
        t.zgt_setup_scalars.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p, u64p, u64p, u64p, u64p, u64p]
        _tools = t
    return _tools


def _np_from(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()


class SyntheticCircuit:
    """total_rows rows in all (body + n_pub output rows + 1 + n_pub input rows)."""

    def __init__(self, total_rows, n_pub=16, seed=None, dense=False, pack_width=248):
        t = tools()
        seed_arr = (C.c_uint32 * 4)(*seed) if seed is not None else None
        self.handle = t.zgt_circuit_generate(total_rows, n_pub, seed_arr, int(dense), pack_width)
        if not self.handle:
            raise ValueError("total_rows too small for %d public inputs" % n_pub)
        self.n_rows = t.zgt_n_rows(self.handle)
        self.n_inputs = t.zgt_n_inputs(self.handle)
        self.n_aux = t.zgt_n_aux(self.handle)
        assert self.n_rows == total_rows

    def __del__(self):
        try:
            if self.handle:
                tools().zgt_circuit_free(self.handle)
                self.handle = None
        except Exception:
            pass

    def satisfied(self):
        return bool(tools().zgt_circuit_satisfied(self.handle))

    def r1cs(self):
        t = tools()
        mats = []
        for k in range(3):
            nnz = t.zgt_nnz(self.handle, k)
            mats.append((_np_from(t.zgt_row_ptr(self.handle, k), (self.n_rows + 1,), np.uint64),
                         _np_from(t.zgt_col(self.handle, k), (nnz,), np.uint32),
                         _np_from(t.zgt_coeff(self.handle, k), (nnz, 4), np.uint64)))
        return R1CS(self.n_inputs, self.n_aux, mats)

    def witness(self):
        t = tools()
        return (_np_from(t.zgt_z_in(self.handle), (self.n_inputs, 4), np.uint64),
                _np_from(t.zgt_z_aux(self.handle), (self.n_aux, 4), np.uint64))


def _fixed_base(ctx, fn, width, base, scalars):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((scalars.shape[0], width), dtype=np.uint64)
    base = np.ascontiguousarray(base, dtype=np.uint64).reshape(width)
    if scalars.shape[0]:
        _check(ctx.handle, fn(ctx.handle, _lib.as_u64p(base), _lib.as_u64p(scalars), scalars.shape[0], _lib.as_u64p(out)))
    return out


def fixed_base_g1(ctx, base, scalars):
    return _fixed_base(ctx, ctx._lib.zg16_fixed_base_g1, 8, base, scalars)


def fixed_base_g2(ctx, base, scalars):
    return _fixed_base(ctx, ctx._lib.zg16_fixed_base_g2, 16, base, scalars)


G1_GEN_MONT = None
G2_GEN_MONT = None


def _generators():
    global G1_GEN_MONT, G2_GEN_MONT
    if G1_GEN_MONT is None:
        from .fields import g1_to_mont, g2_to_mont
        G1_GEN_MONT = g1_to_mont([(1, 2)])[0]
        G2_GEN_MONT = g2_to_mont([((10857046999023057135944570762232829481370756359578518086990519993285655852781,
                                    11559732032986387107991004021392285783925812861821192530917403151452391805634),
                                   (8495653923123431417604973247489272438418190587263600148770280649306958101930,
                                    4082367875863433681332203403145435568316851327593401208105741076214120093531))])[0]
    return G1_GEN_MONT, G2_GEN_MONT


def generate_parameters(ctx, circuit, toxic):
    """bellman `generate_parameters` on the GPU (zg16_generate_parameters) with explicit toxic waste.

    circuit: an R1CS (rows as for proving, input rows appended) or anything with `.r1cs()`.
    toxic: dict of canonical ints k1, k2 (g1 = k1*G1, g2 = k2*G2), alpha, beta, gamma, delta, tau.
    Returns Parameters with `.ic` ((n_inputs, 8) vk.ic points) and `.gamma_g2` attached.
    """
    r1cs = circuit if isinstance(circuit, R1CS) else circuit.r1cs()
    mont = lambda x: ints_to_mont([x % R_MOD], R_MOD)[0]
    g1gen, g2gen = _generators()
    tw = _lib.ToxicWaste()
    tw.g1[:] = [int(x) for x in fixed_base_g1(ctx, g1gen, mont(toxic["k1"]))[0]]
    tw.g2[:] = [int(x) for x in fixed_base_g2(ctx, g2gen, mont(toxic["k2"]))[0]]
    for name in ("alpha", "beta", "gamma", "delta", "tau"):
        getattr(tw, name)[:] = [int(x) for x in mont(toxic[name])]
    m = 1
    while m < r1cs.n_rows:
        m *= 2
    nv = r1cs.n_inputs + r1cs.n_aux
    ic = np.zeros((r1cs.n_inputs, 8), dtype=np.uint64)
    h = np.zeros((max(m - 1, 1), 8), dtype=np.uint64)
    l = np.zeros((max(r1cs.n_aux, 1), 8), dtype=np.uint64)
    a = np.zeros((nv, 8), dtype=np.uint64)
    b1 = np.zeros((nv, 8), dtype=np.uint64)
    b2 = np.zeros((nv, 16), dtype=np.uint64)
    out = _lib.ParametersOut()
    out.ic, out.h, out.l = _lib.as_u64p(ic), _lib.as_u64p(h), _lib.as_u64p(l)
    out.a, out.b_g1, out.b_g2 = _lib.as_u64p(a), _lib.as_u64p(b1), _lib.as_u64p(b2)
    view = r1cs.view()
    _check(ctx.handle, ctx._lib.zg16_generate_parameters(ctx.handle, C.byref(view), C.byref(tw), C.byref(out)))
    arr = lambda f, w: np.array(list(f), dtype=np.uint64).reshape(1, w)
    params = Parameters(alpha_g1=arr(out.alpha_g1, 8), beta_g1=arr(out.beta_g1, 8), beta_g2=arr(out.beta_g2, 16),
                        delta_g1=arr(out.delta_g1, 8), delta_g2=arr(out.delta_g2, 16), h=h[:m - 1], l=l[:r1cs.n_aux],
                        a=a[:out.n_a].copy(), b_g1=b1[:out.n_b].copy(), b_g2=b2[:out.n_b].copy())
    params.gamma_g2 = arr(out.gamma_g2, 16)
    params.ic = ic
    return params


def generate_parameters_host(ctx, circuit, toxic):
    """The same key with the per-variable sums computed on the host by libzg16_tools (a SyntheticCircuit only):
    an independent check of zg16_generate_parameters.

    toxic: dict of canonical ints k1, k2 (g1 = k1*G1, g2 = k2*G2), alpha, beta, gamma, delta, tau.
    Returns (Parameters, ic) where ic is the (n_inputs, 8) array of vk.ic points.
    """
    t = tools()
    m = 1
    while m < circuit.n_rows:
        m *= 2
    tau, alpha, beta, gamma, delta = (toxic[k] % R_MOD for k in ("tau", "alpha", "beta", "gamma", "delta"))
    mont = lambda x: ints_to_mont([x], R_MOD)[0]
    g1gen, g2gen = _generators()
    g1 = fixed_base_g1(ctx, g1gen, mont(toxic["k1"]))[0]
    g2 = fixed_base_g2(ctx, g2gen, mont(toxic["k2"]))[0]
    # L_j(tau): inverse transform of the powers of tau
    powers = np.zeros((m, 4), dtype=np.uint64)
    t.zgt_powers(_lib.as_u64p(mont(tau)), _lib.as_u64p(mont(1)), m, _lib.as_u64p(powers))
    lag = ctx.ntt(powers, inverse=True)
    nv = circuit.n_inputs + circuit.n_aux
    a_s = np.zeros((nv, 4), dtype=np.uint64)
    b_s = np.zeros((nv, 4), dtype=np.uint64)
    ic_s = np.zeros((circuit.n_inputs, 4), dtype=np.uint64)
    l_s = np.zeros((max(circuit.n_aux, 1), 4), dtype=np.uint64)
    t.zgt_setup_scalars(circuit.handle, _lib.as_u64p(lag), _lib.as_u64p(mont(alpha)), _lib.as_u64p(mont(beta)),
                        _lib.as_u64p(mont(pow(gamma, -1, R_MOD))), _lib.as_u64p(mont(pow(delta, -1, R_MOD))),
                        _lib.as_u64p(a_s), _lib.as_u64p(b_s), _lib.as_u64p(ic_s), _lib.as_u64p(l_s))
    l_s = l_s[:circuit.n_aux]
    # h[i] = g1 * tau^i * (tau^m - 1) / delta, i < m - 1
    zt = (pow(tau, m, R_MOD) - 1) * pow(delta, -1, R_MOD) % R_MOD
    h_s = np.zeros((max(m - 1, 1), 4), dtype=np.uint64)
    t.zgt_powers(_lib.as_u64p(mont(tau)), _lib.as_u64p(mont(zt)), m - 1, _lib.as_u64p(h_s))
    h_s = h_s[:m - 1]
    a_keep = a_s.any(axis=1)
    b_keep = b_s.any(axis=1)
    params = Parameters(
        alpha_g1=fixed_base_g1(ctx, g1, mont(alpha)), beta_g1=fixed_base_g1(ctx, g1, mont(beta)),
        beta_g2=fixed_base_g2(ctx, g2, mont(beta)), delta_g1=fixed_base_g1(ctx, g1, mont(delta)),
        delta_g2=fixed_base_g2(ctx, g2, mont(delta)),
        h=fixed_base_g1(ctx, g1, h_s), l=fixed_base_g1(ctx, g1, l_s),
        a=fixed_base_g1(ctx, g1, a_s[a_keep]), b_g1=fixed_base_g1(ctx, g1, b_s[b_keep]),
        b_g2=fixed_base_g2(ctx, g2, b_s[b_keep]))
    ic = fixed_base_g1(ctx, g1, ic_s)
    params.gamma_g2 = fixed_base_g2(ctx, g2, mont(gamma))
    params.ic = ic
    return params


def params_file_bytes(params, threads=None):
    """The bytes bellman's `Parameters::write` produces for `params` (zinc-vm writes them at
    zinc-vm/src/zvm/arguments/command/setup.rs:67-69 and `zvm prove` reads them back at prove.rs:73-76), as a
    uint8 array: verifying key, then h, l, a, b_g1 (G1) and b_g2 (G2), each a big-endian u32 count and uncompressed
    points.  The point encoding runs in libzg16_tools on the host's threads, so that key files of 2^20 .. 2^24-row
    circuits (0.4 .. 6 GB) are made in seconds; zinc_amd.codec.params_write is the same format point by point."""
    t = tools()
    threads = threads or min(16, os.cpu_count() or 1)
    f = lambda x, w: np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, w)
    parts = [(f(params.alpha_g1, 8), 2, False), (f(params.beta_g1, 8), 2, False), (f(params.beta_g2, 16), 4, False),
             (f(params.gamma_g2, 16), 4, False), (f(params.delta_g1, 8), 2, False), (f(params.delta_g2, 16), 4, False),
             (f(params.ic, 8), 2, True), (f(params.h, 8), 2, True), (f(params.l, 8), 2, True), (f(params.a, 8), 2, True),
             (f(params.b_g1, 8), 2, True), (f(params.b_g2, 16), 4, True)]
    total = sum(arr.shape[0] * 32 * coords + (4 if counted else 0) for arr, coords, counted in parts)
    out = np.zeros(total, dtype=np.uint8)
    pos = 0
    for arr, coords, counted in parts:
        n = arr.shape[0]
        if counted:
            out[pos:pos + 4] = np.frombuffer(int(n).to_bytes(4, "big"), dtype=np.uint8)
            pos += 4
        if n:
            t.zgt_points_encode(_lib.as_u64p(arr), n, coords, C.c_void_p(out.ctypes.data + pos), threads)
        pos += n * 32 * coords
    assert pos == total
    return out
