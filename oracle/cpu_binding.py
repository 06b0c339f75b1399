"""ctypes binding of liboracle_cpu.so, the C++ restatement of bellman's prover algorithms.

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py as the checker / timed CPU baseline.  Views are laid out exactly like the product ABI's
(include/zinc_groth16.h) so the same numpy arrays feed both sides.
"""
import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "liboracle_cpu.so")
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class PkView(C.Structure):
    _fields_ = [("alpha_g1", u64p), ("beta_g1", u64p), ("beta_g2", u64p), ("delta_g1", u64p), ("delta_g2", u64p),
                ("h", u64p), ("n_h", C.c_size_t), ("l", u64p), ("n_l", C.c_size_t), ("a", u64p), ("n_a", C.c_size_t),
                ("b_g1", u64p), ("n_b_g1", C.c_size_t), ("b_g2", u64p), ("n_b_g2", C.c_size_t)]


class Csr(C.Structure):
    _fields_ = [("row_ptr", u64p), ("col", u32p), ("coeff", u64p)]


class R1csView(C.Structure):
    _fields_ = [("n_rows", C.c_size_t), ("n_inputs", C.c_size_t), ("n_aux", C.c_size_t), ("a", Csr), ("b", Csr), ("c", Csr)]


class Proof(C.Structure):
    _fields_ = [("a", C.c_uint64 * 8), ("b", C.c_uint64 * 16), ("c", C.c_uint64 * 8)]


class Timings(C.Structure):
    _fields_ = [("eval_s", C.c_double), ("fft_s", C.c_double), ("multiexp_s", C.c_double), ("total_s", C.c_double)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s missing: run `make -C oracle/cpu`" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        l.orc_hardware_threads.restype = C.c_uint
        l.orc_fft.argtypes = [u64p, C.c_uint, C.c_int, C.c_int, C.c_uint]
        l.orc_h_coefficients.argtypes = [u64p, u64p, u64p, C.c_size_t, u64p, C.POINTER(C.c_size_t), C.c_uint]
        for n in ("orc_multiexp_g1", "orc_multiexp_g2"):
            getattr(l, n).argtypes = [u64p, C.c_size_t, u64p, C.c_size_t, u8p, u64p, C.c_uint]
        l.orc_eval_rows.argtypes = [C.POINTER(R1csView), u64p, u64p, u64p, u64p, u64p, u8p, u8p, u8p]
        l.orc_prove.argtypes = [C.POINTER(PkView), C.POINTER(R1csView), u64p, u64p, u64p, u64p, C.POINTER(Proof),
                                C.c_uint, C.POINTER(Timings)]
        _lib = l
    return _lib


def _p(arr, t=u64p):
    return arr.ctypes.data_as(t) if arr is not None and arr.size else None


def hardware_threads():
    return int(load().orc_hardware_threads())


def fft(data, inverse=False, coset=False, threads=1):
    arr = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, 4).copy()
    log_n = arr.shape[0].bit_length() - 1
    st = load().orc_fft(_p(arr), log_n, int(inverse), int(coset), threads)
    assert st == 0, st
    return arr


def h_coefficients(a, b, c, threads=1):
    a, b, c = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (a, b, c))
    n = a.shape[0]
    m = 1
    while m < n:
        m *= 2
    h = np.zeros((m, 4), dtype=np.uint64)
    mo = C.c_size_t()
    st = load().orc_h_coefficients(_p(a), _p(b), _p(c), n, _p(h), C.byref(mo), threads)
    assert st == 0, st
    return h[:m - 1], mo.value


def multiexp(group, bases, scalars, density=None, threads=1):
    width = 8 if group == "g1" else 16
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, width)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(width, dtype=np.uint64)
    d = np.ascontiguousarray(density, dtype=np.uint8) if density is not None else None
    fn = load().orc_multiexp_g1 if group == "g1" else load().orc_multiexp_g2
    st = fn(_p(bases), bases.shape[0], _p(scalars), scalars.shape[0], _p(d, u8p) if d is not None else None, _p(out), threads)
    return st, out


def _views(params, r1cs):
    pv = PkView()
    for name in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2"):
        setattr(pv, name, _p(getattr(params, name)))
    for name, cnt in (("h", "n_h"), ("l", "n_l"), ("a", "n_a"), ("b_g1", "n_b_g1"), ("b_g2", "n_b_g2")):
        arr = getattr(params, name)
        setattr(pv, name, _p(arr))
        setattr(pv, cnt, arr.shape[0])
    rv = R1csView()
    rv.n_rows, rv.n_inputs, rv.n_aux = r1cs.n_rows, r1cs.n_inputs, r1cs.n_aux
    for name, (rp, col, co) in zip("abc", r1cs.mats):
        csr = Csr()
        csr.row_ptr, csr.col, csr.coeff = _p(rp), _p(col, u32p), _p(co)
        setattr(rv, name, csr)
    return pv, rv


def eval_rows(r1cs, z_in, z_aux):
    _, rv = _views(type("P", (), dict(alpha_g1=None, beta_g1=None, beta_g2=None, delta_g1=None, delta_g2=None,
                                        h=np.zeros((0, 8), np.uint64), l=np.zeros((0, 8), np.uint64),
                                        a=np.zeros((0, 8), np.uint64), b_g1=np.zeros((0, 8), np.uint64),
                                        b_g2=np.zeros((0, 16), np.uint64)))(), r1cs)
    n = r1cs.n_rows
    outs = [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)]
    da = np.zeros(max(r1cs.n_aux, 1), np.uint8)
    dbi = np.zeros(r1cs.n_inputs, np.uint8)
    dba = np.zeros(max(r1cs.n_aux, 1), np.uint8)
    zi = np.ascontiguousarray(z_in, dtype=np.uint64)
    za = np.ascontiguousarray(z_aux, dtype=np.uint64)
    st = load().orc_eval_rows(C.byref(rv), _p(zi), _p(za), *[_p(o) for o in outs], _p(da, u8p), _p(dbi, u8p), _p(dba, u8p))
    assert st == 0, st
    return outs, (da[:r1cs.n_aux], dbi, dba[:r1cs.n_aux])


def prove(params, r1cs, z_in, z_aux, r_mont, s_mont, threads=1):
    """Returns (status, proof arrays (a[8], b[16], c[8]), timings dict)."""
    pv, rv = _views(params, r1cs)
    zi = np.ascontiguousarray(z_in, dtype=np.uint64)
    za = np.ascontiguousarray(z_aux, dtype=np.uint64)
    r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64)
    s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
    out, tm = Proof(), Timings()
    st = load().orc_prove(C.byref(pv), C.byref(rv), _p(zi), _p(za), _p(r_mont), _p(s_mont), C.byref(out), threads, C.byref(tm))
    arrs = tuple(np.frombuffer(bytes(x), dtype=np.uint64).copy() for x in (out.a, out.b, out.c))
    return st, arrs, {n: getattr(tm, n) for n, _ in tm._fields_}
