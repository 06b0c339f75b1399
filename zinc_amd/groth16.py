"""Host-side mirror of the bellman Groth16 prover interface that zinc-vm uses.

zinc-vm builds a `bellman::Circuit`, and bellman drives `Circuit::synthesize(&mut cs)` against
its own `ProvingAssignment` before running the FFT / multiexp stages
(zinc-vm/src/core/circuit/facade.rs:145-153, zinc-vm/src/core/circuit/synthesizer.rs:33-38).
This module keeps those names and meanings -- `ProvingAssignment.alloc / alloc_input / enforce`,
`create_proof(circuit, params, r, s)`, `create_random_proof(circuit, params, rng)`,
`SynthesisError` variants -- but the assignment only records; all arithmetic after synthesis runs
in the HIP library through the C ABI (include/zinc_groth16.h).
"""
import ctypes as C
import threading
import secrets

import numpy as np

from . import lib as _lib
from .fields import R_MOD, MONT_R, ints_to_mont, mont_to_ints, g1_to_mont, g2_to_mont, mont_to_g1, mont_to_g2

AUX = 1 << 31


class SynthesisError(_lib.Groth16Error):
    """Failure reported by the prover backend; `.status_name` is the bellman variant name."""


def _check(ctx_handle, status):
    if status != _lib.OK:
        msg = _lib.load().zg16_last_error(ctx_handle).decode() if ctx_handle else ""
        raise SynthesisError(status, msg)


class Context:
    """One GPU: streams, scratch memory, cached NTT tables."""

    def __init__(self, device=0, _default=False):
        self._lib = _lib.load()
        self.handle = C.c_void_p()
        create = self._lib.zg16_default_ctx if _default else self._lib.zg16_ctx_create
        st = create(device, C.byref(self.handle))
        if st != _lib.OK:
            raise SynthesisError(st, "cannot create a context on GPU %d (is an MI355X visible?)" % device)
        self.device = device
        self._owned = not _default

    @classmethod
    def default(cls, device=0):
        """zg16_default_ctx: the process-wide context of a GPU (created on first use, never closed by this
        object), whose cache keeps keys and circuits resident between create_proof calls."""
        return cls(device, _default=True)

    def close(self):
        if self.handle and self._owned:
            self._lib.zg16_ctx_destroy(self.handle)
        self.handle = C.c_void_p()

    def cache_stats(self):
        c = _lib.CacheCounts()
        _check(self.handle, self._lib.zg16_cache_stats(self.handle, C.byref(c)))
        return {n: getattr(c, n) for n, _ in c._fields_}

    def cache_clear(self):
        _check(self.handle, self._lib.zg16_cache_clear(self.handle))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        """zg16_ctx_set_stream: a hipStream_t as an integer (e.g. torch.cuda.Stream.cuda_stream), None = the context's own."""
        _check(self.handle, self._lib.zg16_ctx_set_stream(self.handle, C.c_void_p(hip_stream)))

    def get_stream(self):
        """zg16_ctx_get_stream: the hipStream_t of the context's main-stream work, as an integer."""
        out = C.c_void_p()
        _check(self.handle, self._lib.zg16_ctx_get_stream(self.handle, C.byref(out)))
        return out.value or 0

    def timings(self):
        t = _lib.Timings()
        _check(self.handle, self._lib.zg16_last_timings(self.handle, C.byref(t)))
        return {n: getattr(t, n) for n, _ in t._fields_}

    def profile(self, on=True, reset=True):
        """on: False / True (every kernel kind) / 2 (only the kinds a roofline is reported for)."""
        _check(self.handle, self._lib.zg16_profile_enable(self.handle, int(on)))
        if reset:
            _check(self.handle, self._lib.zg16_profile_reset(self.handle))

    def serial(self, on):
        """True: proves issue every stage on one stream (uncontended per-kernel timings); False: overlapped."""
        _check(self.handle, self._lib.zg16_profile_serial(self.handle, int(on)))

    def graph(self, mode):
        """zg16_graph_mode: True replays every proof as a HIP graph, False never, None the default (off unless
        ZG16_GRAPH=1; replay measured slower than plain multi-stream launches)."""
        _check(self.handle, self._lib.zg16_graph_mode(self.handle, -1 if mode is None else int(bool(mode))))

    def graph_counts(self):
        """(captures, replays, failed captures) since the context was created."""
        c, r, f = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(self.handle, self._lib.zg16_graph_counts(self.handle, C.byref(c), C.byref(r), C.byref(f)))
        return c.value, r.value, f.value

    def kernel_stats(self):
        """{kernel name: (total_ms, launches, units)} since the last reset."""
        out = {}
        for k in range(self._lib.zg16_profile_kernel_count()):
            name, ms, n, units = C.c_char_p(), C.c_double(), C.c_uint64(), C.c_double()
            _check(self.handle, self._lib.zg16_profile_get(self.handle, k, C.byref(name), C.byref(ms), C.byref(n),
                                                           C.byref(units)))
            out[name.value.decode()] = (ms.value, n.value, units.value)
        return out

    # ---- stand-alone stages -------------------------------------------------------------
    def ntt(self, data, inverse=False, coset=False):
        """data: (2^k, 4) uint64 Montgomery Fr; returns the transformed copy."""
        arr = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, 4).copy()
        n = arr.shape[0]
        log_n = n.bit_length() - 1
        assert 1 << log_n == n
        _check(self.handle, self._lib.zg16_ntt(self.handle, arr.ctypes.data_as(C.c_void_p), log_n,
                                               int(inverse), int(coset), 0))
        return arr

    def h_coefficients(self, a, b, c):
        a, b, c = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (a, b, c))
        n_rows = a.shape[0]
        m = 1
        while m < n_rows:
            m *= 2
        h = np.zeros((max(m - 1, 1), 4), dtype=np.uint64)
        m_out = C.c_size_t()
        _check(self.handle, self._lib.zg16_h_coefficients(self.handle, _lib.as_u64p(a), _lib.as_u64p(b),
                                                          _lib.as_u64p(c), n_rows, _lib.as_u64p(h), C.byref(m_out)))
        return h[:m - 1], m_out.value

    def _msm(self, fn, width, bases, scalars, density):
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, width)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(width, dtype=np.uint64)
        dptr = None
        if density is not None:
            density = np.ascontiguousarray(density, dtype=np.uint8)
            assert density.shape[0] == scalars.shape[0]
            dptr = density.ctypes.data_as(_lib.u8p)
        # empty arrays still need a valid pointer type
        bp = _lib.as_u64p(bases) if bases.size else None
        sp = _lib.as_u64p(scalars) if scalars.size else None
        _check(self.handle, fn(self.handle, bp, bases.shape[0], sp, scalars.shape[0], dptr, _lib.as_u64p(out)))
        return out

    def msm_g1(self, bases, scalars, density=None):
        return self._msm(self._lib.zg16_msm_g1, 8, bases, scalars, density)

    def msm_g2(self, bases, scalars, density=None):
        return self._msm(self._lib.zg16_msm_g2, 16, bases, scalars, density)


class ResidentBases:
    """A base set kept in HBM for repeated multiexps (zg16_bases_upload / zg16_msm_resident)."""

    def __init__(self, ctx, group, bases, window_tables=True):
        width = 8 if group == 1 else 16
        arr = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, width)
        self.ctx, self.group, self.n, self.width = ctx, group, arr.shape[0], width
        self.handle = C.c_void_p()
        _check(ctx.handle, ctx._lib.zg16_bases_upload(ctx.handle, group, arr.ctypes.data_as(C.c_void_p), self.n,
                                                      int(window_tables), C.byref(self.handle)))

    def msm(self, scalars=None, device_ptr=None, noalloc=False):
        """Sum of scalars[i] * base[i]; scalars are Montgomery Fr on the host, or `device_ptr` on the GPU.
        noalloc: go through zg16_msm_resident_noalloc (small base sets with window tables only)."""
        fn = self.ctx._lib.zg16_msm_resident_noalloc if noalloc else self.ctx._lib.zg16_msm_resident
        out = np.zeros(self.width, dtype=np.uint64)
        if device_ptr is not None:
            ptr, on_dev = C.c_void_p(device_ptr), 1
        else:
            sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
            assert sc.shape[0] == self.n
            ptr, on_dev = sc.ctypes.data_as(C.c_void_p), 0
        _check(self.ctx.handle, fn(self.ctx.handle, self.handle, ptr, on_dev, _lib.as_u64p(out)))
        return out

    def free(self):
        if self.handle and self.ctx.handle:  # (a handle that outlives its context is not touched again: the library would follow a dangling context pointer)
            self.ctx._lib.zg16_bases_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Proof:
    """bellman `Proof<Bn256>`: three affine points, here as canonical integers."""

    def __init__(self, raw):
        self.raw = raw
        self.a = mont_to_g1(np.frombuffer(bytes(raw.a), dtype=np.uint64))[0]
        self.b = mont_to_g2(np.frombuffer(bytes(raw.b), dtype=np.uint64))[0]
        self.c = mont_to_g1(np.frombuffer(bytes(raw.c), dtype=np.uint64))[0]

    def as_tuple(self):
        return (self.a, self.b, self.c)


class Parameters:
    """bellman `Parameters<Bn256>` as Montgomery limb arrays (what the ABI takes)."""

    def __init__(self, alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, h, l, a, b_g1, b_g2):
        f = lambda x, w: np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, w)
        self.alpha_g1, self.beta_g1, self.delta_g1 = f(alpha_g1, 8), f(beta_g1, 8), f(delta_g1, 8)
        self.beta_g2, self.delta_g2 = f(beta_g2, 16), f(delta_g2, 16)
        self.h, self.l, self.a, self.b_g1 = f(h, 8), f(l, 8), f(a, 8), f(b_g1, 8)
        self.b_g2 = f(b_g2, 16)

    @classmethod
    def from_points(cls, pk):
        """pk: any object with bellman's field names holding affine integer points."""
        return cls(g1_to_mont([pk.alpha_g1]), g1_to_mont([pk.beta_g1]), g2_to_mont([pk.beta_g2]),
                   g1_to_mont([pk.delta_g1]), g2_to_mont([pk.delta_g2]), g1_to_mont(pk.h),
                   g1_to_mont(pk.l), g1_to_mont(pk.a), g1_to_mont(pk.b_g1), g2_to_mont(pk.b_g2))

    def view(self):
        v = _lib.PkView()
        p = lambda a: _lib.as_u64p(a) if a.size else None
        v.alpha_g1, v.beta_g1, v.beta_g2 = p(self.alpha_g1), p(self.beta_g1), p(self.beta_g2)
        v.delta_g1, v.delta_g2 = p(self.delta_g1), p(self.delta_g2)
        v.h, v.n_h = p(self.h), self.h.shape[0]
        v.l, v.n_l = p(self.l), self.l.shape[0]
        v.a, v.n_a = p(self.a), self.a.shape[0]
        v.b_g1, v.n_b_g1 = p(self.b_g1), self.b_g1.shape[0]
        v.b_g2, v.n_b_g2 = p(self.b_g2), self.b_g2.shape[0]
        return v

    def upload(self, ctx, shard_index=0, shard_count=1):
        return DeviceKey(ctx, self, shard_index, shard_count)

    def cached(self, ctx, full_digest=False):
        """zg16_pk_cached: the context's resident handle of this key (uploaded on first sight); returns
        (DeviceKey, was_cached).  The handle belongs to the context; free() on it does nothing."""
        key = DeviceKey(ctx, None)
        key.params = self
        v, hit = self.view(), C.c_int()
        _check(ctx.handle, ctx._lib.zg16_pk_cached(ctx.handle, C.byref(v), int(full_digest), C.byref(key.handle),
                                                   C.byref(hit)))
        key._borrowed = True
        return key, bool(hit.value)


class DeviceKey:
    def __init__(self, ctx, params, shard_index=0, shard_count=1):
        self.ctx, self.params = ctx, params
        self.handle = C.c_void_p()
        if params is None:
            return
        v = params.view()
        _check(ctx.handle, ctx._lib.zg16_pk_upload(ctx.handle, C.byref(v), shard_index, shard_count,
                                                   C.byref(self.handle)))

    @classmethod
    def from_bytes(cls, ctx, data, checked=True, shard_index=0, shard_count=1):
        """bellman `Parameters::read(reader, checked)` on the GPU: `data` is the proving-key file (bytes, or a
        uint8 numpy array -- e.g. an np.memmap of the file -- which is handed over without a copy)."""
        key = cls(ctx, None)
        if isinstance(data, np.ndarray):
            arr = np.ascontiguousarray(data, dtype=np.uint8)
            ptr, size = C.cast(C.c_void_p(arr.ctypes.data), C.c_char_p), arr.size
        else:
            arr = bytes(data)
            ptr, size = arr, len(arr)
        _check(ctx.handle, ctx._lib.zg16_pk_upload_bytes(ctx.handle, ptr, size, int(checked), shard_index,
                                                         shard_count, C.byref(key.handle)))
        return key

    def table_info(self):
        """zg16_pk_table_info: dict of the digit widths per query role, the table step and the HBM bytes held."""
        ch, cw, cg, st, nb = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint(), C.c_size_t()
        _check(self.ctx.handle, self.ctx._lib.zg16_pk_table_info(self.handle, C.byref(ch), C.byref(cw), C.byref(cg),
                                                                 C.byref(st), C.byref(nb)))
        return dict(width_h=ch.value, width_witness=cw.value, width_g2=cg.value, step=st.value, device_bytes=nb.value)

    def vk_view(self):
        v = _lib.PkView()
        _check(self.ctx.handle, self.ctx._lib.zg16_pk_vk_view(self.handle, C.byref(v)))
        return v

    def free(self):
        if self.handle and self.ctx.handle and not getattr(self, "_borrowed", False):
            self.ctx._lib.zg16_pk_free(self.handle)
        self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class R1CS:
    """Recorded constraint matrices (CSR) ready for upload; rows include bellman's input rows."""

    def __init__(self, n_inputs, n_aux, mats):
        """mats: three (row_ptr uint64[n_rows+1], col uint32[nnz], coeff uint64[nnz,4]) triples."""
        self.n_inputs, self.n_aux = n_inputs, n_aux
        self.mats = [(np.ascontiguousarray(rp, dtype=np.uint64), np.ascontiguousarray(col, dtype=np.uint32),
                      np.ascontiguousarray(co, dtype=np.uint64).reshape(-1, 4)) for rp, col, co in mats]
        self.n_rows = self.mats[0][0].shape[0] - 1

    def view(self):
        v = _lib.R1csView()
        v.n_rows, v.n_inputs, v.n_aux = self.n_rows, self.n_inputs, self.n_aux
        for name, (rp, col, co) in zip("abc", self.mats):
            csr = _lib.Csr()
            csr.row_ptr = _lib.as_u64p(rp)
            csr.col = _lib.as_u32p(col) if col.size else None
            csr.coeff = _lib.as_u64p(co) if co.size else None
            setattr(v, name, csr)
        return v

    def upload(self, ctx):
        return DeviceR1CS(ctx, self)

    def cached(self, ctx, circuit_id=None):
        """zg16_r1cs_cached with the matrices in hand: returns (DeviceR1CS, was_cached)."""
        return DeviceR1CS.cached(ctx, circuit_id, self)


class DeviceR1CS:
    def __init__(self, ctx, r1cs):
        self.ctx, self.r1cs = ctx, r1cs
        self.handle = C.c_void_p()
        if r1cs is None:
            return
        v = r1cs.view()
        _check(ctx.handle, ctx._lib.zg16_r1cs_upload(ctx.handle, C.byref(v), C.byref(self.handle)))

    @classmethod
    def cached(cls, ctx, circuit_id, r1cs=None):
        """zg16_r1cs_cached.  circuit_id: bytes naming the circuit (e.g. a digest of bytecode + method), or None to
        have the matrices digested.  r1cs None: look-up only; returns (None, False) for an unknown circuit."""
        self = cls(ctx, None)
        self.r1cs = r1cs
        self._borrowed = True
        hit = C.c_int()
        v = r1cs.view() if r1cs is not None else None
        cid = bytes(circuit_id) if circuit_id is not None else None
        _check(ctx.handle, ctx._lib.zg16_r1cs_cached(ctx.handle, cid, len(cid) if cid else 0,
                                                     C.byref(v) if v is not None else None, C.byref(self.handle),
                                                     C.byref(hit)))
        if not self.handle:
            return None, False
        return self, bool(hit.value)

    def density(self):
        a, bi, ba = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _check(self.ctx.handle, self.ctx._lib.zg16_r1cs_density(self.handle, C.byref(a), C.byref(bi), C.byref(ba)))
        return a.value, bi.value, ba.value

    def shape(self):
        """zg16_r1cs_shape: (n_rows, n_inputs, n_aux) the matrices were uploaded with."""
        n, ni, na = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _check(self.ctx.handle, self.ctx._lib.zg16_r1cs_shape(self.handle, C.byref(n), C.byref(ni), C.byref(na)))
        return n.value, ni.value, na.value

    def spmv(self, z_in, z_aux):
        n = self.r1cs.n_rows
        outs = [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)]
        zi = np.ascontiguousarray(z_in, dtype=np.uint64).reshape(-1, 4)
        za = np.ascontiguousarray(z_aux, dtype=np.uint64).reshape(-1, 4)
        _check(self.ctx.handle, self.ctx._lib.zg16_spmv(
            self.ctx.handle, self.handle, _lib.as_u64p(zi), _lib.as_u64p(za) if za.size else None,
            *[_lib.as_u64p(o) for o in outs]))
        return outs

    def free(self):
        if self.handle and self.ctx.handle and not getattr(self, "_borrowed", False):
            self.ctx._lib.zg16_r1cs_free(self.handle)
        self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ProvingAssignment:
    """Recording constraint system with the call surface of bellman's ProvingAssignment.

    Unlike upstream it does not evaluate rows while synthesizing: it records the three matrices
    in CSR form and the witness, and the GPU evaluates A z, B z, C z.  Linear combinations are
    lists of (variable, coefficient) with canonical integer coefficients; they must already be in
    the canonical form zinc-vm's Dedup wrapper produces (zinc-vm/src/constraint_systems/dedup.rs:50-68).
    """

    def __init__(self):
        self.z_in, self.z_aux = [], []
        self._rp = ([0], [0], [0])
        self._col = ([], [], [])
        self._coeff = ([], [], [])

    @staticmethod
    def one():
        return 0

    def alloc_input(self, value):
        if value is None:
            raise SynthesisError(3, "alloc_input without a value")
        self.z_in.append(int(value) % R_MOD)
        return len(self.z_in) - 1

    def alloc(self, value):
        if value is None:
            raise SynthesisError(3, "alloc without a value")
        self.z_aux.append(int(value) % R_MOD)
        return (len(self.z_aux) - 1) | AUX

    def enforce(self, a_lc, b_lc, c_lc):
        for k, lc in enumerate((a_lc, b_lc, c_lc)):
            for col, coeff in lc:
                self._col[k].append(col)
                self._coeff[k].append(int(coeff) % R_MOD)
            self._rp[k].append(len(self._col[k]))

    def finalize_inputs(self):
        for i in range(len(self.z_in)):
            self.enforce([(i, 1)], [], [])

    def r1cs(self):
        mats = []
        for k in range(3):
            co = ints_to_mont(self._coeff[k], R_MOD) if self._coeff[k] else np.zeros((0, 4), dtype=np.uint64)
            mats.append((np.array(self._rp[k], dtype=np.uint64), np.array(self._col[k], dtype=np.uint32), co))
        return R1CS(len(self.z_in), len(self.z_aux), mats)

    def witness(self):
        zi = ints_to_mont(self.z_in, R_MOD)
        za = ints_to_mont(self.z_aux, R_MOD) if self.z_aux else np.zeros((0, 4), dtype=np.uint64)
        return zi, za


class MultiProver:
    """zg16_multi_*: several GPUs driven from this one process (a context and a host thread per GPU, peer copies
    for the exchanges).  devices may repeat an index: several contexts on one GPU rehearse the schedule."""

    def __init__(self, devices):
        self._lib = _lib.load()
        self.handle = C.c_void_p()
        ids = (C.c_int * len(devices))(*devices)
        st = self._lib.zg16_multi_create(ids, len(devices), C.byref(self.handle))
        if st != _lib.OK:
            raise SynthesisError(st, "cannot create contexts on GPUs %r" % (list(devices),))

    def _check(self, st):
        if st != _lib.OK:
            raise SynthesisError(st, self._lib.zg16_multi_last_error(self.handle).decode())

    def load(self, params, r1cs, sharded_transforms=None):
        """Shard the key over the GPUs and upload the matrices to each.  sharded_transforms: True / False / None
        (the library chooses).  Returns whether the transforms are sharded."""
        pv, rv = params.view(), r1cs.view()
        mode = -1 if sharded_transforms is None else int(bool(sharded_transforms))
        self._check(self._lib.zg16_multi_load(self.handle, C.byref(pv), C.byref(rv), mode))
        return bool(self._lib.zg16_multi_sharded_transforms(self.handle))

    def prove(self, z_in, z_aux, r_mont, s_mont):
        raw = _lib.Proof()
        zi = np.ascontiguousarray(z_in, dtype=np.uint64).reshape(-1, 4)
        za = np.ascontiguousarray(z_aux, dtype=np.uint64).reshape(-1, 4)
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(4)
        s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64).reshape(4)
        self._check(self._lib.zg16_multi_prove(self.handle, zi.ctypes.data_as(C.c_void_p),
                                               za.ctypes.data_as(C.c_void_p) if za.size else None,
                                               _lib.as_u64p(r_mont), _lib.as_u64p(s_mont), C.byref(raw)))
        return Proof(raw)

    def close(self):
        if self.handle:
            self._lib.zg16_multi_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WitnessAssignment(ProvingAssignment):
    """Constraint system for a circuit whose matrices are already resident (DeviceR1CS.cached hit): it keeps the
    witness and drops every row, so synthesis pays for witness generation only.  bellman hands `enforce` three
    closures; a Rust equivalent never calls them, which is where the per-row cost of
    zinc-vm/src/constraint_systems/dedup.rs:30-69 and logging.rs:99-111 goes away."""

    def enforce(self, a_lc, b_lc, c_lc):
        pass

    def finalize_inputs(self):
        pass


def prove_recorded(ctx, dev_key, dev_r1cs, z_in, z_aux, r_mont, s_mont):
    """zg16_prove on uploaded handles; z_in/z_aux/r/s are Montgomery limb arrays."""
    raw = _lib.Proof()
    zi = np.ascontiguousarray(z_in, dtype=np.uint64).reshape(-1, 4)
    za = np.ascontiguousarray(z_aux, dtype=np.uint64).reshape(-1, 4)
    r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(4)
    s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64).reshape(4)
    _check(ctx.handle, ctx._lib.zg16_prove(
        ctx.handle, dev_key.handle, dev_r1cs.handle, zi.ctypes.data_as(C.c_void_p),
        za.ctypes.data_as(C.c_void_p) if za.size else None, 0, _lib.as_u64p(r_mont), _lib.as_u64p(s_mont),
        C.byref(raw)))
    return Proof(raw)


def prove_from_evals(ctx, dev_key, dev_r1cs, a, b, c, z_in, z_aux, r_mont, s_mont):
    """zg16_prove_from_evals: like prove_recorded, with the row evaluations a, b, c ((n_rows, 4) uint64
    Montgomery Fr each, as bellman's ProvingAssignment collects them) supplied by the caller."""
    raw = _lib.Proof()
    ev = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (a, b, c)]
    zi = np.ascontiguousarray(z_in, dtype=np.uint64).reshape(-1, 4)
    za = np.ascontiguousarray(z_aux, dtype=np.uint64).reshape(-1, 4)
    r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(4)
    s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64).reshape(4)
    _check(ctx.handle, ctx._lib.zg16_prove_from_evals(
        ctx.handle, dev_key.handle, dev_r1cs.handle, *(x.ctypes.data_as(C.c_void_p) for x in ev), 0,
        zi.ctypes.data_as(C.c_void_p), za.ctypes.data_as(C.c_void_p) if za.size else None, 0,
        _lib.as_u64p(r_mont), _lib.as_u64p(s_mont), C.byref(raw)))
    return Proof(raw)


# The default context, its caches and the handles they hand out are shared by every thread of the process: one
# create_proof at a time uses them.
_default_ctx_lock = threading.RLock()


def create_proof(circuit, params, r, s, ctx=None, circuit_id=None):
    """bellman `create_proof(circuit, params, r, s)`; r, s are canonical integers in Fr.

    The key and the matrices stay resident in the context's cache (the process-wide default context unless one is
    given), so the second proof with the same `params` skips the upload and table expansion.  circuit_id (bytes,
    e.g. a digest of the bytecode and the method) names the circuit: once its matrices are resident, synthesis runs
    against a WitnessAssignment and records no rows."""
    if ctx is None:
        with _default_ctx_lock:
            return create_proof(circuit, params, r, s, Context.default(), circuit_id)
    mat = None
    if circuit_id is not None:
        mat, _ = DeviceR1CS.cached(ctx, circuit_id)
    pa = WitnessAssignment() if mat is not None else ProvingAssignment()
    pa.alloc_input(1)
    circuit.synthesize(pa)
    pa.finalize_inputs()
    key, _ = params.cached(ctx)
    if mat is None:
        mat, _ = DeviceR1CS.cached(ctx, circuit_id, pa.r1cs())
    # zg16_prove reads as many witness elements as the resident circuit has variables: a circuit_id reused for a
    # circuit of another shape must fail here, not read past the buffers (bellman: AssignmentMissing)
    _, n_in, n_aux = mat.shape()
    if (n_in, n_aux) != (len(pa.z_in), len(pa.z_aux)):
        raise SynthesisError(3, "the circuit resident under this id has %d inputs and %d aux variables, synthesis "
                                "produced %d and %d" % (n_in, n_aux, len(pa.z_in), len(pa.z_aux)))
    zi, za = pa.witness()
    rs = ints_to_mont([r, s], R_MOD)
    return prove_recorded(ctx, key, mat, zi, za, rs[0], rs[1])


def verify_proof(params, proof, public_inputs):
    """bellman `verify_proof(&prepare_verifying_key(&vk), &proof, &public_inputs)` (zinc-vm/src/core/facade.rs:32-38)
    through zg16_verify_proof.  params: anything with alpha_g1, beta_g2, gamma_g2, delta_g2 and ic as Montgomery limb
    arrays (a Parameters from workload.generate_parameters has them); proof: a Proof or a lib.Proof;
    public_inputs: canonical integers, without the leading ONE."""
    f = lambda x, w: np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, w)
    alpha, beta, gamma, delta = f(params.alpha_g1, 8), f(params.beta_g2, 16), f(params.gamma_g2, 16), f(params.delta_g2, 16)
    ic = f(params.ic, 8)
    vk = _lib.VkView(_lib.as_u64p(alpha), _lib.as_u64p(beta), _lib.as_u64p(gamma), _lib.as_u64p(delta), _lib.as_u64p(ic),
                     ic.shape[0])
    raw = proof.raw if isinstance(proof, Proof) else proof
    xs = ints_to_mont([int(x) % R_MOD for x in public_inputs], R_MOD) if len(public_inputs) else np.zeros((0, 4), dtype=np.uint64)
    ok = C.c_int()
    st = _lib.load().zg16_verify_proof(C.byref(vk), C.byref(raw), _lib.as_u64p(xs) if xs.size else None, xs.shape[0],
                                       C.byref(ok))
    if st != _lib.OK:
        raise SynthesisError(st, "zg16_verify_proof")
    return bool(ok.value)


def random_fr(rng=None):
    """ff_ce's `Fr::rand`: 4 x u64, clear the top two bits, retry until below r.  The pattern
    drawn is the Montgomery representation (returned as the canonical value it stands for)."""
    while True:
        raw = 0
        for i in range(4):
            limb = rng.next_u64() if rng is not None else secrets.randbits(64)
            raw |= limb << (64 * i)
        raw &= (1 << 254) - 1
        if raw < R_MOD:
            return raw * pow(MONT_R, -1, R_MOD) % R_MOD


def create_random_proof(circuit, params, rng=None, ctx=None, circuit_id=None):
    """bellman `create_random_proof`: r then s drawn from rng (zinc-vm/src/core/circuit/facade.rs:140-153)."""
    r = random_fr(rng)
    s = random_fr(rng)
    return create_proof(circuit, params, r, s, ctx, circuit_id)
