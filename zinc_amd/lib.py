"""ctypes binding of libzinc_groth16_hip.so (the C ABI in include/zinc_groth16.h).

There is no CPU fallback: if the HIP library has not been built, importing symbols from here
raises, and every entry point needs a visible MI355X.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libzinc_groth16_hip.so")

OK = 0
STATUS_NAMES = {
    0: "OK",
    1: "PolynomialDegreeTooLarge",
    2: "UnexpectedIdentity",
    3: "AssignmentMissing",
    4: "MalformedKey",
    5: "InvalidArgument",
    6: "DeviceError",
    7: "OutOfMemory",
    8: "UnconstrainedVariable",
}

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class PkView(C.Structure):
    _fields_ = [("alpha_g1", u64p), ("beta_g1", u64p), ("beta_g2", u64p), ("delta_g1", u64p),
                ("delta_g2", u64p),
                ("h", u64p), ("n_h", C.c_size_t), ("l", u64p), ("n_l", C.c_size_t),
                ("a", u64p), ("n_a", C.c_size_t), ("b_g1", u64p), ("n_b_g1", C.c_size_t),
                ("b_g2", u64p), ("n_b_g2", C.c_size_t)]


class Csr(C.Structure):
    _fields_ = [("row_ptr", u64p), ("col", u32p), ("coeff", u64p)]


class R1csView(C.Structure):
    _fields_ = [("n_rows", C.c_size_t), ("n_inputs", C.c_size_t), ("n_aux", C.c_size_t),
                ("a", Csr), ("b", Csr), ("c", Csr)]


class Proof(C.Structure):
    _fields_ = [("a", C.c_uint64 * 8), ("b", C.c_uint64 * 16), ("c", C.c_uint64 * 8)]


class Partial(C.Structure):
    _fields_ = [("hl", C.c_uint64 * 16), ("a", C.c_uint64 * 16), ("b1", C.c_uint64 * 16),
                ("b2", C.c_uint64 * 32)]


class ToxicWaste(C.Structure):
    _fields_ = [("g1", C.c_uint64 * 8), ("g2", C.c_uint64 * 16), ("alpha", C.c_uint64 * 4), ("beta", C.c_uint64 * 4),
                ("gamma", C.c_uint64 * 4), ("delta", C.c_uint64 * 4), ("tau", C.c_uint64 * 4)]


class ParametersOut(C.Structure):
    _fields_ = [("alpha_g1", C.c_uint64 * 8), ("beta_g1", C.c_uint64 * 8), ("beta_g2", C.c_uint64 * 16),
                ("gamma_g2", C.c_uint64 * 16), ("delta_g1", C.c_uint64 * 8), ("delta_g2", C.c_uint64 * 16),
                ("ic", u64p), ("h", u64p), ("l", u64p), ("a", u64p), ("n_a", C.c_size_t), ("b_g1", u64p),
                ("b_g2", u64p), ("n_b", C.c_size_t)]


class VkView(C.Structure):
    _fields_ = [("alpha_g1", u64p), ("beta_g2", u64p), ("gamma_g2", u64p), ("delta_g2", u64p), ("ic", u64p),
                ("n_ic", C.c_size_t)]


class CacheCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("keys", "circuits", "key_hits", "key_misses", "circuit_hits", "circuit_misses", "evictions",
                 "key_bytes")]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("upload_ms", "spmv_ms", "ntt_ms", "msm_g1_ms", "msm_g2_ms", "finish_ms", "total_ms")]


# every symbol include/zinc_groth16.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("zg16_ctx_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("zg16_ctx_destroy", None, [C.c_void_p]),
    ("zg16_last_error", C.c_char_p, [C.c_void_p]),
    ("zg16_ctx_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("zg16_ctx_get_stream", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    ("zg16_pk_upload", C.c_int, [C.c_void_p, C.POINTER(PkView), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("zg16_pk_free", None, [C.c_void_p]),
    ("zg16_pk_upload_bytes", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p)]),
    ("zg16_pk_table_info", C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                     C.POINTER(C.c_uint), C.POINTER(C.c_size_t)]),
    ("zg16_pk_vk_view", C.c_int, [C.c_void_p, C.POINTER(PkView)]),
    ("zg16_default_ctx", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("zg16_shutdown", None, []),
    ("zg16_pk_cached", C.c_int, [C.c_void_p, C.POINTER(PkView), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    ("zg16_pk_id", C.c_uint64, [C.c_void_p]),
    ("zg16_pk_cache_holds", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    ("zg16_r1cs_cached", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(R1csView), C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_int)]),
    ("zg16_cache_clear", C.c_int, [C.c_void_p]),
    ("zg16_cache_stats", C.c_int, [C.c_void_p, C.POINTER(CacheCounts)]),
    ("zg16_r1cs_upload", C.c_int, [C.c_void_p, C.POINTER(R1csView), C.POINTER(C.c_void_p)]),
    ("zg16_r1cs_free", None, [C.c_void_p]),
    ("zg16_r1cs_density", C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("zg16_r1cs_shape", C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("zg16_prove", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                             u64p, u64p, C.POINTER(Proof)]),
    ("zg16_prove_from_evals", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int, u64p, u64p, C.POINTER(Proof)]),
    ("zg16_graph_mode", C.c_int, [C.c_void_p, C.c_int]),
    ("zg16_graph_counts", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("zg16_ntt_cross", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int]),
    ("zg16_prove_partial", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.POINTER(Partial)]),
    ("zg16_proof_encode", C.c_int, [C.POINTER(Proof), C.POINTER(C.c_uint8)]),
    ("zg16_eval_chain", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    ("zg16_h_from_cosets", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]),
    ("zg16_prove_partial_h", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.POINTER(Partial)]),
    ("zg16_r1cs_log_domain", C.c_uint, [C.c_void_p]),
    ("zg16_prove_finish", C.c_int, [C.POINTER(PkView), C.POINTER(Partial), C.c_size_t, u64p, u64p,
                                    C.POINTER(Proof)]),
    ("zg16_pk_upload_cyclic_h", C.c_int, [C.c_void_p, C.POINTER(PkView), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("zg16_hshard_create", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_void_p)]),
    ("zg16_hshard_free", None, [C.c_void_p]),
    ("zg16_hshard_vector_bytes", C.c_size_t, [C.c_void_p]),
    ("zg16_hshard_stage", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_int)]),
    ("zg16_hshard_stream_ordered", C.c_int, [C.c_void_p, C.c_int]),
    ("zg16_hshard_finish", C.c_int, [C.c_void_p, C.POINTER(Partial)]),
    ("zg16_hshard_randomness", C.c_int, [C.c_void_p, u64p, u64p, C.c_int]),
    ("zg16_hshard_assemble", C.c_int, [C.c_void_p, C.POINTER(Partial), C.c_size_t, C.POINTER(Proof)]),
    ("zg16_multi_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    ("zg16_multi_destroy", None, [C.c_void_p]),
    ("zg16_multi_last_error", C.c_char_p, [C.c_void_p]),
    ("zg16_multi_load", C.c_int, [C.c_void_p, C.POINTER(PkView), C.POINTER(R1csView), C.c_int]),
    ("zg16_multi_sharded_transforms", C.c_int, [C.c_void_p]),
    ("zg16_multi_prove", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u64p, u64p, C.POINTER(Proof)]),
    ("zg16_ntt", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int]),
    ("zg16_h_coefficients", C.c_int, [C.c_void_p, u64p, u64p, u64p, C.c_size_t, u64p, C.POINTER(C.c_size_t)]),
    ("zg16_msm_g1", C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p, C.c_size_t, u8p, u64p]),
    ("zg16_msm_g2", C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p, C.c_size_t, u8p, u64p]),
    ("zg16_bases_upload", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    ("zg16_bases_free", None, [C.c_void_p]),
    ("zg16_msm_resident", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, u64p]),
    ("zg16_msm_resident_noalloc", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, u64p]),
    ("zg16_spmv", C.c_int, [C.c_void_p, C.c_void_p, u64p, u64p, u64p, u64p, u64p]),
    ("zg16_fixed_base_g1", C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, u64p]),
    ("zg16_fixed_base_g2", C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, u64p]),
    ("zg16_generate_parameters", C.c_int, [C.c_void_p, C.POINTER(R1csView), C.POINTER(ToxicWaste),
                                           C.POINTER(ParametersOut)]),
    ("zg16_verify_proof", C.c_int, [C.POINTER(VkView), C.POINTER(Proof), u64p, C.c_size_t, C.POINTER(C.c_int)]),
    ("zg16_selftest_field", C.c_int, [C.c_void_p, C.c_int, C.c_int, u64p, u64p, C.c_size_t, u64p]),
    ("zg16_selftest_group", C.c_int, [C.c_void_p, C.c_int, C.c_int, u64p, u64p, C.c_size_t, u64p]),
    ("zg16_profile_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("zg16_profile_serial", C.c_int, [C.c_void_p, C.c_int]),
    ("zg16_profile_reset", C.c_int, [C.c_void_p]),
    ("zg16_profile_kernel_count", C.c_int, []),
    ("zg16_profile_get", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    ("zg16_last_timings", C.c_int, [C.c_void_p, C.POINTER(Timings)]),
]

_lib = None


def load():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build it with `make -C zinc_amd/csrc` (or __graft_entry__.build()); "
                "there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class Groth16Error(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status
        self.status_name = STATUS_NAMES.get(status, str(status))


def as_u64p(arr):
    assert arr.dtype == np.uint64 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(u64p)


def as_u32p(arr):
    assert arr.dtype == np.uint32 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(u32p)
