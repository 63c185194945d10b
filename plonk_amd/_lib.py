"""ctypes binding of libplonk_hip.so (the C ABI in include/plonk_hip.h).

The library is the product; there is no CPU fallback.  Importing this module loads the
shared object and raises if it is missing or does not export the declared entry points.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libplonk_hip.so")

PLONK_OK = 0
PLONK_E_INVALID = -1
PLONK_E_HIP = -2
PLONK_E_NOMEM = -3
PLONK_E_INTERNAL = -4

NTT_INVERSE = 1
NTT_COSET = 2
PROOF_WIRE_BYTES = 624
COMM_ID_BYTES = 128


class PlonkError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("plonk_hip error %d: %s" % (code, message))
        self.code = code


class Fr(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class G1(C.Structure):
    _fields_ = [("x", C.c_uint64 * 6), ("y", C.c_uint64 * 6), ("infinity", C.c_uint8), ("pad", C.c_uint8 * 7)]


class G2(C.Structure):
    _fields_ = [("x_c0", C.c_uint64 * 6), ("x_c1", C.c_uint64 * 6), ("y_c0", C.c_uint64 * 6), ("y_c1", C.c_uint64 * 6),
                ("infinity", C.c_uint8), ("pad", C.c_uint8 * 7)]


class ShardSlices(C.Structure):
    _fields_ = [("srs", C.c_void_p * 3), ("lo", C.c_uint64 * 3)]


class Vk(C.Structure):
    _fields_ = [("n", C.c_uint64)] + [(k, G1) for k in ("q_m_comm", "q_l_comm", "q_r_comm", "q_o_comm", "q_c_comm",
                                                        "s_sigma1_comm", "s_sigma2_comm", "s_sigma3_comm")] + \
               [("omega", Fr), ("k1", Fr), ("k2", Fr)]


class Proof(C.Structure):
    _fields_ = [("a_comm", G1), ("b_comm", G1), ("c_comm", G1), ("z_comm", G1), ("t_lo_comm", G1),
                ("t_mid_comm", G1), ("t_hi_comm", G1),
                ("a_eval", Fr), ("b_eval", Fr), ("c_eval", Fr), ("s_sigma1_eval", Fr), ("s_sigma2_eval", Fr),
                ("z_omega_eval", Fr),
                ("w_zeta_comm", G1), ("w_zeta_omega_comm", G1)]


assert C.sizeof(G1) == 104 and C.sizeof(Fr) == 32

_vp = C.c_void_p
_sz = C.c_size_t
class MsmTiming(C.Structure):
    _fields_ = [("sort_ms", C.c_float), ("big_bucket_ms", C.c_float), ("bucket_ms", C.c_float), ("reduce_ms", C.c_float),
                ("entries", C.c_uint64), ("window_bits", C.c_uint32), ("bucket_sets", C.c_uint32)]


_u32 = C.c_uint32
_i32 = C.c_int32
ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

# name -> (restype, argtypes); kept in the order of include/plonk_hip.h
SIGNATURES = {
    "plonk_ctx_create": (_i32, [_i32, C.POINTER(_vp)]),
    "plonk_ctx_destroy": (None, [_vp]),
    "plonk_last_error": (C.c_char_p, [_vp]),
    "plonk_sync": (_i32, [_vp]),
    "plonk_ctx_stream": (_vp, [_vp]),
    "plonk_dev_alloc": (_i32, [_vp, _sz, C.POINTER(_vp)]),
    "plonk_dev_free": (_i32, [_vp, _vp]),
    "plonk_dev_upload": (_i32, [_vp, _vp, _vp, _sz]),
    "plonk_dev_download": (_i32, [_vp, _vp, _vp, _sz]),
    "plonk_ntt_dev": (_i32, [_vp, _vp, _u32, _u32, _vp, _u32, _sz]),
    "plonk_ntt_sweeps": (_u32, [_u32]),
    "plonk_ntt": (_i32, [_vp, _vp, _u32, _u32, _vp, _u32, _sz]),
    "plonk_ntt_dev_noalloc": (_i32, [_vp, _vp, _u32, _u32, _vp]),
    "plonk_ntt_noalloc_max_log_n": (_u32, []),
    "plonk_srs_upload": (_i32, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "plonk_srs_from_tau": (_i32, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "plonk_srs_from_tau_range": (_i32, [_vp, _vp, _sz, _sz, C.POINTER(_vp)]),
    "plonk_srs_len": (_sz, [_vp]),
    "plonk_srs_download": (_i32, [_vp, _vp, _sz, _sz, _vp]),
    "plonk_srs_free": (None, [_vp]),
    "plonk_commit": (_i32, [_vp, _vp, _vp, _sz, _vp]),
    "plonk_commit_dev": (_i32, [_vp, _vp, _vp, _sz, _vp]),
    "plonk_msm": (_i32, [_vp, _vp, _vp, _sz, _vp]),
    "plonk_poly_eval_dev": (_i32, [_vp, _vp, _sz, _vp, _vp]),
    "plonk_div_by_linear_dev": (_i32, [_vp, _vp, _sz, _vp, _vp]),
    "plonk_open_dev": (_i32, [_vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "plonk_open_batch_dev": (_i32, [_vp, _vp, C.POINTER(_vp), C.POINTER(_sz), _sz, _vp, _vp, _vp, _vp]),
    "plonk_pk_create": (_i32, [_vp, _u32] + [_vp] * 10 + [_vp, _sz, C.POINTER(_vp)]),
    "plonk_pk_free": (None, [_vp]),
    "plonk_pk_commitments": (_i32, [_vp, _vp, _vp, _vp]),
    "plonk_pk_poly": (_i32, [_vp, _vp, _u32, _vp, C.POINTER(_sz)]),
    "plonk_quotient_3wire_dev": (_i32, [_vp] * 11),
    "plonk_prove_3wire": (_i32, [_vp] * 7 + [_sz, _vp, _vp]),
    "plonk_prove_3wire_dev": (_i32, [_vp] * 7 + [_sz, _vp, _vp]),
    "plonk_comm_unique_id": (_i32, [_vp]),
    "plonk_comm_create_rccl": (_i32, [_vp, _vp, _u32, _u32, C.POINTER(_vp)]),
    "plonk_comm_create_callback": (_i32, [_vp, _u32, _u32, _vp, _vp, C.POINTER(_vp)]),
    "plonk_comm_create_loopback": (_i32, [_vp, _u32, _u32, _sz, C.POINTER(_vp)]),
    "plonk_comm_destroy": (None, [_vp]),
    "plonk_comm_stats": (_i32, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "plonk_comm_timing": (_i32, [_vp, _i32, C.POINTER(C.c_double)]),
    "plonk_prove_3wire_sharded": (_i32, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "plonk_prove_3wire_grouped": (_i32, [_vp, _vp, C.POINTER(ShardSlices), _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "plonk_g1_sum": (_i32, [_vp, _sz, _vp]),
    "plonk_msm_window_bits": (_u32, [_sz]),
    "plonk_msm_set_timing": (_i32, [_vp, _i32]),
    "plonk_msm_last_timing": (_i32, [_vp, C.POINTER(MsmTiming)]),
    "plonk_quotient_last_timing": (_i32, [_vp, C.POINTER(C.c_double)]),
    "plonk_g1_compress": (_i32, [_vp, _vp]),
    "plonk_g1_decompress": (_i32, [_vp, _vp]),
    "plonk_proof_deserialize": (_i32, [_vp, _vp]),
    "plonk_fr_to_bytes": (_i32, [_vp, _vp]),
    "plonk_fr_from_bytes": (_i32, [_vp, _vp]),
    "plonk_proof_serialize": (_i32, [_vp, _vp]),
    "plonk_proof_challenges": (_i32, [_vp, _vp, _sz, _vp]),
    "plonk_g2_generator": (_i32, [_vp]),
    "plonk_g2_mul": (_i32, [_vp, _vp, _vp]),
    "plonk_pairing_check": (_i32, [_vp, _vp, _vp, _vp, C.POINTER(_i32)]),
    "plonk_open_verify": (_i32, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_i32)]),
    "plonk_vk_from_pk": (_i32, [_vp, _vp, _vp, _vp]),
    "plonk_verify_3wire": (_i32, [_vp, _vp, _sz, _vp, _vp, _vp, _sz, C.POINTER(_i32)]),
    "plonk_proof_to_json": (_i32, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "plonk_proof_from_json": (_i32, [C.c_char_p, _sz, _vp]),
    "plonk_synthetic_circuit": (_i32, [_u32] + [_vp] * 15),
    "plonk_bench_field_mul": (_i32, [_vp, _u32, C.POINTER(C.c_double)]),
    "plonk_timer_start": (_i32, [_vp]),
    "plonk_timer_stop": (_i32, [_vp, C.POINTER(C.c_double)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "plonk_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C plonk_amd/csrc). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
