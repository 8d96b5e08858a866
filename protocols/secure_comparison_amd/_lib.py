"""ctypes binding of libsc_amd.so (C ABI: include/sc_amd.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SC_AMD_LIB", os.path.join(HERE, "libsc_amd.so"))  # override only for A/B experiments

# every symbol include/sc_amd.h and include/sc_amd_dev.h declare (tests check that the library exports all of them)
SYMBOLS = [
    "sc_ctx_create", "sc_ctx_destroy", "sc_ctx_set_stream", "sc_ctx_synchronize", "sc_last_error", "sc_last_bad_index", "sc_abi_version",
    "sc_malloc", "sc_free", "sc_memcpy_h2d", "sc_memcpy_d2h",
    "sc_mod_create", "sc_mod_words", "sc_exp_create", "sc_const_create", "sc_fbt_create", "sc_fbt_import", "sc_fbt_bytes",
    "sc_modmul", "sc_modmul_const", "sc_modmul_const_sel", "sc_modexp_shared", "sc_modexp_shared_sq", "sc_mod_supports_sq", "sc_modexp_shared_isone", "sc_modexp_shared_isone_any", "sc_fixedbase_pow", "sc_modexp_var", "sc_modexp_var_scatter",
    "sc_modinv", "sc_paillier_encrypt_raw", "sc_paillier_encrypt_raw_neg", "sc_paillier_l_mul", "sc_crt_combine", "sc_plain_alice", "sc_plain_bob", "sc_dgk_step4",
    "sc_paillier_key_create", "sc_paillier_key_mods", "sc_paillier_encrypt", "sc_paillier_randomize", "sc_paillier_decrypt",
    "sc_dgk_key_create", "sc_dgk_key_info", "sc_dgk_randomize", "sc_dgk_encrypt_bits_randomized", "sc_dgk_is_zero", "sc_dgk_any_zero",
    "sc_initiator_step1", "sc_keyholder_step2_4b", "sc_initiator_step4", "sc_initiator_step4i", "sc_keyholder_step4j_5", "sc_initiator_step67",
    "sc_rng_seed", "sc_rng_bits", "sc_rng_below", "sc_rng_coins", "sc_rng_permutations",
    "sc_peak_probe", "sc_mac_counter", "sc_table_traffic_probe", "sc_ctx_set_latency_mode", "sc_ctx_set_onelane_mode", "sc_ctx_set_chip_share", "sc_ctx_set_fork_mode", "sc_ctx_set_pair_policy", "sc_ctx_stats", "sc_ctx_policy", "sc_clock_probe", "sc_comm_unique_id", "sc_comm_init", "sc_allgather", "sc_comm_destroy",
]


ABI_VERSION = 5   # SC_ABI_VERSION of include/sc_amd.h


class ScError(RuntimeError):
    """Raised when a library call fails (HIP error, unsupported size, ...)."""


_lib = None


def load() -> C.CDLL:
    """Load the HIP library.  There is no CPU fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScError(
            f"{LIB_PATH} not found: build it with `python -m protocols.secure_comparison_amd.build` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    vp, i32, u64, i64p = C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_int64)
    ip = C.POINTER(C.c_int)
    sig = {
        "sc_ctx_create": (i32, [i32, C.POINTER(vp)]),
        "sc_ctx_destroy": (None, [vp]),
        "sc_ctx_set_stream": (i32, [vp, vp]),
        "sc_ctx_synchronize": (i32, [vp]),
        "sc_last_error": (C.c_char_p, [vp]),
        "sc_last_bad_index": (C.c_int64, [vp]),
        "sc_abi_version": (i32, []),
        "sc_malloc": (i32, [vp, C.c_size_t, C.POINTER(vp)]),
        "sc_free": (i32, [vp, vp]),
        "sc_memcpy_h2d": (i32, [vp, vp, vp, C.c_size_t]),
        "sc_memcpy_d2h": (i32, [vp, vp, vp, C.c_size_t]),
        "sc_mod_create": (i32, [vp, vp, i32, ip]),
        "sc_mod_words": (i32, [vp, i32]),
        "sc_exp_create": (i32, [vp, vp, i32, ip]),
        "sc_const_create": (i32, [vp, i32, vp, i32, ip]),
        "sc_fbt_create": (i32, [vp, i32, vp, i32, i32, ip]),
        "sc_fbt_import": (i32, [vp, i32, vp, i32, ip]),
        "sc_fbt_bytes": (i32, [vp, i32, C.POINTER(C.c_uint64)]),
        "sc_modmul": (i32, [vp, i32, vp, i32, vp, i32, vp, u64]),
        "sc_modmul_const": (i32, [vp, i32, vp, i32, vp, u64]),
        "sc_modmul_const_sel": (i32, [vp, i32, vp, i32, i32, vp, vp, u64]),
        "sc_modexp_shared": (i32, [vp, i32, i32, vp, i32, vp, vp, u64]),
        "sc_modexp_shared_sq": (i32, [vp, i32, i32, i32, vp, i32, vp, vp, u64]),
        "sc_mod_supports_sq": (i32, [vp, i32]),
        "sc_modexp_shared_isone": (i32, [vp, i32, i32, vp, i32, vp, u64]),
        "sc_modexp_shared_isone_any": (i32, [vp, i32, i32, vp, i32, u64, vp, u64]),
        "sc_fixedbase_pow": (i32, [vp, i32, vp, i32, vp, vp, u64]),
        "sc_modexp_var": (i32, [vp, i32, vp, vp, i32, i32, i32, vp, i32, vp, u64]),
        "sc_modexp_var_scatter": (i32, [vp, i32, vp, vp, i32, i32, i32, vp, i32, vp, vp, u64]),
        "sc_modinv": (i32, [vp, i32, vp, vp, u64, i64p]),
        "sc_paillier_encrypt_raw": (i32, [vp, i32, i32, vp, i32, vp, u64]),
        "sc_paillier_encrypt_raw_neg": (i32, [vp, i32, i32, vp, i32, vp, u64]),
        "sc_paillier_l_mul": (i32, [vp, i32, i32, vp, i32, vp, u64]),
        "sc_crt_combine": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, u64]),
        "sc_plain_alice": (i32, [vp, vp, vp, i32, i32, u64, vp, vp, vp, vp, vp]),
        "sc_plain_bob": (i32, [vp, vp, vp, i32, i32, u64, vp, vp, vp, vp]),
        "sc_dgk_step4": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64]),
        "sc_paillier_key_create": (i32, [vp, vp, i32, vp, vp, i32, i32, ip]),
        "sc_paillier_key_mods": (i32, [vp, i32, ip, ip]),
        "sc_paillier_encrypt": (i32, [vp, i32, vp, i32, i32, vp, u64]),
        "sc_paillier_randomize": (i32, [vp, i32, vp, vp, vp, u64]),
        "sc_paillier_decrypt": (i32, [vp, i32, vp, vp, u64]),
        "sc_dgk_key_create": (i32, [vp, vp, vp, vp, i32, vp, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, i32, ip]),
        "sc_dgk_key_info": (i32, [vp, i32, ip, ip, C.POINTER(C.c_uint64)]),
        "sc_dgk_randomize": (i32, [vp, i32, vp, vp, i32, vp, u64]),
        "sc_dgk_encrypt_bits_randomized": (i32, [vp, i32, vp, vp, i32, vp, u64]),
        "sc_dgk_is_zero": (i32, [vp, i32, vp, vp, u64]),
        "sc_dgk_any_zero": (i32, [vp, i32, vp, i32, u64, vp]),
        "sc_initiator_step1": (i32, [vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, u64]),
        "sc_keyholder_step2_4b": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, u64]),
        "sc_initiator_step4": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, u64]),
        "sc_initiator_step4i": (i32, [vp, i32, i32, vp, vp, i32, vp, vp, i32, i32, vp, u64]),
        "sc_keyholder_step4j_5": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, u64]),
        "sc_initiator_step67": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, u64]),
        "sc_rng_seed": (i32, [vp, vp]),
        "sc_rng_bits": (i32, [vp, i32, vp, u64]),
        "sc_rng_below": (i32, [vp, vp, i32, i32, vp, u64]),
        "sc_rng_coins": (i32, [vp, vp, u64]),
        "sc_rng_permutations": (i32, [vp, i32, vp, u64]),
        "sc_peak_probe": (i32, [vp, C.POINTER(C.c_double)]),
        "sc_mac_counter": (i32, [vp, i32, C.POINTER(C.c_double)]),
        "sc_table_traffic_probe": (i32, [vp, i32, vp, vp, u64, i32, i32, C.POINTER(C.c_int)]),
        "sc_ctx_set_latency_mode": (i32, [vp, i32]),
        "sc_ctx_set_onelane_mode": (i32, [vp, i32]),
        "sc_ctx_set_chip_share": (i32, [vp, i32]),
        "sc_ctx_set_fork_mode": (i32, [vp, i32]),
        "sc_ctx_set_pair_policy": (i32, [vp, C.c_double, C.c_double]),
        "sc_ctx_stats": (i32, [vp, C.POINTER(C.c_uint64), i32]),
        "sc_ctx_policy": (i32, [vp, C.POINTER(C.c_double)]),
        "sc_clock_probe": (i32, [vp, i32, vp, u64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "sc_comm_unique_id": (i32, [vp, vp]),
        "sc_comm_init": (i32, [vp, vp, i32, i32]),
        "sc_allgather": (i32, [vp, vp, vp, u64]),
        "sc_comm_destroy": (i32, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.sc_abi_version() != ABI_VERSION:
        raise ScError(f"{LIB_PATH} speaks ABI version {lib.sc_abi_version()}, this binding expects {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib
