"""ctypes binding of the C ABI declared in include/secp256k1_zkp_amd.h.

The shared library is built in-tree (``secp256k1_zkp_amd/libsecp256k1_zkp_amd.so``, see ``__graft_entry__.build``)
and loaded from there; there is no Python or CPU implementation behind it -- if the library or a HIP device is
missing the import / engine creation raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S2K_LIB") or os.path.join(_HERE, "libsecp256k1_zkp_amd.so")     # S2K_LIB: diagnostic builds (tools/prof_regions.py)

_c = ctypes
_vp, _sz, _i32p = _c.c_void_p, _c.c_size_t, _c.c_void_p

# name -> (restype, argtypes); must list every S2K_API symbol of include/secp256k1_zkp_amd.h
SIGNATURES = {
    "s2k_engine_create": (_vp, [_c.c_int]),
    "s2k_engine_destroy": (None, [_vp]),
    "s2k_last_error": (_c.c_char_p, []),
    "s2k_last_status": (_c.c_int, []),
    "s2k_clear_status": (None, []),
    "s2k_engine_reserve": (_c.c_int, [_vp, _sz]),
    "s2k_engine_set_option": (_c.c_int, [_vp, _c.c_int, _c.c_long]),
    "s2k_engine_sync": (_c.c_int, [_vp]),
    "s2k_engine_cache_generator": (_c.c_int, [_vp, _vp]),
    "s2k_engine_generator_cached": (_c.c_int, [_vp, _vp]),
    "s2k_engine_gtable": (_vp, [_vp, _c.POINTER(_sz)]),
    "s2k_engine_gtable_bits": (_c.c_int, [_vp]),
    "s2k_engine_gtable_build_ms": (_c.c_float, [_vp]),
    "s2k_engine_last_ms": (_c.c_float, [_vp, _c.c_int]),
    "s2k_engine_last_msm_fallback": (_c.c_int, [_vp]),
    "s2k_engine_rp_handback": (_c.c_int, [_vp, _vp]),
    "s2k_ecmult_batch": (_c.c_int, [_vp] + [_vp] * 6 + [_sz]),
    "s2k_ecmult_batch_dev": (_c.c_int, [_vp, _vp] + [_vp] * 6 + [_sz]),
    "s2k_ecmult_multi": (_c.c_int, [_vp] + [_vp] * 6 + [_sz]),
    "s2k_ecmult_multi_dev": (_c.c_int, [_vp, _vp] + [_vp] * 6 + [_sz]),
    "s2k_ecmult_multi_partial_dev": (_c.c_int, [_vp, _vp] + [_vp] * 5 + [_sz]),
    "s2k_ecmult_multi_many": (_c.c_int, [_vp] + [_vp] * 7 + [_sz]),
    "s2k_ecmult_multi_many_dev": (_c.c_int, [_vp, _vp] + [_vp] * 7 + [_sz]),
    "s2k_gej_sum_dev": (_c.c_int, [_vp, _vp] + [_vp] * 3 + [_sz]),
    "s2k_ecmult_multi_window_partial_dev": (_c.c_int, [_vp, _vp] + [_vp] * 5 + [_sz, _c.c_uint32, _c.c_uint32]),
    "secp256k1_schnorrsig_verify_batch": (_c.c_int, [_vp, _vp, _vp, _vp, _sz, _vp, _c.c_int, _sz]),
    "secp256k1_schnorrsig_verify_batch_dev": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _c.c_int, _sz]),
    "secp256k1_schnorrsig_aggverify_amd": (_c.c_int, [_vp, _vp, _vp, _c.c_int, _vp, _sz, _vp, _sz]),
    "secp256k1_pedersen_verify_tally_batch": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "secp256k1_rangeproof_verify_batch": (_c.c_int, [_vp] + [_vp] * 9 + [_sz]),
    "secp256k1_rangeproof_verify_batch_ptrs": (_c.c_int, [_vp] + [_vp] * 9 + [_sz]),
    "secp256k1_rangeproof_verify_batch_submit": (_c.c_int, [_vp, _vp] + [_vp] * 9 + [_sz]),
    "secp256k1_rangeproof_verify_batch_ptrs_submit": (_c.c_int, [_vp, _vp] + [_vp] * 9 + [_sz]),
    "secp256k1_rangeproof_verify_batch_wait": (_c.c_int, [_vp, _c.c_uint64]),
    "secp256k1_rangeproof_verify_batch_dev": (_c.c_int, [_vp, _vp] + [_vp] * 9 + [_sz]),
    "secp256k1_rangeproof_rewind_batch": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "secp256k1_rangeproof_verify_amd": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "secp256k1_schnorrsig_verify_amd": (_c.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "secp256k1_pedersen_verify_tally_amd": (_c.c_int, [_vp, _vp, _sz, _vp, _sz]),
    "secp256k1_surjectionproof_verify_amd": (_c.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "secp256k1_surjectionproof_verify_batch": (_c.c_int, [_vp] + [_vp] * 6 + [_sz]),
    "secp256k1_surjectionproof_verify_batch_dev": (_c.c_int, [_vp, _vp] + [_vp] * 6 + [_sz]),
    "secp256k1_bppp_norm_product_verify_batch_dev": (_c.c_int, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp, _sz]),
    "secp256k1_schnorrsig_aggverify_dev": (_c.c_int, [_vp, _vp, _vp, _vp, _c.c_int, _vp, _sz, _vp, _sz]),
    "secp256k1_schnorrsig_aggverify_dev_chain": (_c.c_int, [_vp, _vp, _vp, _vp, _c.c_int, _vp, _sz, _vp, _sz, _vp]),
    "s2k_halfagg_chain_states": (_c.c_int, [_vp, _vp, _c.c_int, _vp, _sz, _vp]),
    "secp256k1_pedersen_verify_tally_batch_dev": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "secp256k1_rangeproof_rewind_batch_dev": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "secp256k1_bppp_commit_batch": (_c.c_int, [_vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _sz, _vp, _sz]),
    "secp256k1_bppp_commit_batch_dev": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _sz, _vp, _sz]),
    "secp256k1_bppp_norm_product_verify_batch": (_c.c_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp, _sz]),
    "s2k_group_create": (_vp, [_vp, _c.c_int]),
    "s2k_group_destroy": (None, [_vp]),
    "s2k_group_size": (_c.c_int, [_vp]),
    "s2k_group_engine": (_vp, [_vp, _c.c_int]),
    "secp256k1_rangeproof_verify_batch_group": (_c.c_int, [_vp] + [_vp] * 9 + [_sz]),
    "secp256k1_rangeproof_verify_batch_ptrs_group": (_c.c_int, [_vp] + [_vp] * 9 + [_sz]),
    "secp256k1_schnorrsig_verify_batch_group": (_c.c_int, [_vp, _vp, _vp, _vp, _sz, _vp, _c.c_int, _sz]),
    "s2k_ecmult_multi_group": (_c.c_int, [_vp] + [_vp] * 6 + [_sz]),
    "s2k_ecmult_multi_group_dev": (_c.c_int, [_vp] + [_vp] * 7),
}

_lib = None


def load():
    """Load the native library (once) and attach prototypes. Raises OSError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no fallback implementation.")
        # PyTorch-ROCm wheels bundle their OWN copy of the HIP / HSA runtime and load it by path; this library links the system's
        # (/opt/rocm).  Two HSA runtimes in one process cannot both own the device: whichever initialises second sees "no HIP device".
        # With torch imported FIRST its copy is already mapped under the same SONAME and this library binds to it -- one runtime, and both
        # sides see the GPU (measured on the GPU box in every order: tools/r6_load_order.sh).  Plain C callers never load torch: system runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if os.environ.get("S2K_LIB"):        # an older build of the library in an A/B (tools/ab_probe.py): entry points it lacks stay unbound
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    return load().s2k_last_error().decode()


def sources_sha256():
    """sha256 over the library's sources (csrc/*, sorted by name, and the public header): what tools/profile_round.sh stamps into the
    counter files next to the binary's own hash -- hipcc's output is not reproducible byte for byte.  None when the sources are not there."""
    import hashlib
    csrc = os.path.join(_HERE, "csrc"); hdr = os.path.join(os.path.dirname(_HERE), "include", "secp256k1_zkp_amd.h")
    try:
        h = hashlib.sha256()
        for f in sorted(os.listdir(csrc)):
            if not f.endswith((".h", ".hip")):          # (object directories of build_lib.py live next to the sources)
                continue
            h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
        h.update(open(hdr, "rb").read())
        return h.hexdigest()
    except OSError:
        return None
