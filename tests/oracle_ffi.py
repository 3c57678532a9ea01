"""ctypes view of oracle/liboracle.so (TEST INFRASTRUCTURE: the CPU checker, never the product)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class Fp(C.Structure):
    _fields_ = [("l", C.c_uint64 * 6)]


class Fr(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class G1Affine(C.Structure):
    _fields_ = [("x", Fp), ("y", Fp)]


class G1(C.Structure):
    _fields_ = [("x", Fp), ("y", Fp), ("z", Fp)]


class FFTSettings(C.Structure):
    _fields_ = [("max_width", C.c_size_t), ("roots_of_unity", C.POINTER(Fr)),
                ("reverse_roots_of_unity", C.POINTER(Fr)), ("brp_roots_of_unity", C.POINTER(Fr))]


class Settings(C.Structure):
    _fields_ = [("g1_lagrange_brp", C.POINTER(G1Affine)), ("g1_monomial", C.POINTER(G1Affine)),
                ("g2_monomial_bytes", C.POINTER(C.c_uint8)), ("fs", FFTSettings), ("bgmw", C.c_void_p)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    u8p, frp, g1p, afp = C.POINTER(C.c_uint8), C.POINTER(Fr), C.POINTER(G1), C.POINTER(G1Affine)
    sig = {
        "ofr_from_be32": (C.c_int, [frp, C.c_char_p]),
        "ofr_to_be32": (None, [C.c_char_p, frp]),
        "ofr_mul": (None, [frp, frp, frp]),
        "ofr_add": (None, [frp, frp, frp]),
        "ofr_sub": (None, [frp, frp, frp]),
        "ofr_inv": (None, [frp, frp]),
        "ofr_from_u64": (None, [frp, C.c_uint64]),
        "ofr_from_u64_arr": (None, [frp, C.POINTER(C.c_uint64)]),
        "ofr_to_u64_arr": (None, [C.POINTER(C.c_uint64), frp]),
        "ofp_mul": (None, [C.POINTER(Fp)] * 3),
        "ofp_add": (None, [C.POINTER(Fp)] * 3),
        "ofp_sub": (None, [C.POINTER(Fp)] * 3),
        "ofp_inv": (None, [C.POINTER(Fp)] * 2),
        "ofp_from_be48": (C.c_int, [C.POINTER(Fp), C.c_char_p]),
        "ofp_to_be48": (None, [C.c_char_p, C.POINTER(Fp)]),
        "og1_generator": (None, [g1p]),
        "og1_from_affine": (None, [g1p, afp]),
        "og1_to_affine": (None, [afp, g1p]),
        "og1_add_or_dbl": (None, [g1p, g1p, g1p]),
        "og1_dbl": (None, [g1p, g1p]),
        "og1_mul": (None, [g1p, g1p, frp]),
        "og1_equal": (C.c_int, [g1p, g1p]),
        "og1_is_inf": (C.c_int, [g1p]),
        "og1_in_subgroup": (C.c_int, [g1p]),
        "og1_affine_on_curve": (C.c_int, [afp]),
        "og1_uncompress": (C.c_int, [afp, C.c_char_p]),
        "og1_compress": (None, [C.c_char_p, g1p]),
        "opippenger_window_size": (C.c_size_t, [C.c_size_t]),
        "omsm_tiling_pippenger": (None, [g1p, afp, C.c_char_p, C.c_size_t]),
        "og1_lincomb": (None, [g1p, g1p, frp, C.c_size_t]),
        "omsm_affine": (None, [g1p, afp, frp, C.c_size_t]),
        "omsm_naive": (None, [g1p, afp, frp, C.c_size_t]),
        "omsm_affine_mt": (None, [g1p, afp, frp, C.c_size_t, C.c_int]),
        "offt_settings_new": (C.c_int, [C.POINTER(FFTSettings), C.c_uint]),
        "offt_settings_free": (None, [C.POINTER(FFTSettings)]),
        "oscale2_root_of_unity": (None, [C.POINTER(C.c_uint64), C.c_uint]),
        "offt_fr": (C.c_int, [C.POINTER(FFTSettings), frp, frp, C.c_size_t, C.c_int]),
        "offt_fr_slow": (None, [C.POINTER(FFTSettings), frp, frp, C.c_size_t]),
        "odas_fft_extension": (C.c_int, [C.POINTER(FFTSettings), frp, frp, C.c_size_t]),
        "offt_g1": (C.c_int, [C.POINTER(FFTSettings), g1p, g1p, C.c_size_t, C.c_int]),
        "offt_g1_slow": (None, [C.POINTER(FFTSettings), g1p, g1p, C.c_size_t]),
        "osha256": (None, [C.c_char_p, C.c_char_p, C.c_size_t]),
        "oload_trusted_setup_text": (C.c_int, [C.POINTER(Settings), C.c_char_p, C.c_size_t]),
        "ofree_trusted_setup": (None, [C.POINTER(Settings)]),
        "oblob_to_fr": (C.c_int, [frp, C.c_char_p]),
        "oblob_to_kzg_commitment": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(Settings)]),
        "ocompute_challenge": (None, [frp, frp, C.c_char_p]),
        "ocompute_kzg_proof": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Settings)]),
        "ocompute_blob_kzg_proof": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Settings)]),
        "ocompute_cells": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(Settings)]),
        "ocompute_cell_proof": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(Settings)]),
        "oblob_to_kzg_commitment_bgmw": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(Settings)]),
        "obgmw_window_size": (C.c_size_t, [C.c_size_t]),
        "ocompute_r_powers": (C.c_int, [frp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
        "overify_kzg_proof_batch_g1": (C.c_int, [g1p, g1p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


# ---- conversions to/from Python ints (canonical, non-Montgomery) ----
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def fr_from_int(v):
    f = Fr()
    arr = (C.c_uint64 * 4)(*[(v >> (64 * i)) & (2**64 - 1) for i in range(4)])
    lib().ofr_from_u64_arr(C.byref(f), arr)
    return f


def fr_to_int(f):
    arr = (C.c_uint64 * 4)()
    lib().ofr_to_u64_arr(arr, C.byref(f))
    return sum(int(arr[i]) << (64 * i) for i in range(4))


def fp_from_int(v):
    f = Fp()
    assert lib().ofp_from_be48(C.byref(f), v.to_bytes(48, "big")) == 1
    return f


def fp_to_int(f):
    buf = C.create_string_buffer(48)
    lib().ofp_to_be48(buf, C.byref(f))
    return int.from_bytes(buf.raw, "big")


def fr_array(ints):
    arr = (Fr * len(ints))()
    for i, v in enumerate(ints):
        arr[i] = fr_from_int(v)
    return arr


def load_settings(text: bytes):
    s = Settings()
    rc = lib().oload_trusted_setup_text(C.byref(s), text, len(text))
    return rc, s
