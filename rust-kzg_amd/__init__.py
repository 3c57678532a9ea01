"""rust-kzg_amd — MI355X-native KZG hot path (MSM + NTT) behind rust-kzg's own seams.

This package is host-side plumbing only: it loads the C-ABI library
(csrc/libkzg_mi355x.so, built by build.py with hipcc for gfx950) and mirrors the
reference's plug-in interface for the path:

  prepare_multi_scalar_mult / multi_scalar_mult_prepared / multi_scalar_mult
      = blst-sppark/src/lib.rs:8-62 (the three FFI wrappers `rust-kzg-blst` calls)
  FFTSettings.fft_fr / das_fft_extension / fft_g1
      = kzg::FFTFr / kzg::DASExtension / kzg::FFTG1 for FsFFTSettings (blst/src/fft_fr.rs:156-165,
        blst/src/data_availability_sampling.rs:78-100, blst/src/fft_g1.rs:54-83)

There is no CPU fallback: if the library is missing or no GPU is visible the calls raise.
The directory name contains a hyphen; import it with `load()` from __graft_entry__ /
tests (importlib under the module name `rust_kzg_amd`).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KZGAMD_LIB") or os.path.join(HERE, "csrc", "libkzg_mi355x.so")  # KZGAMD_LIB: A/B builds


class KzgAmdError(RuntimeError):
    pass


class RustError(C.Structure):
    _fields_ = [("code", C.c_int), ("message", C.c_char_p)]


class BlstFr(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class BlstFp(C.Structure):
    _fields_ = [("l", C.c_uint64 * 6)]


class BlstP1Affine(C.Structure):
    _fields_ = [("x", BlstFp), ("y", BlstFp)]


class BlstP1(C.Structure):
    _fields_ = [("x", BlstFp), ("y", BlstFp), ("z", BlstFp)]


class BlstFp2(C.Structure):
    _fields_ = [("fp", BlstFp * 2)]


class BlstP2(C.Structure):
    _fields_ = [("x", BlstFp2), ("y", BlstFp2), ("z", BlstFp2)]


_lib = None

# every symbol include/kzg_mi355x.h declares; tests check the library exports all of them
EXPORTS = [
    "kzgamd_config_init", "kzgamd_tuning_keys", "kzgamd_msm_attach_matrix", "kzgamd_msm_matrix_shape", "kzgamd_pin_host_buffer", "kzgamd_unpin_host_buffer", "kzgamd_prepare_msm_ex", "kzgamd_prepare_msm_matrix", "kzgamd_mult_pippenger_matrix",
    "kzgamd_msm_create_device_ex", "kzgamd_ntt_new_ex", "kzgamd_load_trusted_setup_ex", "kzgamd_load_trusted_setup_file_ex",
    "kzgamd_load_trusted_setup_file_multi_ex",
    "prepare_msm", "mult_pippenger_prepared", "mult_pippenger", "free_msm", "mult_pippenger_prepared_batch",
    "kzgamd_msm_prepared_batch_device", "kzgamd_msm_info", "kzgamd_msm_uses_wide_table", "kzgamd_msm_set_profile", "kzgamd_msm_get_profile",
    "kzgamd_device_count", "kzgamd_version", "kzgamd_msm_create_device", "kzgamd_generate_points",
    "kzgamd_ntt_new", "kzgamd_ntt_free", "ntt_fr", "das_fft_extension", "kzgamd_ntt_fr_device", "kzgamd_ntt_roots", "kzgamd_ntt_plan_dump", "kzgamd_ntt_das_plan_dump", "kzgamd_das_fft_extension_device",
    "fft_g1", "kzgamd_fft_g1_batch", "kzgamd_g1_sum",
    "load_trusted_setup", "load_trusted_setup_file", "free_trusted_setup", "blob_to_kzg_commitment",
    "compute_kzg_proof", "compute_blob_kzg_proof", "kzgamd_compute_blob_kzg_proof_batch", "compute_challenge",
    "bytes_to_kzg_commitment", "bytes_from_bls_field", "compute_cells_and_kzg_proofs",
    "recover_cells_and_kzg_proofs", "verify_cell_kzg_proof_batch", "compute_verify_cell_kzg_proof_batch_challenge",
    "kzgamd_compute_cells_and_kzg_proofs_batch", "kzgamd_compute_challenges_and_evaluate_batch",
    "kzgamd_blob_to_kzg_commitment_batch", "kzgamd_blob_to_kzg_commitment_device", "kzgamd_settings_msm_handle",
    "kzgamd_msm_reserve", "kzgamd_msm_device", "kzgamd_set_device", "kzgamd_get_device", "kzgamd_settings_device",
    "kzgamd_settings_reserve", "kzgamd_settings_table_info", "kzgamd_verify_kzg_proof_batch_g1", "kzgamd_verify_blob_kzg_proof_batch_g1",
    "verify_kzg_proof", "verify_blob_kzg_proof", "verify_blob_kzg_proof_batch", "kzgamd_pairings_verify",
    "kzgamd_compute_blob_kzg_proof_device",
    "kzgamd_p2_uncompress", "kzgamd_p2_compress", "kzgamd_p2_generator", "kzgamd_p2_mult", "kzgamd_p2_add",
    "kzgamd_load_trusted_setup_file_multi", "kzgamd_free_trusted_setup_multi", "kzgamd_blob_to_kzg_commitment_batch_multi",
    "kzgamd_compute_blob_kzg_proof_batch_multi", "kzgamd_compute_cells_and_kzg_proofs_batch_multi",
    "kzgamd_verify_blob_kzg_proof_batch_multi", "kzgamd_mult_pippenger_prepared_multi", "kzgamd_shard_range",
]


class CKZGSettings(C.Structure):
    """kzg/src/eth/c_bindings.rs:55-108"""
    _fields_ = [("roots_of_unity", C.c_void_p), ("brp_roots_of_unity", C.c_void_p),
                ("reverse_roots_of_unity", C.c_void_p), ("g1_values_monomial", C.c_void_p),
                ("g1_values_lagrange_brp", C.c_void_p), ("g2_values_monomial", C.c_void_p),
                ("x_ext_fft_columns", C.c_void_p), ("tables", C.c_void_p), ("wbits", C.c_size_t),
                ("scratch_size", C.c_size_t)]


C_KZG_OK, C_KZG_BADARGS, C_KZG_ERROR, C_KZG_MALLOC = 0, 1, 2, 3
BYTES_PER_BLOB = 131072


class KzgAmdConfig(C.Structure):
    """include/kzg_mi355x.h: the configuration every creating entry point's _ex form takes"""
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("table_budget_bytes", C.c_uint64), ("tuning", C.c_char_p)]


NO_TABLES = (1 << 64) - 1


def make_config(device=-1, table_budget_gb=None, tuning=None, no_tables=False):
    """KzgAmdConfig for the `config=` arguments below.  table_budget_gb: HBM each fixed-base table may take (None = the
    library's default); tuning: a dict or "key=value;key=value" string of the measured switches (tuning_keys())."""
    cfg = KzgAmdConfig()
    lib().kzgamd_config_init(C.byref(cfg))
    cfg.device = device
    if no_tables:
        cfg.table_budget_bytes = NO_TABLES
    elif table_budget_gb is not None:
        cfg.table_budget_bytes = max(1, int(table_budget_gb * 1e9))
    if isinstance(tuning, dict):
        tuning = ";".join("%s=%d" % (k, int(v)) for k, v in tuning.items())
    if tuning:
        cfg.tuning = tuning.encode()
    return cfg


def _cfgp(config):
    return C.byref(config) if config is not None else None


def tuning_keys():
    """{name: (default, lo, hi, meaning)} — the table of rust-kzg_amd/csrc/config.h"""
    out = {}
    for line in lib().kzgamd_tuning_keys().decode().splitlines():
        name, d, lo, hi, what = line.split(" ", 4)
        out[name] = (int(d), int(lo), int(hi), what)
    return out


def lib():
    """Load libkzg_mi355x.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KzgAmdError("libkzg_mi355x.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.prepare_msm.restype = vp
    L.prepare_msm.argtypes = [vp, sz]
    cp = C.POINTER(KzgAmdConfig)
    L.kzgamd_config_init.restype = None
    L.kzgamd_config_init.argtypes = [cp]
    L.kzgamd_tuning_keys.restype = C.c_char_p
    L.kzgamd_prepare_msm_ex.restype = vp
    L.kzgamd_prepare_msm_ex.argtypes = [vp, sz, cp]
    L.kzgamd_prepare_msm_matrix.restype = vp
    L.kzgamd_prepare_msm_matrix.argtypes = [vp, sz, sz, cp]
    L.kzgamd_pin_host_buffer.restype = C.c_int
    L.kzgamd_pin_host_buffer.argtypes = [vp, sz]
    L.kzgamd_unpin_host_buffer.restype = C.c_int
    L.kzgamd_unpin_host_buffer.argtypes = [vp]
    L.kzgamd_msm_matrix_shape.restype = C.c_int
    L.kzgamd_msm_matrix_shape.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    L.kzgamd_msm_attach_matrix.restype = RustError
    L.kzgamd_msm_attach_matrix.argtypes = [vp, vp, sz, sz, cp]
    L.kzgamd_mult_pippenger_matrix.restype = RustError
    L.kzgamd_mult_pippenger_matrix.argtypes = [vp, vp, vp, sz]
    L.kzgamd_msm_create_device_ex.restype = vp
    L.kzgamd_msm_create_device_ex.argtypes = [vp, sz, C.c_int, cp]
    L.kzgamd_ntt_new_ex.restype = vp
    L.kzgamd_ntt_new_ex.argtypes = [C.c_uint, cp]
    L.free_msm.restype = None
    L.free_msm.argtypes = [vp]
    L.mult_pippenger_prepared.restype = RustError
    L.mult_pippenger_prepared.argtypes = [vp, vp, sz, vp]
    L.mult_pippenger_prepared_batch.restype = RustError
    L.mult_pippenger_prepared_batch.argtypes = [vp, vp, sz, sz, vp]
    L.mult_pippenger.restype = RustError
    L.mult_pippenger.argtypes = [vp, vp, sz, vp]
    L.kzgamd_msm_prepared_batch_device.restype = RustError
    L.kzgamd_msm_prepared_batch_device.argtypes = [vp, vp, vp, sz, sz, C.c_int, vp]
    L.kzgamd_msm_info.restype = C.c_int
    L.kzgamd_msm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(sz), C.POINTER(sz)]
    L.kzgamd_msm_uses_wide_table.restype = C.c_int
    L.kzgamd_msm_uses_wide_table.argtypes = [vp]
    L.kzgamd_device_count.restype = C.c_int
    L.kzgamd_version.restype = C.c_char_p
    L.kzgamd_msm_set_profile.restype = C.c_int
    L.kzgamd_msm_set_profile.argtypes = [vp, C.c_int]
    L.kzgamd_msm_get_profile.restype = C.c_int
    L.kzgamd_msm_get_profile.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.kzgamd_msm_create_device.restype = vp
    L.kzgamd_msm_create_device.argtypes = [vp, sz, C.c_int]
    L.kzgamd_generate_points.restype = RustError
    L.kzgamd_generate_points.argtypes = [vp, sz, C.c_uint64, vp]
    L.kzgamd_ntt_new.restype = vp
    L.kzgamd_ntt_new.argtypes = [C.c_uint]
    L.kzgamd_ntt_free.restype = None
    L.kzgamd_ntt_free.argtypes = [vp]
    L.ntt_fr.restype = C.c_int
    L.ntt_fr.argtypes = [vp, vp, vp, sz, C.c_int]
    L.das_fft_extension.restype = C.c_int
    L.das_fft_extension.argtypes = [vp, vp, vp, sz]
    L.kzgamd_ntt_fr_device.restype = C.c_int
    L.kzgamd_ntt_fr_device.argtypes = [vp, vp, vp, sz, sz, C.c_int, vp]
    L.kzgamd_das_fft_extension_device.restype = C.c_int
    L.kzgamd_das_fft_extension_device.argtypes = [vp, vp, vp, vp, sz, sz, vp]
    L.kzgamd_g1_sum.restype = None
    L.kzgamd_g1_sum.argtypes = [vp, vp, sz]
    L.fft_g1.restype = C.c_int
    L.fft_g1.argtypes = [vp, vp, vp, sz, C.c_int]
    L.kzgamd_fft_g1_batch.restype = C.c_int
    L.kzgamd_fft_g1_batch.argtypes = [vp, vp, vp, sz, sz, C.c_int]
    L.kzgamd_ntt_roots.restype = C.c_int
    L.kzgamd_ntt_roots.argtypes = [vp, vp, vp, vp]
    sp = C.POINTER(CKZGSettings)
    L.load_trusted_setup.restype = C.c_int
    L.load_trusted_setup.argtypes = [sp, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64,
                                     C.c_uint64]
    L.load_trusted_setup_file.restype = C.c_int
    L.load_trusted_setup_file.argtypes = [sp, vp]
    L.kzgamd_load_trusted_setup_ex.restype = C.c_int
    L.kzgamd_load_trusted_setup_ex.argtypes = [sp, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64,
                                               C.c_uint64, cp]
    L.kzgamd_load_trusted_setup_file_ex.restype = C.c_int
    L.kzgamd_load_trusted_setup_file_ex.argtypes = [sp, vp, cp]
    L.kzgamd_load_trusted_setup_file_multi_ex.restype = C.c_int
    L.kzgamd_load_trusted_setup_file_multi_ex.argtypes = [vp, vp, sz, vp, cp]
    L.free_trusted_setup.restype = None
    L.free_trusted_setup.argtypes = [sp]
    L.kzgamd_compute_challenges_and_evaluate_batch.restype = C.c_int
    L.kzgamd_compute_challenges_and_evaluate_batch.argtypes = [vp, vp, vp, vp, sz, sp]
    L.blob_to_kzg_commitment.restype = C.c_int
    L.blob_to_kzg_commitment.argtypes = [vp, vp, sp]
    L.kzgamd_blob_to_kzg_commitment_batch.restype = C.c_int
    L.kzgamd_blob_to_kzg_commitment_batch.argtypes = [vp, vp, sz, sp]
    L.kzgamd_blob_to_kzg_commitment_device.restype = C.c_int
    L.kzgamd_blob_to_kzg_commitment_device.argtypes = [vp, vp, vp, vp, sz, sp, vp]
    L.compute_kzg_proof.restype = C.c_int
    L.compute_kzg_proof.argtypes = [vp, vp, vp, vp, sp]
    L.compute_blob_kzg_proof.restype = C.c_int
    L.compute_blob_kzg_proof.argtypes = [vp, vp, vp, sp]
    L.kzgamd_compute_blob_kzg_proof_device.restype = C.c_int
    L.kzgamd_compute_blob_kzg_proof_device.argtypes = [vp, vp, vp, vp, vp, sz, sp, vp]
    L.kzgamd_compute_blob_kzg_proof_batch.restype = C.c_int
    L.kzgamd_compute_blob_kzg_proof_batch.argtypes = [vp, vp, vp, sz, sp]
    L.compute_challenge.restype = None
    L.compute_challenge.argtypes = [vp, vp, vp]
    L.bytes_to_kzg_commitment.restype = C.c_int
    L.bytes_to_kzg_commitment.argtypes = [vp, vp]
    L.bytes_from_bls_field.restype = None
    L.bytes_from_bls_field.argtypes = [vp, vp]
    L.compute_cells_and_kzg_proofs.restype = C.c_int
    L.compute_cells_and_kzg_proofs.argtypes = [vp, vp, vp, sp]
    L.kzgamd_compute_cells_and_kzg_proofs_batch.restype = C.c_int
    L.kzgamd_compute_cells_and_kzg_proofs_batch.argtypes = [vp, vp, vp, sz, sp]
    L.kzgamd_settings_msm_handle.restype = vp
    L.kzgamd_settings_msm_handle.argtypes = [sp]
    L.kzgamd_msm_reserve.restype = RustError
    L.kzgamd_msm_reserve.argtypes = [vp, sz, sz, vp]
    L.kzgamd_msm_device.restype = C.c_int
    L.kzgamd_msm_device.argtypes = [vp]
    L.kzgamd_set_device.restype = C.c_int
    L.kzgamd_set_device.argtypes = [C.c_int]
    L.kzgamd_get_device.restype = C.c_int
    L.kzgamd_settings_device.restype = C.c_int
    L.kzgamd_settings_device.argtypes = [sp]
    L.kzgamd_settings_reserve.restype = C.c_int
    L.kzgamd_settings_reserve.argtypes = [sp, sz, vp]
    L.kzgamd_verify_kzg_proof_batch_g1.restype = C.c_int
    L.kzgamd_verify_kzg_proof_batch_g1.argtypes = [vp, vp, vp, vp, vp, vp, sz, sp]
    L.kzgamd_verify_blob_kzg_proof_batch_g1.restype = C.c_int
    L.kzgamd_verify_blob_kzg_proof_batch_g1.argtypes = [vp, vp, vp, vp, vp, sz, sp]
    bp = C.POINTER(C.c_bool)
    L.verify_kzg_proof.restype = C.c_int
    L.verify_kzg_proof.argtypes = [bp, vp, vp, vp, vp, sp]
    L.verify_blob_kzg_proof.restype = C.c_int
    L.verify_blob_kzg_proof.argtypes = [bp, vp, vp, vp, sp]
    L.verify_blob_kzg_proof_batch.restype = C.c_int
    L.verify_blob_kzg_proof_batch.argtypes = [bp, vp, vp, vp, sz, sp]
    L.kzgamd_pairings_verify.restype = C.c_int
    L.kzgamd_pairings_verify.argtypes = [vp, vp, vp, vp]
    L.kzgamd_p2_uncompress.restype = C.c_int
    L.kzgamd_p2_uncompress.argtypes = [vp, vp]
    L.kzgamd_p2_compress.restype = None
    L.kzgamd_p2_compress.argtypes = [vp, vp]
    L.kzgamd_p2_generator.restype = None
    L.kzgamd_p2_generator.argtypes = [vp]
    L.kzgamd_p2_mult.restype = None
    L.kzgamd_p2_mult.argtypes = [vp, vp, vp]
    L.kzgamd_p2_add.restype = None
    L.kzgamd_p2_add.argtypes = [vp, vp, vp]
    L.kzgamd_load_trusted_setup_file_multi.restype = C.c_int
    L.kzgamd_load_trusted_setup_file_multi.argtypes = [vp, vp, sz, vp]
    L.kzgamd_free_trusted_setup_multi.restype = None
    L.kzgamd_free_trusted_setup_multi.argtypes = [vp, sz]
    L.kzgamd_blob_to_kzg_commitment_batch_multi.restype = C.c_int
    L.kzgamd_blob_to_kzg_commitment_batch_multi.argtypes = [vp, vp, sz, vp, sz]
    L.kzgamd_compute_blob_kzg_proof_batch_multi.restype = C.c_int
    L.kzgamd_compute_blob_kzg_proof_batch_multi.argtypes = [vp, vp, vp, sz, vp, sz]
    L.kzgamd_compute_cells_and_kzg_proofs_batch_multi.restype = C.c_int
    L.kzgamd_compute_cells_and_kzg_proofs_batch_multi.argtypes = [vp, vp, vp, sz, vp, sz]
    L.kzgamd_verify_blob_kzg_proof_batch_multi.restype = C.c_int
    L.kzgamd_verify_blob_kzg_proof_batch_multi.argtypes = [bp, vp, vp, vp, sz, vp, sz]
    L.kzgamd_shard_range.restype = C.c_int
    L.kzgamd_shard_range.argtypes = [sz, sz, sz, C.POINTER(sz), C.POINTER(sz)]
    L.kzgamd_mult_pippenger_prepared_multi.restype = RustError
    L.kzgamd_mult_pippenger_prepared_multi.argtypes = [vp, sz, vp, vp, vp]
    _lib = L
    return L


def _check(err, what):
    if err.code != 0:
        msg = err.message.decode() if err.message else "error %d" % err.code
        raise KzgAmdError("%s: %s" % (what, msg))


def _addr(buf):
    return C.cast(buf, C.c_void_p) if not isinstance(buf, int) else C.c_void_p(buf)


class PreparedMsm:
    """Owning handle = the reference's SpparkPrecomputation.table (kzg/src/msm/sppark.rs:5-22)."""

    def __init__(self, points, npoints, config=None):
        self.npoints = npoints
        self.handle = lib().kzgamd_prepare_msm_ex(_addr(points), npoints, _cfgp(config)) if config is not None \
            else lib().prepare_msm(_addr(points), npoints)
        if not self.handle:
            raise KzgAmdError("prepare_msm failed (no GPU, or bad arguments)")

    def attach_matrix(self, points, rows, cols, config=None):
        """kzgamd_msm_attach_matrix: this handle also answers multiply_batch (the reference's precompute(points, matrix))"""
        _check(lib().kzgamd_msm_attach_matrix(self.handle, _addr(points), rows, cols, _cfgp(config)), "kzgamd_msm_attach_matrix")
        self.rows, self.cols = rows, cols

    def multiply_batch(self, scalars, nmat=1):
        """scalars: blst_fr[nmat * rows * cols] (Montgomery) -> (BlstP1 * (nmat * rows)); needs a matrix (attach_matrix / MatrixMsm)"""
        out = (BlstP1 * (nmat * self.rows))()
        _check(lib().kzgamd_mult_pippenger_matrix(self.handle, out, _addr(scalars), nmat), "kzgamd_mult_pippenger_matrix")
        return out

    def info(self):
        c, rows, nb, n = C.c_int(), C.c_int(), C.c_size_t(), C.c_size_t()
        lib().kzgamd_msm_info(self.handle, C.byref(c), C.byref(rows), C.byref(nb), C.byref(n))
        wide = lib().kzgamd_msm_uses_wide_table(self.handle)
        return {"window_bits": c.value, "rows": rows.value, "nbuckets": nb.value, "npoints": n.value,
                "wide_table": bool(wide), "wide_glv": wide == 2,
                "adds_per_scalar": rows.value * (2 if wide == 2 else 1)}

    def close(self):
        if self.handle:
            lib().free_msm(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prepare_multi_scalar_mult(points, npoints, config=None):
    """blst-sppark/src/lib.rs:8-17.  points: ctypes array / buffer of blst_p1_affine."""
    return PreparedMsm(points, npoints, config)


class MatrixMsm(PreparedMsm):
    """rows base sets of cols points in one table: the precomputation behind G1LinComb::g1_lincomb_batch
    (kzg/src/lib.rs:156-181 -> BgmwTable::multiply_batch, kzg/src/msm/bgmw.rs:306-380).  points[r * cols + c]."""

    def __init__(self, points, rows, cols, config=None):
        self.rows, self.cols, self.npoints = rows, cols, rows * cols
        self.handle = lib().kzgamd_prepare_msm_matrix(_addr(points), rows, cols, _cfgp(config))
        if not self.handle:
            raise KzgAmdError("kzgamd_prepare_msm_matrix failed (no GPU, bad arguments, or no wide table fits the budget)")



def multi_scalar_mult_prepared(msm, scalars, npoints):
    """blst-sppark/src/lib.rs:19-38.  scalars: blst_fr[npoints] (Montgomery).  Returns BlstP1."""
    out = BlstP1()
    _check(lib().mult_pippenger_prepared(msm.handle, C.byref(out), npoints, _addr(scalars)), "mult_pippenger_prepared")
    return out


def multi_scalar_mult_prepared_batch(msm, scalars, npoints, nbatch):
    out = (BlstP1 * nbatch)()
    _check(lib().mult_pippenger_prepared_batch(msm.handle, out, npoints, nbatch, _addr(scalars)),
           "mult_pippenger_prepared_batch")
    return out


def multi_scalar_mult(points, scalars, npoints):
    """blst-sppark/src/lib.rs:40-62."""
    out = BlstP1()
    _check(lib().mult_pippenger(C.byref(out), _addr(points), npoints, _addr(scalars)), "mult_pippenger")
    return out


def msm_prepared_batch_device(msm, d_out, d_scalars, npoints, nbatch, scalars_mont=True, stream=0):
    """Device-resident form: raw device pointers (ints), enqueued on `stream` (hipStream_t as int)."""
    _check(lib().kzgamd_msm_prepared_batch_device(msm.handle, C.c_void_p(d_out), C.c_void_p(d_scalars), npoints, nbatch,
                                                  1 if scalars_mont else 0, C.c_void_p(stream)),
           "kzgamd_msm_prepared_batch_device")


def device_count():
    return lib().kzgamd_device_count()


def set_device(device):
    """GPU for handles created afterwards by this thread (kzgamd_set_device)."""
    if lib().kzgamd_set_device(device) != 0:
        raise KzgAmdError("kzgamd_set_device(%d) failed" % device)


def get_device():
    return lib().kzgamd_get_device()


def msm_reserve(handle, npoints, nbatch, stream=0):
    _check(lib().kzgamd_msm_reserve(C.c_void_p(handle), npoints, nbatch, C.c_void_p(stream)), "kzgamd_msm_reserve")


# ---------------------------------------------------------------- c-kzg-4844 surface (B3)
_libc = None


def _fopen(path):
    global _libc
    if _libc is None:
        _libc = C.CDLL(None)
        _libc.fopen.restype = C.c_void_p
        _libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        _libc.fclose.argtypes = [C.c_void_p]
    f = _libc.fopen(os.fsencode(path), b"r")
    if not f:
        raise FileNotFoundError(path)
    return f


class KZGSettings:
    """Owns a CKZGSettings loaded through the library's own load_trusted_setup(_file)."""

    def __init__(self):
        self.c = CKZGSettings()
        self.loaded = False

    @classmethod
    def from_file(cls, path, config=None):
        self = cls()
        self._config = config  # keeps the tuning string alive for the duration of the call
        f = _fopen(path)
        try:
            rc = lib().kzgamd_load_trusted_setup_file_ex(C.byref(self.c), f, _cfgp(config)) if config is not None \
                else lib().load_trusted_setup_file(C.byref(self.c), f)
        finally:
            _libc.fclose(f)
        if rc != C_KZG_OK:
            raise KzgAmdError("load_trusted_setup_file: C_KZG_RET %d" % rc)
        self.loaded = True
        return self

    @classmethod
    def from_bytes(cls, g1_monomial, g1_lagrange, g2_monomial, config=None):
        self = cls()
        if config is not None:
            rc = lib().kzgamd_load_trusted_setup_ex(C.byref(self.c), g1_monomial, len(g1_monomial), g1_lagrange, len(g1_lagrange),
                                                    g2_monomial, len(g2_monomial), 0, _cfgp(config))
        else:
            rc = lib().load_trusted_setup(C.byref(self.c), g1_monomial, len(g1_monomial), g1_lagrange, len(g1_lagrange),
                                          g2_monomial, len(g2_monomial), 0)
        if rc != C_KZG_OK:
            raise KzgAmdError("load_trusted_setup: C_KZG_RET %d" % rc)
        self.loaded = True
        return self

    def msm_handle(self):
        return lib().kzgamd_settings_msm_handle(C.byref(self.c))

    def device(self):
        return lib().kzgamd_settings_device(C.byref(self.c))

    def reserve(self, nblobs, stream=0):
        rc = lib().kzgamd_settings_reserve(C.byref(self.c), nblobs, C.c_void_p(stream))
        if rc != C_KZG_OK:
            raise KzgAmdError("kzgamd_settings_reserve: C_KZG_RET %d" % rc)

    def g1_lagrange_brp(self):
        return (BlstP1 * 4096).from_address(self.c.g1_values_lagrange_brp)

    def g1_lagrange_affine(self):
        """the same points as blst_p1_affine (a loaded setup's points have Z = 1: x and y are the affine coordinates)"""
        jac = self.g1_lagrange_brp()
        aff = (BlstP1Affine * 4096)()
        for i in range(4096):
            aff[i].x, aff[i].y = jac[i].x, jac[i].y
        return aff

    def close(self):
        if self.loaded:
            lib().free_trusted_setup(C.byref(self.c))
            self.loaded = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def blob_to_kzg_commitment(blob: bytes, settings: KZGSettings) -> bytes:
    """blst/src/eip_4844.rs:163-175; raises KzgAmdError on C_KZG_BADARGS."""
    if len(blob) != BYTES_PER_BLOB:
        raise KzgAmdError("blob_to_kzg_commitment: C_KZG_RET %d" % C_KZG_BADARGS)
    out = C.create_string_buffer(48)
    rc = lib().blob_to_kzg_commitment(out, blob, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("blob_to_kzg_commitment: C_KZG_RET %d" % rc)
    return out.raw


def blob_to_kzg_commitment_batch(blobs: bytes, n: int, settings: KZGSettings):
    out = C.create_string_buffer(48 * n)
    rc = lib().kzgamd_blob_to_kzg_commitment_batch(out, blobs, n, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("kzgamd_blob_to_kzg_commitment_batch: C_KZG_RET %d" % rc)
    raw = out.raw  # one copy (out.raw copies the whole buffer at every access)
    return [raw[48 * i:48 * i + 48] for i in range(n)]


def blob_to_kzg_commitment_device(d_out, d_status, d_scratch, d_blobs, n, settings, stream=0):
    rc = lib().kzgamd_blob_to_kzg_commitment_device(C.c_void_p(d_out), C.c_void_p(d_status), C.c_void_p(d_scratch),
                                                    C.c_void_p(d_blobs), n, C.byref(settings.c), C.c_void_p(stream))
    if rc != C_KZG_OK:
        raise KzgAmdError("kzgamd_blob_to_kzg_commitment_device: C_KZG_RET %d" % rc)


PROOF_SCRATCH_BYTES = 131072 + 64


def compute_blob_kzg_proof_device(d_proofs, d_status, d_scratch, d_blobs, d_commitments, n, settings, stream=0):
    """device pointers (ints); d_scratch = n * PROOF_SCRATCH_BYTES; enqueued on `stream`, not synchronised"""
    rc = lib().kzgamd_compute_blob_kzg_proof_device(C.c_void_p(d_proofs), C.c_void_p(d_status), C.c_void_p(d_scratch),
                                                    C.c_void_p(d_blobs), C.c_void_p(d_commitments), n, C.byref(settings.c),
                                                    C.c_void_p(stream))
    if rc != C_KZG_OK:
        raise KzgAmdError("kzgamd_compute_blob_kzg_proof_device: C_KZG_RET %d" % rc)


def msm_set_profile(handle, on=True):
    lib().kzgamd_msm_set_profile(C.c_void_p(handle), 1 if on else 0)


def msm_get_profile(handle):
    """(avg k_accum ms, avg whole-enqueue ms, enqueues averaged) or None"""
    a, t = C.c_float(), C.c_float()
    cnt = lib().kzgamd_msm_get_profile(C.c_void_p(handle), C.byref(a), C.byref(t))
    if cnt <= 0:
        return None
    return a.value, t.value, cnt


class DeviceMsm(PreparedMsm):
    """Handle over device-resident bases (kzgamd_msm_create_device)."""

    def __init__(self, d_points, npoints, prepare, config=None):
        self.npoints = npoints
        self.handle = lib().kzgamd_msm_create_device_ex(C.c_void_p(d_points), npoints, 1 if prepare else 0, _cfgp(config))
        if not self.handle:
            raise KzgAmdError("kzgamd_msm_create_device failed")


def generate_points(d_out, npoints, seed, stream=0):
    _check(lib().kzgamd_generate_points(C.c_void_p(d_out), npoints, seed, C.c_void_p(stream)), "kzgamd_generate_points")


# ---------------------------------------------------------------- NTT plug-in (B2)
def g1_sum(points):
    """Sum of Jacobian blst_p1 values given as 144-byte strings (host arithmetic; the combine step of msm_sharded)."""
    n = len(points)
    buf = (BlstP1 * max(n, 1))()
    for i, pt in enumerate(points):
        C.memmove(C.byref(buf[i]), bytes(pt), 144)
    out = BlstP1()
    lib().kzgamd_g1_sum(C.byref(out), buf, n)
    return bytes(out)


class FFTSettings:
    """Mirror of FsFFTSettings (blst/src/types/fft_settings.rs:14-58) + FFTFr / DASExtension
    (blst/src/fft_fr.rs:156-165, blst/src/data_availability_sampling.rs:78-100).
    Errors are raised with the reference's messages."""

    def __init__(self, scale, config=None):
        if scale >= 32:
            raise KzgAmdError("Scale is expected to be within root of unity matrix row size")
        self.scale = scale
        self.max_width = 1 << scale
        self.handle = lib().kzgamd_ntt_new_ex(scale, _cfgp(config)) if config is not None else lib().kzgamd_ntt_new(scale)
        if not self.handle:
            raise KzgAmdError("kzgamd_ntt_new failed (no GPU?)")

    def fft_fr(self, data, n, inverse=False):
        """data: blst_fr[n] (ctypes array / buffer). Returns a new (BlstFr * n)."""
        out = (BlstFr * max(n, 1))()
        rc = lib().ntt_fr(self.handle, out, _addr(data), n, 1 if inverse else 0)
        if rc == 1:
            raise KzgAmdError("Supplied list is longer than the available max width")
        if rc == 2:
            raise KzgAmdError("A list with power-of-two length expected")
        if rc != 0:
            raise KzgAmdError("ntt_fr: device error %d" % rc)
        return out

    def fft_g1(self, data, n, inverse=False, nbatch=1):
        """FFTG1::fft_g1 (blst/src/fft_g1.rs:54-83). data: blst_p1[n * nbatch]. Returns a new (BlstP1 * (n * nbatch))."""
        out = (BlstP1 * max(n * nbatch, 1))()
        if nbatch == 1:
            rc = lib().fft_g1(self.handle, out, _addr(data), n, 1 if inverse else 0)
        else:
            rc = lib().kzgamd_fft_g1_batch(self.handle, out, _addr(data), n, nbatch, 1 if inverse else 0)
        if rc == 1:
            raise KzgAmdError("Supplied list is longer than the available max width")
        if rc == 2:
            raise KzgAmdError("A list with power-of-two length expected")
        if rc != 0:
            raise KzgAmdError("fft_g1: device error %d" % rc)
        return out

    def das_fft_extension(self, evens, n):
        out = (BlstFr * max(n, 1))()
        rc = lib().das_fft_extension(self.handle, out, _addr(evens), n)
        if rc == 1:
            raise KzgAmdError("A non-zero list ab expected")
        if rc == 2:
            raise KzgAmdError("A list with power-of-two length expected")
        if rc == 3:
            raise KzgAmdError("Supplied list is longer than the available max width")
        if rc != 0:
            raise KzgAmdError("das_fft_extension: device error %d" % rc)
        return out

    def fft_fr_device(self, d_out, d_in, n, nbatch=1, inverse=False, stream=0):
        rc = lib().kzgamd_ntt_fr_device(self.handle, C.c_void_p(d_out), C.c_void_p(d_in), n, nbatch, 1 if inverse else 0,
                                        C.c_void_p(stream))
        if rc != 0:
            raise KzgAmdError("kzgamd_ntt_fr_device: %d" % rc)

    def das_fft_extension_device(self, d_odds, d_evens, d_scratch, half_n, nbatch=1, stream=0):
        rc = lib().kzgamd_das_fft_extension_device(self.handle, C.c_void_p(d_odds), C.c_void_p(d_evens), C.c_void_p(d_scratch),
                                                   half_n, nbatch, C.c_void_p(stream))
        if rc != 0:
            raise KzgAmdError("kzgamd_das_fft_extension_device: %d" % rc)

    def roots(self):
        W = self.max_width
        r, rr, br = (BlstFr * (W + 1))(), (BlstFr * (W + 1))(), (BlstFr * W)()
        lib().kzgamd_ntt_roots(self.handle, r, rr, br)
        return r, rr, br

    def close(self):
        if self.handle:
            lib().kzgamd_ntt_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def compute_kzg_proof(blob: bytes, z: bytes, settings: KZGSettings):
    """blst/src/eip_4844.rs:476-496 -> (proof48, y32)"""
    if len(blob) != BYTES_PER_BLOB or len(z) != 32:
        raise KzgAmdError("compute_kzg_proof: C_KZG_RET %d" % C_KZG_BADARGS)
    proof, y = C.create_string_buffer(48), C.create_string_buffer(32)
    rc = lib().compute_kzg_proof(proof, y, blob, z, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("compute_kzg_proof: C_KZG_RET %d" % rc)
    return proof.raw, y.raw


def compute_blob_kzg_proof(blob: bytes, commitment: bytes, settings: KZGSettings) -> bytes:
    """blst/src/eip_4844.rs:274-291"""
    if len(blob) != BYTES_PER_BLOB or len(commitment) != 48:
        raise KzgAmdError("compute_blob_kzg_proof: C_KZG_RET %d" % C_KZG_BADARGS)
    proof = C.create_string_buffer(48)
    rc = lib().compute_blob_kzg_proof(proof, blob, commitment, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("compute_blob_kzg_proof: C_KZG_RET %d" % rc)
    return proof.raw


def compute_blob_kzg_proof_batch(blobs: bytes, commitments: bytes, n: int, settings: KZGSettings):
    """New batched API (BASELINE.json configs[4]); equals n compute_blob_kzg_proof calls."""
    out = C.create_string_buffer(48 * n)
    rc = lib().kzgamd_compute_blob_kzg_proof_batch(out, blobs, commitments, n, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("kzgamd_compute_blob_kzg_proof_batch: C_KZG_RET %d" % rc)
    raw = out.raw  # one copy (out.raw copies the whole buffer at every access)
    return [raw[48 * i:48 * i + 48] for i in range(n)]


def compute_challenges_and_evaluate_batch(blobs: bytes, commitments: bytes, n: int, settings: KZGSettings):
    """compute_challenges_and_evaluate_polynomial (kzg/src/eip_4844.rs:690-719): per blob the Fiat-Shamir challenge z
    and y = p(z), 32-byte big-endian each — the field work of verify_blob_kzg_proof_batch."""
    zs, ys = C.create_string_buffer(32 * n), C.create_string_buffer(32 * n)
    rc = lib().kzgamd_compute_challenges_and_evaluate_batch(zs, ys, blobs, commitments, n, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("kzgamd_compute_challenges_and_evaluate_batch: C_KZG_RET %d" % rc)
    zr, yr = zs.raw, ys.raw
    return ([zr[32 * i:32 * i + 32] for i in range(n)], [yr[32 * i:32 * i + 32] for i in range(n)])


def verify_kzg_proof_batch_g1(commitments: bytes, zs: bytes, ys: bytes, proofs: bytes, n: int, settings: KZGSettings):
    """verify_kzg_proof_batch (kzg/src/eip_4844.rs:380-435) up to the pairing -> (proof_lincomb, rhs) as BlstP1."""
    a, b = BlstP1(), BlstP1()
    rc = lib().kzgamd_verify_kzg_proof_batch_g1(C.byref(a), C.byref(b), commitments, zs, ys, proofs, n, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("kzgamd_verify_kzg_proof_batch_g1: C_KZG_RET %d" % rc)
    return a, b


def verify_blob_kzg_proof_batch_g1(blobs: bytes, commitments: bytes, proofs: bytes, n: int, settings: KZGSettings):
    """verify_blob_kzg_proof_batch (kzg/src/eip_4844.rs:736-832) up to the pairing -> (proof_lincomb, rhs)."""
    a, b = BlstP1(), BlstP1()
    rc = lib().kzgamd_verify_blob_kzg_proof_batch_g1(C.byref(a), C.byref(b), blobs, commitments, proofs, n, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("kzgamd_verify_blob_kzg_proof_batch_g1: C_KZG_RET %d" % rc)
    return a, b


def _verdict(rc, ok, what):
    if rc != C_KZG_OK:
        raise KzgAmdError("%s: C_KZG_RET %d" % (what, rc))
    return bool(ok.value)


def verify_kzg_proof(commitment: bytes, z: bytes, y: bytes, proof: bytes, settings: KZGSettings) -> bool:
    """blst/src/eip_4844.rs:383-405"""
    if len(commitment) != 48 or len(z) != 32 or len(y) != 32 or len(proof) != 48:
        raise KzgAmdError("verify_kzg_proof: C_KZG_RET %d" % C_KZG_BADARGS)
    ok = C.c_bool(False)
    return _verdict(lib().verify_kzg_proof(C.byref(ok), commitment, z, y, proof, C.byref(settings.c)), ok, "verify_kzg_proof")


def verify_blob_kzg_proof(blob: bytes, commitment: bytes, proof: bytes, settings: KZGSettings) -> bool:
    """blst/src/eip_4844.rs:410-430"""
    if len(blob) != BYTES_PER_BLOB or len(commitment) != 48 or len(proof) != 48:
        raise KzgAmdError("verify_blob_kzg_proof: C_KZG_RET %d" % C_KZG_BADARGS)
    ok = C.c_bool(False)
    return _verdict(lib().verify_blob_kzg_proof(C.byref(ok), blob, commitment, proof, C.byref(settings.c)), ok,
                    "verify_blob_kzg_proof")


def verify_blob_kzg_proof_batch(blobs, commitments, proofs, settings: KZGSettings) -> bool:
    """blst/src/eip_4844.rs:435-471; lists of bytes.  Length mismatches are the reference's 'Invalid amount of arguments'."""
    n = len(blobs)
    if len(commitments) != n or len(proofs) != n or any(len(b) != BYTES_PER_BLOB for b in blobs) or \
            any(len(c) != 48 for c in commitments) or any(len(p) != 48 for p in proofs):
        raise KzgAmdError("verify_blob_kzg_proof_batch: C_KZG_RET %d" % C_KZG_BADARGS)
    ok = C.c_bool(False)
    return _verdict(lib().verify_blob_kzg_proof_batch(C.byref(ok), b"".join(blobs), b"".join(commitments), b"".join(proofs), n,
                                                      C.byref(settings.c)), ok, "verify_blob_kzg_proof_batch")


def pairings_verify(a1, a2, b1, b2) -> bool:
    """blst/src/kzg_proofs.rs:73-100 on BlstP1 / BlstP2 values (host arithmetic, no GPU)."""
    rc = lib().kzgamd_pairings_verify(C.byref(a1), C.byref(a2), C.byref(b1), C.byref(b2))
    if rc < 0:
        raise KzgAmdError("kzgamd_pairings_verify")
    return rc == 1


def p2_uncompress(b: bytes) -> BlstP2:
    out = BlstP2()
    if len(b) != 96 or lib().kzgamd_p2_uncompress(C.byref(out), b) != 0:
        raise KzgAmdError("kzgamd_p2_uncompress: invalid G2 encoding")
    return out


def p2_compress(p) -> bytes:
    out = C.create_string_buffer(96)
    lib().kzgamd_p2_compress(out, C.byref(p))
    return out.raw


def p2_generator() -> BlstP2:
    out = BlstP2()
    lib().kzgamd_p2_generator(C.byref(out))
    return out


def p2_mult(p, fr) -> BlstP2:
    out = BlstP2()
    lib().kzgamd_p2_mult(C.byref(out), C.byref(p), C.byref(fr))
    return out


def p2_add(a, b) -> BlstP2:
    out = BlstP2()
    lib().kzgamd_p2_add(C.byref(out), C.byref(a), C.byref(b))
    return out


def compute_challenge(blob: bytes, commitment_p1) -> BlstFr:
    """blst/src/eip_4844.rs:501-514 (commitment: BlstP1)"""
    out = BlstFr()
    lib().compute_challenge(C.byref(out), blob, C.byref(commitment_p1))
    return out


def bytes_to_kzg_commitment(b: bytes) -> BlstP1:
    out = BlstP1()
    if len(b) != 48 or lib().bytes_to_kzg_commitment(C.byref(out), b) != C_KZG_OK:
        raise KzgAmdError("bytes_to_kzg_commitment: C_KZG_RET %d" % C_KZG_BADARGS)
    return out


def bytes_from_bls_field(fr) -> bytes:
    out = C.create_string_buffer(32)
    lib().bytes_from_bls_field(out, C.byref(fr))
    return out.raw


def compute_cells_and_kzg_proofs(blob: bytes, settings: KZGSettings, want_cells=True, want_proofs=True):
    """kzg/src/eth/c_bindings.rs:356-372 -> (cells bytes 128*2048 | None, proofs bytes 128*48 | None)"""
    if len(blob) != BYTES_PER_BLOB:
        raise KzgAmdError("compute_cells_and_kzg_proofs: C_KZG_RET %d" % C_KZG_BADARGS)
    cells = C.create_string_buffer(128 * 2048) if want_cells else None
    proofs = C.create_string_buffer(128 * 48) if want_proofs else None
    rc = lib().compute_cells_and_kzg_proofs(cells, proofs, blob, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("compute_cells_and_kzg_proofs: C_KZG_RET %d" % rc)
    return (cells.raw if cells else None), (proofs.raw if proofs else None)


def recover_cells_and_kzg_proofs(cell_indices, cells: bytes, settings: KZGSettings, want_proofs=True):
    """kzg/src/eth/c_bindings.rs:202-289 -> (cells bytes 128*2048, proofs bytes 128*48 | None)"""
    n = len(cell_indices)
    idx = (C.c_uint64 * max(n, 1))(*cell_indices)
    out_cells = C.create_string_buffer(128 * 2048)
    out_proofs = C.create_string_buffer(128 * 48) if want_proofs else None
    f = lib().recover_cells_and_kzg_proofs
    f.restype = C.c_int
    rc = f(out_cells, out_proofs, idx, cells, C.c_uint64(n), C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("recover_cells_and_kzg_proofs: C_KZG_RET %d" % rc)
    return out_cells.raw, (out_proofs.raw if out_proofs else None)


def verify_cell_kzg_proof_batch(commitments: bytes, cell_indices, cells: bytes, proofs: bytes, settings: KZGSettings):
    """kzg/src/eth/c_bindings.rs:290-355 -> bool; len(cell_indices) tuples of 48 + 2048 + 48 bytes"""
    n = len(cell_indices)
    idx = (C.c_uint64 * max(n, 1))(*cell_indices)
    ok = C.c_bool(False)
    f = lib().verify_cell_kzg_proof_batch
    f.restype = C.c_int
    rc = f(C.byref(ok), commitments, idx, cells, proofs, C.c_uint64(n), C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("verify_cell_kzg_proof_batch: C_KZG_RET %d" % rc)
    return bool(ok.value)


def compute_verify_cell_kzg_proof_batch_challenge(commitments: bytes, commitment_indices, cell_indices, cells: bytes, proofs: bytes):
    """blst/src/eip_7594.rs:35-97 -> the challenge as 32 big-endian bytes (canonical)"""
    n = len(cell_indices)
    ci = (C.c_uint64 * max(n, 1))(*commitment_indices)
    idx = (C.c_uint64 * max(n, 1))(*cell_indices)
    out = BlstFr()
    f = lib().compute_verify_cell_kzg_proof_batch_challenge
    f.restype = C.c_int
    rc = f(C.byref(out), commitments, C.c_uint64(len(commitments) // 48), ci, idx, cells, proofs, C.c_uint64(n))
    if rc != C_KZG_OK:
        raise KzgAmdError("compute_verify_cell_kzg_proof_batch_challenge: C_KZG_RET %d" % rc)
    b = C.create_string_buffer(32)
    lib().bytes_from_bls_field(b, C.byref(out))
    return b.raw


def compute_cells_and_kzg_proofs_batch(blobs: bytes, n: int, settings: KZGSettings):
    cells = C.create_string_buffer(n * 128 * 2048)
    proofs = C.create_string_buffer(n * 128 * 48)
    rc = lib().kzgamd_compute_cells_and_kzg_proofs_batch(cells, proofs, blobs, n, C.byref(settings.c))
    if rc != C_KZG_OK:
        raise KzgAmdError("kzgamd_compute_cells_and_kzg_proofs_batch: C_KZG_RET %d" % rc)
    return cells.raw, proofs.raw


# ---------------------------------------------------------------- multi-GPU inside the library (csrc/multi.hip)
class MultiKZGSettings:
    """ndev CKZGSettings objects of one trusted setup, one per entry of `devices` (kzgamd_load_trusted_setup_file_multi):
    the in-process multi-GPU path — contiguous slabs of blobs per settings object, one host thread each, inside the
    library.  Two entries may name the same GPU (how the path is tested on a one-GPU box)."""

    def __init__(self, path, devices, config=None):
        self.ndev = len(devices)
        self.devices = list(devices)
        self.arr = (CKZGSettings * self.ndev)()
        self.loaded = False
        devs = (C.c_int * self.ndev)(*self.devices)
        f = _fopen(path)
        try:
            rc = lib().kzgamd_load_trusted_setup_file_multi_ex(self.arr, devs, self.ndev, f, _cfgp(config)) if config is not None \
                else lib().kzgamd_load_trusted_setup_file_multi(self.arr, devs, self.ndev, f)
        finally:
            _libc.fclose(f)
        if rc != C_KZG_OK:
            raise KzgAmdError("kzgamd_load_trusted_setup_file_multi: C_KZG_RET %d" % rc)
        self.loaded = True
        self.ptrs = (C.POINTER(CKZGSettings) * self.ndev)(*[C.pointer(self.arr[d]) for d in range(self.ndev)])

    def settings_devices(self):
        return [lib().kzgamd_settings_device(C.byref(self.arr[d])) for d in range(self.ndev)]

    def table_info(self, d, which=0):
        c, rows, wide = C.c_int(), C.c_int(), C.c_int()
        rc = lib().kzgamd_settings_table_info(C.byref(self.arr[d]), which, C.byref(c), C.byref(rows), C.byref(wide))
        return {"rc": rc, "window_bits": c.value, "rows": rows.value, "wide_table": wide.value}

    def commit_batch(self, blobs: bytes, n: int):
        out = C.create_string_buffer(48 * max(n, 1))
        rc = lib().kzgamd_blob_to_kzg_commitment_batch_multi(out, blobs, n, self.ptrs, self.ndev)
        if rc != C_KZG_OK:
            raise KzgAmdError("kzgamd_blob_to_kzg_commitment_batch_multi: C_KZG_RET %d" % rc)
        raw = out.raw
        return [raw[48 * i:48 * i + 48] for i in range(n)]

    def proof_batch(self, blobs: bytes, commitments: bytes, n: int):
        out = C.create_string_buffer(48 * max(n, 1))
        rc = lib().kzgamd_compute_blob_kzg_proof_batch_multi(out, blobs, commitments, n, self.ptrs, self.ndev)
        if rc != C_KZG_OK:
            raise KzgAmdError("kzgamd_compute_blob_kzg_proof_batch_multi: C_KZG_RET %d" % rc)
        raw = out.raw
        return [raw[48 * i:48 * i + 48] for i in range(n)]

    def cells_and_proofs_batch(self, blobs: bytes, n: int):
        cells = C.create_string_buffer(max(n, 1) * 128 * 2048)
        proofs = C.create_string_buffer(max(n, 1) * 128 * 48)
        rc = lib().kzgamd_compute_cells_and_kzg_proofs_batch_multi(cells, proofs, blobs, n, self.ptrs, self.ndev)
        if rc != C_KZG_OK:
            raise KzgAmdError("kzgamd_compute_cells_and_kzg_proofs_batch_multi: C_KZG_RET %d" % rc)
        return cells.raw[:n * 128 * 2048], proofs.raw[:n * 128 * 48]

    def verify_blob_batch(self, blobs: bytes, commitments: bytes, proofs: bytes, n: int) -> bool:
        ok = C.c_bool(False)
        return _verdict(lib().kzgamd_verify_blob_kzg_proof_batch_multi(C.byref(ok), blobs, commitments, proofs, n, self.ptrs,
                                                                       self.ndev), ok, "kzgamd_verify_blob_kzg_proof_batch_multi")

    def close(self):
        if self.loaded:
            lib().kzgamd_free_trusted_setup_multi(self.arr, self.ndev)
            self.loaded = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mult_pippenger_prepared_multi(handles, offsets, scalars):
    """kzgamd_mult_pippenger_prepared_multi: handles[d] prepared over points[offsets[d]:offsets[d+1]]; returns BlstP1."""
    nd = len(handles)
    hs = (C.c_void_p * nd)(*[h.handle for h in handles])
    offs = (C.c_size_t * (nd + 1))(*offsets)
    out = BlstP1()
    _check(lib().kzgamd_mult_pippenger_prepared_multi(hs, nd, C.byref(out), offs, _addr(scalars)),
           "kzgamd_mult_pippenger_prepared_multi")
    return out
