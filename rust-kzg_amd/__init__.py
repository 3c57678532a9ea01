"""rust-kzg_amd — MI355X-native KZG hot path (MSM + NTT) behind rust-kzg's own seams.

This package is host-side plumbing only: it loads the C-ABI library
(csrc/libkzg_mi355x.so, built by build.py with hipcc for gfx950) and mirrors the
reference's plug-in interface for the path:

  prepare_multi_scalar_mult / multi_scalar_mult_prepared / multi_scalar_mult
      = blst-sppark/src/lib.rs:8-62 (the three FFI wrappers `rust-kzg-blst` calls)
  FFTSettings.fft_fr / das_fft_extension
      = kzg::FFTFr / kzg::DASExtension for FsFFTSettings (blst/src/fft_fr.rs:156-165,
        blst/src/data_availability_sampling.rs:78-100)

There is no CPU fallback: if the library is missing or no GPU is visible the calls raise.
The directory name contains a hyphen; import it with `load()` from __graft_entry__ /
tests (importlib under the module name `rust_kzg_amd`).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libkzg_mi355x.so")


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


_lib = None

# every symbol include/kzg_mi355x.h declares; tests check the library exports all of them
EXPORTS = [
    "prepare_msm", "mult_pippenger_prepared", "mult_pippenger", "free_msm", "mult_pippenger_prepared_batch",
    "kzgamd_msm_prepared_batch_device", "kzgamd_msm_info", "kzgamd_device_count", "kzgamd_version",
]


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
    L.kzgamd_device_count.restype = C.c_int
    L.kzgamd_version.restype = C.c_char_p
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

    def __init__(self, points, npoints):
        self.npoints = npoints
        self.handle = lib().prepare_msm(_addr(points), npoints)
        if not self.handle:
            raise KzgAmdError("prepare_msm failed (no GPU, or bad arguments)")

    def info(self):
        c, rows, nb, n = C.c_int(), C.c_int(), C.c_size_t(), C.c_size_t()
        lib().kzgamd_msm_info(self.handle, C.byref(c), C.byref(rows), C.byref(nb), C.byref(n))
        return {"window_bits": c.value, "rows": rows.value, "nbuckets": nb.value, "npoints": n.value}

    def close(self):
        if self.handle:
            lib().free_msm(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prepare_multi_scalar_mult(points, npoints):
    """blst-sppark/src/lib.rs:8-17.  points: ctypes array / buffer of blst_p1_affine."""
    return PreparedMsm(points, npoints)


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
