//! FFI to libkzg_mi355x.so (include/kzg_mi355x.h) and the thin safe layer `rust-kzg-blst` would call.
//!
//! NOT COMPILED in the build image (no rustc/cargo there); every behaviour that matters is implemented
//! and tested on the C side of this boundary.  Mapping to the reference:
//!   * `GpuMsm` / `msm_prepared` / `msm`  <->  blst-sppark/src/lib.rs:8-62 (same three C symbols)
//!   * `GpuNtt::fft_fr` / `das_fft_extension` / `fft_g1`  <->  blst/src/fft_fr.rs:156-165,
//!     blst/src/data_availability_sampling.rs:78-100, blst/src/fft_g1.rs:54-83
//!   * `g1_sum`  —  combine step of one MSM split over several GPUs
use blst::{blst_fr, blst_p1, blst_p1_affine};
use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct RustError {
    pub code: c_int,
    pub message: *mut c_char,
}

/// `KzgAmdConfig` of include/kzg_mi355x.h: device, HBM budget per fixed-base table, tuning string.
#[repr(C)]
pub struct KzgAmdConfig {
    pub struct_size: u32,
    pub device: i32,
    pub table_budget_bytes: u64,
    pub tuning: *const c_char,
}

impl Default for KzgAmdConfig {
    fn default() -> Self {
        let mut cfg = core::mem::MaybeUninit::<KzgAmdConfig>::uninit();
        unsafe {
            kzgamd_config_init(cfg.as_mut_ptr());
            cfg.assume_init()
        }
    }
}

extern "C" {
    fn kzgamd_config_init(cfg: *mut KzgAmdConfig);
    fn kzgamd_prepare_msm_ex(points: *const blst_p1_affine, npoints: usize, cfg: *const KzgAmdConfig) -> *mut c_void;
    fn kzgamd_prepare_msm_matrix(points: *const blst_p1_affine, rows: usize, cols: usize, cfg: *const KzgAmdConfig) -> *mut c_void;
    fn kzgamd_msm_attach_matrix(msm: *mut c_void, points: *const blst_p1_affine, rows: usize, cols: usize,
                                cfg: *const KzgAmdConfig) -> RustError;
    fn kzgamd_mult_pippenger_matrix(msm: *mut c_void, out: *mut blst_p1, scalars: *const blst_fr, nmat: usize) -> RustError;
    fn kzgamd_msm_matrix_shape(msm: *mut c_void, rows: *mut usize, cols: *mut usize) -> c_int;
    fn kzgamd_ntt_new_ex(scale: u32, cfg: *const KzgAmdConfig) -> *mut c_void;
    fn prepare_msm(points: *const blst_p1_affine, npoints: usize) -> *mut c_void;
    fn free_msm(msm: *mut c_void);
    fn mult_pippenger_prepared(msm: *mut c_void, out: *mut blst_p1, npoints: usize, scalars: *const blst_fr) -> RustError;
    fn mult_pippenger_prepared_batch(msm: *mut c_void, out: *mut blst_p1, npoints: usize, nbatch: usize,
                                     scalars: *const blst_fr) -> RustError;
    fn mult_pippenger(out: *mut blst_p1, points: *const blst_p1_affine, npoints: usize, scalars: *const blst_fr) -> RustError;

    fn kzgamd_ntt_new(scale: u32) -> *mut c_void;
    fn kzgamd_ntt_free(ctx: *mut c_void);
    fn ntt_fr(ctx: *mut c_void, out: *mut blst_fr, input: *const blst_fr, n: usize, inverse: c_int) -> c_int;
    fn das_fft_extension(ctx: *mut c_void, odds: *mut blst_fr, evens: *const blst_fr, half_n: usize) -> c_int;
    fn fft_g1(ctx: *mut c_void, out: *mut blst_p1, input: *const blst_p1, n: usize, inverse: c_int) -> c_int;
    fn kzgamd_g1_sum(out: *mut blst_p1, input: *const blst_p1, n: usize);
}

fn check(err: RustError, what: &str) -> Result<(), String> {
    if err.code == 0 {
        return Ok(());
    }
    let msg = if err.message.is_null() {
        format!("{what}: error {}", err.code)
    } else {
        // the library malloc()s the message; take a copy and release it
        let s = unsafe { std::ffi::CStr::from_ptr(err.message) }.to_string_lossy().into_owned();
        unsafe { libc_free(err.message as *mut c_void) };
        format!("{what}: {s}")
    };
    Err(msg)
}

extern "C" {
    #[link_name = "free"]
    fn libc_free(p: *mut c_void);
}

/// Owning handle over the device-resident fixed-base table (what `SpparkPrecomputation.table` points at,
/// kzg/src/msm/sppark.rs:5-22).  Unlike the reference it is released on drop.
pub struct GpuMsm {
    handle: *mut c_void,
    npoints: usize,
}
unsafe impl Send for GpuMsm {}
unsafe impl Sync for GpuMsm {} // calls on one handle serialise inside the library

impl GpuMsm {
    pub fn new(points: &[blst_p1_affine]) -> Result<Self, String> {
        if points.is_empty() {
            return Err("empty point set".into());
        }
        let handle = unsafe { prepare_msm(points.as_ptr(), points.len()) };
        if handle.is_null() {
            return Err("prepare_msm failed (no gfx950 device?)".into());
        }
        Ok(Self { handle, npoints: points.len() })
    }

    /// Sum of scalars[i] * points[i] over the first scalars.len() points; scalars in Montgomery form.
    pub fn msm_prepared(&self, scalars: &[blst_fr]) -> Result<blst_p1, String> {
        if scalars.len() > self.npoints {
            return Err("more scalars than prepared points".into());
        }
        let mut out = blst_p1::default();
        check(unsafe { mult_pippenger_prepared(self.handle, &mut out, scalars.len(), scalars.as_ptr()) },
              "mult_pippenger_prepared")?;
        Ok(out)
    }

    /// `nbatch` MSMs at once; `scalars` is nbatch x npoints, row-major.
    pub fn msm_prepared_batch(&self, scalars: &[blst_fr], npoints: usize) -> Result<Vec<blst_p1>, String> {
        if npoints == 0 || scalars.len() % npoints != 0 || npoints > self.npoints {
            return Err("bad batch shape".into());
        }
        let nbatch = scalars.len() / npoints;
        let mut out = vec![blst_p1::default(); nbatch];
        check(unsafe { mult_pippenger_prepared_batch(self.handle, out.as_mut_ptr(), npoints, nbatch, scalars.as_ptr()) },
              "mult_pippenger_prepared_batch")?;
        Ok(out)
    }
}

/// Raw-handle forms for callers that keep the table behind `kzg::msm::precompute::PrecomputationTable`
/// (an opaque `*mut c_void`, kzg/src/msm/sppark.rs:5-22) instead of a `GpuMsm`.
///
/// # Safety
/// `handle` must come from [`prepare_raw`] (or `GpuMsm`) and not have been freed.
pub unsafe fn msm_prepared_raw(handle: *mut c_void, scalars: &[blst_fr]) -> Result<blst_p1, String> {
    let mut out = blst_p1::default();
    check(mult_pippenger_prepared(handle, &mut out, scalars.len(), scalars.as_ptr()), "mult_pippenger_prepared")?;
    Ok(out)
}

/// # Safety
/// as [`msm_prepared_raw`]; `scalars` is nbatch x npoints, row-major.
pub unsafe fn msm_prepared_batch_raw(handle: *mut c_void, scalars: &[blst_fr], npoints: usize) -> Result<Vec<blst_p1>, String> {
    if npoints == 0 || scalars.len() % npoints != 0 {
        return Err("bad batch shape".into());
    }
    let nbatch = scalars.len() / npoints;
    let mut out = vec![blst_p1::default(); nbatch];
    check(mult_pippenger_prepared_batch(handle, out.as_mut_ptr(), npoints, nbatch, scalars.as_ptr()),
          "mult_pippenger_prepared_batch")?;
    Ok(out)
}

/// The matrix of `precompute(points, matrix)` (kzg/src/msm/bgmw.rs:206-304) attached to a handle of [`prepare_raw`]:
/// `rows` base sets of `cols` points, row-major.  One pointer then serves `g1_lincomb` and `g1_lincomb_batch`.
///
/// # Safety
/// `handle` must come from [`prepare_raw`] and not have been freed.
pub unsafe fn attach_matrix_raw(handle: *mut c_void, points: &[blst_p1_affine], rows: usize, cols: usize,
                                cfg: Option<&KzgAmdConfig>) -> Result<(), String> {
    if rows == 0 || cols == 0 || points.len() != rows * cols {
        return Err("bad matrix shape".into());
    }
    let cfg_ptr = cfg.map_or(core::ptr::null(), |c| c as *const KzgAmdConfig);
    check(kzgamd_msm_attach_matrix(handle, points.as_ptr(), rows, cols, cfg_ptr), "kzgamd_msm_attach_matrix")
}

/// `multiply_batch` (kzg/src/msm/bgmw.rs:306-380) on the attached matrix: `scalars` is rows x cols, row-major;
/// one launch for all rows.
///
/// # Safety
/// `handle` must carry a matrix ([`attach_matrix_raw`]) of `rows` rows.
pub unsafe fn msm_matrix_raw(handle: *mut c_void, scalars: &[blst_fr], rows: usize) -> Result<Vec<blst_p1>, String> {
    let (mut hr, mut hc) = (0usize, 0usize);
    if kzgamd_msm_matrix_shape(handle, &mut hr, &mut hc) != 0 {
        return Err("no matrix attached to this handle".into());
    }
    if rows != hr || scalars.len() != hr * hc {
        return Err("bad matrix shape".into());
    }
    let mut out = vec![blst_p1::default(); rows];
    check(kzgamd_mult_pippenger_matrix(handle, out.as_mut_ptr(), scalars.as_ptr(), 1), "kzgamd_mult_pippenger_matrix")?;
    Ok(out)
}

/// A stand-alone matrix handle (`kzgamd_prepare_msm_matrix`); release with [`free_raw`].
pub fn prepare_matrix_raw(points: &[blst_p1_affine], rows: usize, cols: usize, cfg: Option<&KzgAmdConfig>) -> *mut c_void {
    if rows == 0 || cols == 0 || points.len() != rows * cols {
        return core::ptr::null_mut();
    }
    let cfg_ptr = cfg.map_or(core::ptr::null(), |c| c as *const KzgAmdConfig);
    unsafe { kzgamd_prepare_msm_matrix(points.as_ptr(), rows, cols, cfg_ptr) }
}

/// [`prepare_raw`] with a configuration (device, table budget, tuning).
pub fn prepare_raw_with(points: &[blst_p1_affine], cfg: &KzgAmdConfig) -> *mut c_void {
    unsafe { kzgamd_prepare_msm_ex(points.as_ptr(), points.len(), cfg) }
}

/// `prepare_multi_scalar_mult` of blst-sppark/src/lib.rs:8-17: the handle is leaked into the settings object like
/// the reference's (release it with [`free_raw`] when the settings go away).
pub fn prepare_raw(points: &[blst_p1_affine]) -> *mut c_void {
    unsafe { prepare_msm(points.as_ptr(), points.len()) }
}

/// # Safety
/// `handle` must come from [`prepare_raw`]; it is dead afterwards.
pub unsafe fn free_raw(handle: *mut c_void) {
    free_msm(handle)
}

impl Drop for GpuMsm {
    fn drop(&mut self) {
        unsafe { free_msm(self.handle) }
    }
}

/// Variable-base MSM (the `None` precomputation arm of blst/src/kzg_proofs.rs:53-57).
pub fn msm(points: &[blst_p1_affine], scalars: &[blst_fr]) -> Result<blst_p1, String> {
    if points.len() != scalars.len() {
        return Err("length mismatch".into());
    }
    let mut out = blst_p1::default();
    if points.is_empty() {
        return Ok(out);
    }
    check(unsafe { mult_pippenger(&mut out, points.as_ptr(), points.len(), scalars.as_ptr()) }, "mult_pippenger")?;
    Ok(out)
}

/// Device NTT context; one per `FsFFTSettings` (created in `FFTSettings::new(scale)`).
pub struct GpuNtt {
    ctx: *mut c_void,
}
unsafe impl Send for GpuNtt {}
unsafe impl Sync for GpuNtt {}

impl GpuNtt {
    pub fn new(scale: usize) -> Result<Self, String> {
        if scale >= 32 {
            return Err(String::from("Scale is expected to be within root of unity matrix row size"));
        }
        let ctx = unsafe { kzgamd_ntt_new(scale as u32) };
        if ctx.is_null() {
            return Err("kzgamd_ntt_new failed (no gfx950 device?)".into());
        }
        Ok(Self { ctx })
    }

    /// the same on a chosen device / with tuning keys (`KzgAmdConfig`)
    pub fn with_config(scale: usize, cfg: &KzgAmdConfig) -> Result<Self, String> {
        if scale >= 32 {
            return Err(String::from("Scale is expected to be within root of unity matrix row size"));
        }
        let ctx = unsafe { kzgamd_ntt_new_ex(scale as u32, cfg) };
        if ctx.is_null() {
            return Err("kzgamd_ntt_new_ex failed (no gfx950 device, or a malformed configuration)".into());
        }
        Ok(Self { ctx })
    }

    /// `FFTFr::fft_fr`: natural order in and out, inverse scaled by 1/n; error strings as in the reference.
    pub fn fft_fr(&self, data: &[blst_fr], inverse: bool) -> Result<Vec<blst_fr>, String> {
        let mut out = vec![blst_fr::default(); data.len()];
        match unsafe { ntt_fr(self.ctx, out.as_mut_ptr(), data.as_ptr(), data.len(), inverse as c_int) } {
            0 => Ok(out),
            1 => Err(String::from("Supplied list is longer than the available max width")),
            2 => Err(String::from("A list with power-of-two length expected")),
            e => Err(format!("GPU NTT failed: {e}")),
        }
    }

    /// `FFTG1::fft_g1` (blst/src/fft_g1.rs:54-83); results equal the reference's as group elements.
    pub fn fft_g1(&self, data: &[blst_p1], inverse: bool) -> Result<Vec<blst_p1>, String> {
        let mut out = vec![blst_p1::default(); data.len()];
        match unsafe { fft_g1(self.ctx, out.as_mut_ptr(), data.as_ptr(), data.len(), inverse as c_int) } {
            0 => Ok(out),
            1 => Err(String::from("Supplied list is longer than the available max width")),
            2 => Err(String::from("A list with power-of-two length expected")),
            e => Err(format!("GPU fft_g1 failed: {e}")),
        }
    }

    /// `DASExtension::das_fft_extension`.
    pub fn das_fft_extension(&self, evens: &[blst_fr]) -> Result<Vec<blst_fr>, String> {
        let mut out = vec![blst_fr::default(); evens.len()];
        match unsafe { das_fft_extension(self.ctx, out.as_mut_ptr(), evens.as_ptr(), evens.len()) } {
            0 => Ok(out),
            1 => Err(String::from("A non-zero list ab expected")),
            2 => Err(String::from("A list with power-of-two length expected")),
            3 => Err(String::from("Supplied list is longer than the available max width")),
            e => Err(format!("GPU DAS extension failed: {e}")),
        }
    }
}

impl Drop for GpuNtt {
    fn drop(&mut self) {
        unsafe { kzgamd_ntt_free(self.ctx) }
    }
}

/// Sum of Jacobian points on the host: each rank of a multi-GPU MSM contributes one partial.
pub fn g1_sum(partials: &[blst_p1]) -> blst_p1 {
    let mut out = blst_p1::default();
    unsafe { kzgamd_g1_sum(&mut out, partials.as_ptr(), partials.len()) };
    out
}

/// The c-kzg-4844 surface of libkzg_mi355x.so under its exact names (include/kzg_mi355x.h, B3): the same
/// signatures the blst crate exports with `#[no_mangle]` (blst/src/eip_4844.rs:160-530, blst/src/eip_7594.rs:35-44,
/// kzg/src/eth/c_bindings.rs:202-372), so that the binding test-suite (kzg-bench/src/tests/c_bindings.rs) and any
/// consumer of the reference's C API can be pointed at the GPU library.  A process must not also link the blst crate
/// with its `c_bindings` feature (duplicate symbols): use libkzg_mi355x_prefixed.so for that.
pub mod ckzg {
    use kzg::eth::c_bindings::{Blob, Bytes32, Bytes48, CKZGSettings, CKzgRet, Cell, KZGCommitment, KZGProof};
    use libc::FILE;

    extern "C" {
        pub fn load_trusted_setup(out: *mut CKZGSettings, g1_monomial_bytes: *const u8, num_g1_monomial_bytes: u64,
                                  g1_lagrange_bytes: *const u8, num_g1_lagrange_bytes: u64, g2_monomial_bytes: *const u8,
                                  num_g2_monomial_bytes: u64, precompute: u64) -> CKzgRet;
        pub fn load_trusted_setup_file(out: *mut CKZGSettings, in_: *mut FILE) -> CKzgRet;
        pub fn free_trusted_setup(s: *mut CKZGSettings);
        pub fn blob_to_kzg_commitment(out: *mut KZGCommitment, blob: *const Blob, s: &CKZGSettings) -> CKzgRet;
        pub fn compute_kzg_proof(proof_out: *mut KZGProof, y_out: *mut Bytes32, blob: *const Blob, z_bytes: *const Bytes32,
                                 s: &CKZGSettings) -> CKzgRet;
        pub fn compute_blob_kzg_proof(out: *mut KZGProof, blob: *const Blob, commitment_bytes: *const Bytes48,
                                      s: &CKZGSettings) -> CKzgRet;
        pub fn verify_kzg_proof(ok: *mut bool, commitment_bytes: *const Bytes48, z_bytes: *const Bytes32, y_bytes: *const Bytes32,
                                proof_bytes: *const Bytes48, s: &CKZGSettings) -> CKzgRet;
        pub fn verify_blob_kzg_proof(ok: *mut bool, blob: *const Blob, commitment_bytes: *const Bytes48,
                                     proof_bytes: *const Bytes48, s: &CKZGSettings) -> CKzgRet;
        pub fn verify_blob_kzg_proof_batch(ok: *mut bool, blobs: *const Blob, commitments_bytes: *const Bytes48,
                                           proofs_bytes: *const Bytes48, n: usize, s: &CKZGSettings) -> CKzgRet;
        pub fn compute_cells_and_kzg_proofs(cells: *mut Cell, proofs: *mut KZGProof, blob: *const Blob, s: &CKZGSettings) -> CKzgRet;
        pub fn recover_cells_and_kzg_proofs(recovered_cells: *mut Cell, recovered_proofs: *mut KZGProof, cell_indices: *const u64,
                                            cells: *const Cell, num_cells: u64, s: &CKZGSettings) -> CKzgRet;
        pub fn verify_cell_kzg_proof_batch(ok: *mut bool, commitments_bytes: *const Bytes48, cell_indices: *const u64,
                                           cells: *const Cell, proofs_bytes: *const Bytes48, num_cells: u64, s: &CKZGSettings) -> CKzgRet;
        // batched and multi-GPU forms (new API, include/kzg_mi355x.h): contiguous slabs of the batch per settings object
        pub fn kzgamd_blob_to_kzg_commitment_batch(out: *mut KZGCommitment, blobs: *const Blob, n: usize, s: &CKZGSettings) -> CKzgRet;
        pub fn kzgamd_compute_blob_kzg_proof_batch(out: *mut KZGProof, blobs: *const Blob, commitments: *const Bytes48, n: usize,
                                                   s: &CKZGSettings) -> CKzgRet;
        pub fn kzgamd_load_trusted_setup_file_multi(out: *mut CKZGSettings, devices: *const core::ffi::c_int, ndev: usize,
                                                    in_: *mut FILE) -> CKzgRet;
        pub fn kzgamd_free_trusted_setup_multi(s: *mut CKZGSettings, ndev: usize);
        pub fn kzgamd_blob_to_kzg_commitment_batch_multi(out: *mut KZGCommitment, blobs: *const Blob, n: usize,
                                                         s: *const *const CKZGSettings, ndev: usize) -> CKzgRet;
        pub fn kzgamd_compute_blob_kzg_proof_batch_multi(out: *mut KZGProof, blobs: *const Blob, commitments: *const Bytes48,
                                                         n: usize, s: *const *const CKZGSettings, ndev: usize) -> CKzgRet;
        pub fn kzgamd_verify_blob_kzg_proof_batch_multi(ok: *mut bool, blobs: *const Blob, commitments: *const Bytes48,
                                                        proofs: *const Bytes48, n: usize, s: *const *const CKZGSettings,
                                                        ndev: usize) -> CKzgRet;
        pub fn kzgamd_pin_host_buffer(p: *mut core::ffi::c_void, bytes: usize) -> core::ffi::c_int;
        pub fn kzgamd_unpin_host_buffer(p: *mut core::ffi::c_void) -> core::ffi::c_int;
        pub fn kzgamd_device_count() -> core::ffi::c_int;
        /// the slab [lo, hi) the `_multi` entry points give settings object `k` of `parts` for a batch of `n`
        pub fn kzgamd_shard_range(n: usize, parts: usize, k: usize, lo: *mut usize, hi: *mut usize) -> core::ffi::c_int;
    }
}
