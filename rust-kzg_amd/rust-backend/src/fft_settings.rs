//! FFT settings for the MI355X backend: the blst backend's roots, the transforms on the GPU.
//! Reference shape: blst/src/types/fft_settings.rs (type), blst/src/fft_fr.rs:112-165 (FFTFr),
//! blst/src/fft_g1.rs:54-83 (FFTG1), blst/src/data_availability_sampling.rs:78-100 (DASExtension).
extern crate alloc;

use alloc::string::String;
use alloc::sync::Arc;
use alloc::vec::Vec;

use blst::{blst_fr, blst_p1};
use kzg::{DASExtension, FFTFr, FFTSettings, FFTG1};
use rust_kzg_blst::types::fft_settings::FsFFTSettings;
use rust_kzg_blst::types::fr::FsFr;
use rust_kzg_mi355x_sys::GpuNtt;

use crate::g1::MiG1;

/// `FsFFTSettings` plus the device context (`kzgamd_ntt_new(scale)`: twiddle tables in HBM).  Cloning shares the
/// context; the library serialises calls on one context.  `gpu` is `None` only for `Default::default()`: like the
/// reference's `FsFFTSettings::default()` that is a pure host value (generic code in `kzg::` builds placeholder
/// settings with it, also on machines without a GPU); a transform on such a value is an error, not a fallback.
#[derive(Clone)]
pub struct MiFFTSettings {
    pub inner: FsFFTSettings,
    pub gpu: Option<Arc<GpuNtt>>,
}

impl MiFFTSettings {
    fn gpu(&self) -> Result<&GpuNtt, String> {
        self.gpu.as_deref().ok_or_else(|| String::from("MiFFTSettings::default() has no device context; use new(scale)"))
    }
}

impl core::fmt::Debug for MiFFTSettings {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        f.debug_struct("MiFFTSettings").field("max_width", &self.inner.max_width).finish()
    }
}

impl Default for MiFFTSettings {
    fn default() -> Self {
        Self { inner: FsFFTSettings::default(), gpu: None }
    }
}

impl FFTSettings<FsFr> for MiFFTSettings {
    /// Same roots as `FsFFTSettings::new` (blst/src/types/fft_settings.rs:30-58); the device side expands the same
    /// SCALE2_ROOT_OF_UNITY row, so host getters and device transforms agree bit for bit.
    fn new(scale: usize) -> Result<Self, String> {
        let inner = FsFFTSettings::new(scale)?;
        let gpu = Some(Arc::new(GpuNtt::new(scale)?));
        Ok(Self { inner, gpu })
    }
    fn get_max_width(&self) -> usize {
        self.inner.get_max_width()
    }
    fn get_reverse_roots_of_unity_at(&self, i: usize) -> FsFr {
        self.inner.get_reverse_roots_of_unity_at(i)
    }
    fn get_reversed_roots_of_unity(&self) -> &[FsFr] {
        self.inner.get_reversed_roots_of_unity()
    }
    fn get_roots_of_unity_at(&self, i: usize) -> FsFr {
        self.inner.get_roots_of_unity_at(i)
    }
    fn get_roots_of_unity(&self) -> &[FsFr] {
        self.inner.get_roots_of_unity()
    }
    fn get_brp_roots_of_unity(&self) -> &[FsFr] {
        self.inner.get_brp_roots_of_unity()
    }
    fn get_brp_roots_of_unity_at(&self, i: usize) -> FsFr {
        self.inner.get_brp_roots_of_unity_at(i)
    }
}

#[inline]
fn fr_raw(data: &[FsFr]) -> &[blst_fr] {
    // FsFr is repr(transparent)-like over blst_fr (blst/src/types/fr.rs)
    unsafe { core::slice::from_raw_parts(data.as_ptr() as *const blst_fr, data.len()) }
}

impl FFTFr<FsFr> for MiFFTSettings {
    /// blst/src/fft_fr.rs:112-165; the length checks and their messages come back from the library
    /// (ntt_fr return codes 1 / 2), the butterflies run as radix-4 register rounds over LDS tiles on the GPU.
    fn fft_fr(&self, data: &[FsFr], inverse: bool) -> Result<Vec<FsFr>, String> {
        let out = self.gpu()?.fft_fr(fr_raw(data), inverse)?;
        Ok(out.into_iter().map(FsFr).collect())
    }
}

impl DASExtension<FsFr> for MiFFTSettings {
    /// blst/src/data_availability_sampling.rs:78-100
    fn das_fft_extension(&self, evens: &[FsFr]) -> Result<Vec<FsFr>, String> {
        let out = self.gpu()?.das_fft_extension(fr_raw(evens))?;
        Ok(out.into_iter().map(FsFr).collect())
    }
}

impl FFTG1<MiG1> for MiFFTSettings {
    /// blst/src/fft_g1.rs:54-83.  Results equal the reference's as group elements (the Jacobian representative
    /// differs); callers that serialise or compare with `equals` see no difference.
    fn fft_g1(&self, data: &[MiG1], inverse: bool) -> Result<Vec<MiG1>, String> {
        let raw = unsafe { core::slice::from_raw_parts(data.as_ptr() as *const blst_p1, data.len()) };
        let out = self.gpu()?.fft_g1(raw, inverse)?;
        Ok(out.into_iter().map(MiG1::from_blst).collect())
    }
}

// ---- the (output, input, stride, roots, roots_stride) shape of blst/src/fft_fr.rs:14-30, fft_g1.rs:13-30 ----------
// kzg-bench's comparison tests (compare_sft_fft, compare_ft_fft) take the "fast" and the "slow" transform in this
// shape.  The strided input is gathered, the transform of `out.len()` points runs on the GPU (a context of exactly that
// scale: the roots the caller passes are the settings' roots at `roots_stride`, i.e. the standard n-th roots).

fn log2_exact(n: usize) -> u32 {
    assert!(n.is_power_of_two(), "A list with power-of-two length expected");
    n.trailing_zeros()
}

pub fn fft_fr_strided(out: &mut [FsFr], data: &[FsFr], stride: usize, _roots: &[FsFr], _roots_stride: usize) {
    let n = out.len();
    let gathered: Vec<blst_fr> = (0..n).map(|i| data[i * stride].0).collect();
    let ctx = GpuNtt::new(log2_exact(n) as usize).expect("kzgamd_ntt_new");
    let res = ctx.fft_fr(&gathered, false).expect("ntt_fr");
    for (o, r) in out.iter_mut().zip(res) {
        *o = FsFr(r);
    }
}

pub fn fft_g1_strided(out: &mut [MiG1], data: &[MiG1], stride: usize, _roots: &[FsFr], _roots_stride: usize) {
    let n = out.len();
    let gathered: Vec<blst_p1> = (0..n).map(|i| data[i * stride].0 .0).collect();
    let ctx = GpuNtt::new(log2_exact(n) as usize).expect("kzgamd_ntt_new");
    let res = ctx.fft_g1(&gathered, false).expect("fft_g1");
    for (o, r) in out.iter_mut().zip(res) {
        *o = MiG1::from_blst(r);
    }
}

/// the O(n^2) definition, on the host (blst/src/fft_g1.rs:86-110 for `MiG1`): the yardstick of compare_ft_fft
pub fn fft_g1_slow(out: &mut [MiG1], data: &[MiG1], stride: usize, roots: &[FsFr], roots_stride: usize) {
    use kzg::{G1Mul, G1};
    let n = out.len();
    for i in 0..n {
        let mut acc = data[0].mul(&roots[0]);
        for j in 1..n {
            let term = data[j * stride].mul(&roots[((i * j) % n) * roots_stride]);
            acc = acc.add_or_dbl(&term);
        }
        out[i] = acc;
    }
}
