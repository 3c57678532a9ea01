//! kzg-bench/src/tests/{fft_fr, fft_g1, das}.rs for `MiFFTSettings` (blst/tests/fft_fr.rs, fft_g1.rs, das.rs): the
//! transforms run in `ntt.hip` / `fftg1.hip`; the "slow" reference each is compared with is the blst crate's O(n^2) one.
#[macro_use]
mod common;

use kzg::{FFTFr, FFTG1, G1};
use rust_kzg_mi355x::{FsFr, MiFFTSettings, MiG1};

fn points(n: usize) -> Vec<MiG1> {
    let g = MiG1::generator();
    let mut out = Vec::with_capacity(n);
    let mut acc = g;
    for _ in 0..n {
        out.push(acc);
        acc = acc.add_or_dbl(&g);
    }
    out
}

// the O(n^2) transforms of the blst crate on the wrapped settings, and this backend's GPU ones behind the same
// (output, input, stride, roots, roots_stride) shape the comparison tests call
fn fr_slow(out: &mut [FsFr], data: &[FsFr], stride: usize, roots: &[FsFr], roots_stride: usize) {
    rust_kzg_blst::fft_fr::fft_fr_slow(out, data, stride, roots, roots_stride)
}
fn fr_fast(out: &mut [FsFr], data: &[FsFr], stride: usize, roots: &[FsFr], roots_stride: usize) {
    rust_kzg_mi355x::fft_settings::fft_fr_strided(out, data, stride, roots, roots_stride)
}
fn g1_slow(out: &mut [MiG1], data: &[MiG1], stride: usize, roots: &[FsFr], roots_stride: usize) {
    rust_kzg_mi355x::fft_settings::fft_g1_slow(out, data, stride, roots, roots_stride)
}
fn g1_fast(out: &mut [MiG1], data: &[MiG1], stride: usize, roots: &[FsFr], roots_stride: usize) {
    rust_kzg_mi355x::fft_settings::fft_g1_strided(out, data, stride, roots, roots_stride)
}

mod fr {
    use super::*;
    use kzg_bench::tests::fft_fr::*;
    case!(against_the_slow_transform, compare_sft_fft::<FsFr, MiFFTSettings>(&fr_slow, &fr_fast));
    case!(roundtrip, roundtrip_fft::<FsFr, MiFFTSettings>());
    case!(inverse, inverse_fft::<FsFr, MiFFTSettings>());
    case!(stride, stride_fft::<FsFr, MiFFTSettings>());
}

mod g1 {
    use super::*;
    use kzg_bench::tests::fft_g1::*;
    case!(roundtrip, roundtrip_fft::<FsFr, MiG1, MiFFTSettings>(&points));
    case!(stride, stride_fft::<FsFr, MiG1, MiFFTSettings>(&points));
    case!(against_the_slow_transform, compare_ft_fft::<FsFr, MiG1, MiFFTSettings>(&g1_slow, &g1_fast, &points));
}

mod das {
    use super::*;
    use kzg_bench::tests::das::*;
    case!(known, das_extension_test_known::<FsFr, MiFFTSettings>());
    case!(random, das_extension_test_random::<FsFr, MiFFTSettings>());
}

#[test]
fn settings_expose_the_transforms() {
    // the trait objects the suite goes through
    fn takes<T: FFTFr<FsFr> + FFTG1<MiG1>>() {}
    takes::<MiFFTSettings>();
}
