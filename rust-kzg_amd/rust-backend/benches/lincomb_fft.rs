//! kzg-bench/src/benches/{lincomb, fft, das, eip_7594}.rs for the MI355X backend (blst/benches/lincomb.rs, fft.rs,
//! das.rs, eip_7594.rs): g1_lincomb at the sizes of the reference's sweep, fft_fr / fft_g1, the DAS extension and
//! the cell functions.
use criterion::{criterion_group, criterion_main, Criterion};
use kzg::eip_4844::{blob_to_kzg_commitment_rust, bytes_to_blob};
use kzg_bench::benches::das::bench_das_extension;
use kzg_bench::benches::eip_7594::bench_eip_7594;
use kzg_bench::benches::fft::{bench_fft_fr, bench_fft_g1};
use kzg_bench::benches::lincomb::bench_g1_lincomb;
use rust_kzg_mi355x::backend::load_trusted_setup_filename_rust;
use rust_kzg_mi355x::g1::{g1_linear_combination, MiG1Affine, MiG1ProjAddAffine};
use rust_kzg_mi355x::{FsFp, FsFr, MiBackend, MiFFTSettings, MiG1};

fn lincomb(c: &mut Criterion) {
    bench_g1_lincomb::<FsFr, MiG1, FsFp, MiG1Affine, MiG1ProjAddAffine>(c, &g1_linear_combination);
}
fn fft_fr(c: &mut Criterion) {
    bench_fft_fr::<FsFr, MiFFTSettings>(c);
}
fn fft_g1(c: &mut Criterion) {
    bench_fft_g1::<FsFr, MiG1, MiFFTSettings>(c);
}
fn das(c: &mut Criterion) {
    bench_das_extension::<FsFr, MiFFTSettings>(c);
}
fn cells(c: &mut Criterion) {
    bench_eip_7594::<MiBackend>(c, &load_trusted_setup_filename_rust, &bytes_to_blob, &blob_to_kzg_commitment_rust);
}

criterion_group! {
    name = benches;
    config = Criterion::default().sample_size(10);
    targets = lincomb, fft_fr, fft_g1, das, cells
}
criterion_main!(benches);
