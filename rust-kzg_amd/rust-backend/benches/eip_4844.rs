//! kzg-bench/src/benches/eip_4844.rs for the MI355X backend (blst/benches/eip_4844.rs): the reference's own criterion
//! benchmark of blob_to_kzg_commitment / compute_*_proof / verify_*, the number bench/run_reference_suite.sh puts next
//! to `cargo bench -p rust-kzg-blst --bench eip_4844`.
use criterion::{criterion_group, criterion_main, Criterion};
use kzg::eip_4844::{
    blob_to_kzg_commitment_rust, bytes_to_blob, compute_blob_kzg_proof_rust, compute_kzg_proof_rust,
    verify_blob_kzg_proof_batch_rust, verify_blob_kzg_proof_rust, verify_kzg_proof_rust,
};
use kzg_bench::benches::eip_4844::bench_eip_4844;
use rust_kzg_mi355x::backend::load_trusted_setup_filename_rust;
use rust_kzg_mi355x::g1::{MiG1Affine, MiG1ProjAddAffine};
use rust_kzg_mi355x::{FsFp, FsFr, FsG2, FsPoly, MiFFTSettings, MiG1, MiKZGSettings};

fn run(c: &mut Criterion) {
    bench_eip_4844::<FsFr, MiG1, FsG2, FsPoly, MiFFTSettings, MiKZGSettings, FsFp, MiG1Affine, MiG1ProjAddAffine>(
        c,
        &load_trusted_setup_filename_rust,
        &blob_to_kzg_commitment_rust,
        &bytes_to_blob,
        &compute_kzg_proof_rust,
        &verify_kzg_proof_rust,
        &compute_blob_kzg_proof_rust,
        &verify_blob_kzg_proof_rust,
        &verify_blob_kzg_proof_batch_rust,
    );
}

criterion_group!(benches, run);
criterion_main!(benches);
