//! kzg-bench/src/tests/{kzg_proofs, fk20_proofs}.rs for the MI355X backend (blst/tests/kzg_proofs.rs:60-95,
//! fk20_proofs.rs): commitments and single / multi proofs over a generated setup — `commit_to_poly` and
//! `compute_proof_single` are the lincombs that run on the GPU.  The FK20 settings types are the blst crate's generic ones
//! instantiated over this backend's KZG settings.
#[macro_use]
mod common;

use kzg_bench::tests::kzg_proofs::*;
use rust_kzg_mi355x::backend::generate_trusted_setup;
use rust_kzg_mi355x::MiBackend;

case!(setup_in_correct_form, trusted_setup_in_correct_form::<MiBackend>(&generate_trusted_setup));
case!(single_proof, proof_single::<MiBackend>(&generate_trusted_setup));
case!(nil_poly, commit_to_nil_poly::<MiBackend>(&generate_trusted_setup));
case!(too_long_poly, commit_to_too_long_poly_returns_err::<MiBackend>(&generate_trusted_setup));
case!(multi_proof, proof_multi::<MiBackend>(&generate_trusted_setup));
