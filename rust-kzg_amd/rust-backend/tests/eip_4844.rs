//! kzg-bench/src/tests/eip_4844.rs for the MI355X backend (the blst instantiation: blst/tests/eip_4844.rs:38-448).
//! Every blob_to_kzg_commitment / compute_*_proof below reaches `MiG1::g1_lincomb` -> `mult_pippenger_prepared` on the GPU.
#[macro_use]
mod common;

use kzg::eip_4844::{
    blob_to_kzg_commitment_rust as commit, blob_to_polynomial as to_poly, bytes_to_blob as to_blob,
    compute_blob_kzg_proof_rust as blob_proof, compute_challenge_rust as challenge, compute_kzg_proof_rust as proof,
    compute_powers, evaluate_polynomial_in_evaluation_form as eval, verify_blob_kzg_proof_batch_rust as verify_batch,
    verify_blob_kzg_proof_rust as verify_blob, verify_kzg_proof_rust as verify,
};
use kzg_bench::tests::eip_4844::*;
use rust_kzg_mi355x::backend::load_trusted_setup_filename_rust as load;
use rust_kzg_mi355x::{FsFr, FsPoly, MiG1};

case!(bytes_to_bls_field, bytes_to_bls_field_test::<FsFr>());
case!(powers, compute_powers_test::<FsFr>(&compute_powers));
case!(commitment, nine!(blob_to_kzg_commitment_test)(&load, &commit));
case!(kzg_proof, nine!(compute_kzg_proof_test)(&load, &proof, &to_poly, &eval));
case!(kzg_proof_round_trip, nine!(compute_and_verify_kzg_proof_round_trip_test)(&load, &commit, &to_blob, &proof, &to_poly, &eval, &verify));
case!(kzg_proof_within_domain, nine!(compute_and_verify_kzg_proof_within_domain_test)(&load, &commit, &to_blob, &proof, &to_poly, &eval, &verify));
case!(kzg_proof_incorrect, nine!(compute_and_verify_kzg_proof_fails_with_incorrect_proof_test)(&load, &commit, &to_blob, &proof, &to_poly, &eval, &verify));
case!(blob_proof_round_trip, nine!(compute_and_verify_blob_kzg_proof_test)(&load, &commit, &to_blob, &blob_proof, &verify_blob));
case!(blob_proof_incorrect, nine!(compute_and_verify_blob_kzg_proof_fails_with_incorrect_proof_test)(&load, &commit, &to_blob, &blob_proof, &verify_blob));
case!(batch, nine!(verify_kzg_proof_batch_test)(&load, &commit, &to_blob, &blob_proof, &verify_batch));
case!(batch_incorrect, nine!(verify_kzg_proof_batch_fails_with_incorrect_proof_test)(&load, &commit, &to_blob, &blob_proof, &verify_batch));
// the c-kzg-4844 mainnet vectors (the same files tests/golden/kzg_mainnet.json is generated from)
case!(vectors_commitment, nine!(test_vectors_blob_to_kzg_commitment)(&load, &commit, &to_blob));
case!(vectors_kzg_proof, nine!(test_vectors_compute_kzg_proof)(&load, &proof, &to_blob));
case!(vectors_blob_proof, nine!(test_vectors_compute_blob_kzg_proof)(&load, &to_blob, &blob_proof));
case!(vectors_verify, nine!(test_vectors_verify_kzg_proof)(&load, &verify));
case!(vectors_verify_blob, nine!(test_vectors_verify_blob_kzg_proof)(&load, &to_blob, &verify_blob));
case!(vectors_verify_batch, nine!(test_vectors_verify_blob_kzg_proof_batch)(&load, &to_blob, &verify_batch));
case!(vectors_challenge, test_vectors_compute_challenge::<FsFr, MiG1>(&to_blob, &challenge));
// argument validation
case!(incorrect_blob_length, compute_kzg_proof_incorrect_blob_length_test::<FsFr, FsPoly>(&to_poly));
case!(incorrect_poly_length, nine_poly_first!(compute_kzg_proof_incorrect_poly_length_test)(&eval));
case!(empty_blob_vector, nine_poly_first!(compute_kzg_proof_empty_blob_vector_test)(&verify_batch));
case!(incorrect_commitments_len, nine_poly_first!(compute_kzg_proof_incorrect_commitments_len_test)(&verify_batch));
case!(incorrect_proofs_len, nine_poly_first!(compute_kzg_proof_incorrect_proofs_len_test)(&verify_batch));
case!(batched_input, nine_poly_first!(validate_batched_input_test)(&verify_batch, &load));
