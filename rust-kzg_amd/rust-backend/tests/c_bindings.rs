//! kzg-bench/src/tests/c_bindings.rs against the C-ABI of libkzg_mi355x.so itself (blst/tests/c_bindings.rs:1-80 runs
//! them against the blst crate's `#[no_mangle]` exports): the exact c-kzg-4844 names and signatures, bound in the sys
//! crate (`rust_kzg_mi355x_sys::ckzg`).  Also what tests/c_abi_harness.c replays from C.
#[macro_use]
mod common;

use kzg_bench::tests::c_bindings::*;
use rust_kzg_mi355x_sys::ckzg::{
    blob_to_kzg_commitment, compute_blob_kzg_proof, free_trusted_setup, load_trusted_setup, load_trusted_setup_file,
};

case!(commitment_invalid_blob, blob_to_kzg_commitment_invalid_blob_test(blob_to_kzg_commitment, load_trusted_setup_file));
case!(setup_invalid_g1_length, load_trusted_setup_invalid_g1_byte_length_test(load_trusted_setup));
case!(setup_invalid_g1_point, load_trusted_setup_invalid_g1_point_test(load_trusted_setup));
case!(setup_invalid_g2_length, load_trusted_setup_invalid_g2_byte_length_test(load_trusted_setup));
case!(setup_invalid_g2_point, load_trusted_setup_invalid_g2_point_test(load_trusted_setup));
case!(setup_invalid_form, load_trusted_setup_invalid_form_test(load_trusted_setup));
case!(setup_file_invalid_format, load_trusted_setup_file_invalid_format_test(load_trusted_setup_file));
case!(setup_file_valid_format, load_trusted_setup_file_valid_format_test(load_trusted_setup_file));
case!(free_null, free_trusted_setup_null_ptr_test(free_trusted_setup));
case!(free_clears, free_trusted_setup_set_all_values_to_null_test(free_trusted_setup, load_trusted_setup_file));
case!(proof_invalid_blob, compute_blob_kzg_proof_invalid_blob_test(compute_blob_kzg_proof, load_trusted_setup_file));
case!(proof_commitment_at_infinity, compute_blob_kzg_proof_commitment_is_point_at_infinity_test(compute_blob_kzg_proof, load_trusted_setup_file));
