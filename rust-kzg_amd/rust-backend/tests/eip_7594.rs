//! kzg-bench/src/tests/eip_7594.rs for the MI355X backend (blst/tests/eip_7594.rs): cells, cell proofs (FK20 through
//! `fft_fr`, `g1_lincomb_batch` and `fft_g1` on the GPU), recovery and cell verification on the c-kzg vectors.
#[macro_use]
mod common;

use kzg::eip_4844::bytes_to_blob;
use kzg_bench::tests::eip_7594::*;
use rust_kzg_mi355x::backend::load_trusted_setup_filename_rust as load;
use rust_kzg_mi355x::MiBackend;

case!(vectors_cells, test_vectors_compute_cells::<MiBackend>(&load, &bytes_to_blob));
case!(vectors_cells_and_proofs, test_vectors_compute_cells_and_kzg_proofs::<MiBackend>(&load, &bytes_to_blob));
case!(vectors_recover, test_vectors_recover_cells_and_kzg_proofs::<MiBackend>(&load));
case!(vectors_verify_cells, test_vectors_verify_cell_kzg_proof_batch::<MiBackend>(&load));
case!(vectors_cell_challenge, test_vectors_compute_verify_cell_kzg_proof_batch_challenge::<MiBackend>());
