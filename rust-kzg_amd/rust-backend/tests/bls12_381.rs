//! kzg-bench/src/tests/bls12_381.rs: the G1 half for `MiG1` (blst/tests/bls12_381.rs:1-141).  The Fr / G2 cases of
//! that file test `FsFr` / `FsG2`, which this backend re-exports unchanged, so they stay with the blst crate.
#[macro_use]
mod common;

use kzg_bench::tests::bls12_381::*;
use rust_kzg_mi355x::g1::g1_linear_combination;
use rust_kzg_mi355x::{FsFr, FsG2, MiG1};

case!(p1_mul, p1_mul_works::<FsFr, MiG1>());
case!(p1_sub, p1_sub_works::<MiG1>());
case!(identity_is_infinity, g1_identity_is_infinity::<MiG1>());
case!(identity_is_identity, g1_identity_is_identity::<MiG1>());
// g1_linear_combination: len < 8 on the host, otherwise mult_pippenger on the GPU (blst/src/kzg_proofs.rs:25-72)
case!(lincomb_made, five!(g1_make_linear_combination)(&g1_linear_combination));
case!(lincomb_random, five!(g1_random_linear_combination)(&g1_linear_combination));
case!(lincomb_infinity_points, five!(g1_linear_combination_infinity_points)(&g1_linear_combination));
case!(lincomb_small, five!(g1_small_linear_combination)(&g1_linear_combination));
case!(pairings, pairings_work::<FsFr, MiG1, FsG2>(&|a1: &MiG1, a2: &FsG2, b1: &MiG1, b2: &FsG2| {
    rust_kzg_blst::kzg_proofs::pairings_verify(&a1.0, a2, &b1.0, b2)
}));
