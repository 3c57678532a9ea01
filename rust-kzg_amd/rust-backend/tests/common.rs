//! Shorthands for the suite files: the reference's generic tests (kzg-bench/src/tests/*) are functions over a list of
//! backend types and the entry points under test; every file of this directory instantiates them for the MI355X backend,
//! as blst/tests/*.rs does for the blst one.  `NINE!` is the type list most EIP-4844 tests take
//! (Fr, G1, G2, Poly, FFTSettings, KZGSettings, Fp, G1Affine, G1ProjAddAffine); `case!` declares one #[test].
#![allow(unused_macros)]

macro_rules! case {
    ($name:ident, $call:expr) => {
        #[test]
        fn $name() {
            $call;
        }
    };
}

macro_rules! nine {
    ($f:ident) => {
        $f::<
            rust_kzg_mi355x::FsFr,
            rust_kzg_mi355x::MiG1,
            rust_kzg_mi355x::FsG2,
            rust_kzg_mi355x::FsPoly,
            rust_kzg_mi355x::MiFFTSettings,
            rust_kzg_mi355x::MiKZGSettings,
            rust_kzg_mi355x::FsFp,
            rust_kzg_mi355x::g1::MiG1Affine,
            rust_kzg_mi355x::g1::MiG1ProjAddAffine,
        >
    };
}

// the same nine with Poly first (the argument-validation tests of kzg-bench take them in that order)
macro_rules! nine_poly_first {
    ($f:ident) => {
        $f::<
            rust_kzg_mi355x::FsPoly,
            rust_kzg_mi355x::FsFr,
            rust_kzg_mi355x::MiG1,
            rust_kzg_mi355x::FsG2,
            rust_kzg_mi355x::MiFFTSettings,
            rust_kzg_mi355x::MiKZGSettings,
            rust_kzg_mi355x::FsFp,
            rust_kzg_mi355x::g1::MiG1Affine,
            rust_kzg_mi355x::g1::MiG1ProjAddAffine,
        >
    };
}

// (Fr, G1, Fp, G1Affine, G1ProjAddAffine): the linear-combination tests
macro_rules! five {
    ($f:ident) => {
        $f::<
            rust_kzg_mi355x::FsFr,
            rust_kzg_mi355x::MiG1,
            rust_kzg_mi355x::FsFp,
            rust_kzg_mi355x::g1::MiG1Affine,
            rust_kzg_mi355x::g1::MiG1ProjAddAffine,
        >
    };
}
