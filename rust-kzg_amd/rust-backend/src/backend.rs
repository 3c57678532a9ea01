//! The type bundle the generic test-suite and DAS code are instantiated with (`kzg::EcBackend`), the counterpart of
//! `rust_kzg_blst::eip_7594::BlstBackend` (blst/src/eip_7594.rs:16-28), plus the setup loaders / generators the
//! suite's entry points take (blst/src/eip_4844.rs:100-158, blst/src/utils.rs `generate_trusted_setup`).
extern crate alloc;

use alloc::string::{String, ToString};
use alloc::vec::Vec;

use kzg::eip_4844::{load_trusted_setup_string, FIELD_ELEMENTS_PER_CELL};
use kzg::{EcBackend, Fr, G1Mul, G2Mul, KZGSettings, G1, G2};
use rust_kzg_blst::types::fp::FsFp;
use rust_kzg_blst::types::fr::FsFr;
use rust_kzg_blst::types::g2::FsG2;
use rust_kzg_blst::types::poly::FsPoly;

use crate::fft_settings::MiFFTSettings;
use crate::g1::{MiG1, MiG1Affine, MiG1ProjAddAffine};
use crate::kzg_settings::MiKZGSettings;

pub struct MiBackend;

impl EcBackend for MiBackend {
    type Fr = FsFr;
    type G1Fp = FsFp;
    type G1Affine = MiG1Affine;
    type G1 = MiG1;
    type G2 = FsG2;
    type Poly = FsPoly;
    type FFTSettings = MiFFTSettings;
    type KZGSettings = MiKZGSettings;
    type G1ProjAddAffine = MiG1ProjAddAffine;
}

/// blst/src/eip_4844.rs:100-144 with this backend's types: the three byte arrays -> points -> `MiKZGSettings::new`
/// (which builds the device tables).
pub fn load_trusted_setup_rust(
    g1_monomial_bytes: &[u8],
    g1_lagrange_bytes: &[u8],
    g2_monomial_bytes: &[u8],
) -> Result<MiKZGSettings, String> {
    let mut g1_monomial = g1_monomial_bytes
        .chunks(48)
        .map(MiG1::from_bytes)
        .collect::<Result<Vec<_>, _>>()?;
    let g1_lagrange = g1_lagrange_bytes
        .chunks(48)
        .map(MiG1::from_bytes)
        .collect::<Result<Vec<_>, _>>()?;
    let g2_monomial = g2_monomial_bytes
        .chunks(96)
        .map(FsG2::from_bytes)
        .collect::<Result<Vec<_>, _>>()?;
    let mut max_scale = 0usize;
    while (1usize << max_scale) < core::cmp::max(g1_monomial.len(), g2_monomial.len()) {
        max_scale += 1;
    }
    let fs = <MiFFTSettings as kzg::FFTSettings<FsFr>>::new(max_scale)?;
    // the Lagrange points are stored bit-reversed, as in the reference
    let mut g1_lagrange_brp = g1_lagrange;
    kzg::common_utils::reverse_bit_order(&mut g1_lagrange_brp)?;
    let _ = &mut g1_monomial;
    MiKZGSettings::new(&g1_monomial, &g1_lagrange_brp, &g2_monomial, &fs, FIELD_ELEMENTS_PER_CELL)
}

#[cfg(feature = "std")]
pub fn load_trusted_setup_filename_rust(filepath: &str) -> Result<MiKZGSettings, String> {
    let contents = std::fs::read_to_string(filepath).map_err(|_| "Unable to open file".to_string())?;
    let (g1_monomial, g1_lagrange, g2_monomial) = load_trusted_setup_string(&contents)?;
    load_trusted_setup_rust(&g1_monomial, &g1_lagrange, &g2_monomial)
}

/// blst/src/utils.rs `generate_trusted_setup`: powers of a secret for the unit tests (insecure by construction).
pub fn generate_trusted_setup(n: usize, secret: [u8; 32usize]) -> (Vec<MiG1>, Vec<MiG1>, Vec<FsG2>) {
    let s = FsFr::hash_to_bls_field(&secret);
    let mut s_pow = FsFr::one();
    let mut s1 = Vec::with_capacity(n);
    let mut s2 = Vec::with_capacity(n);
    let mut s3 = Vec::with_capacity(n);
    for _ in 0..n {
        s1.push(MiG1::generator().mul(&s_pow));
        s2.push(MiG1::generator()); // unused by the tests that take the monomial form
        s3.push(FsG2::generator().mul(&s_pow));
        s_pow = s_pow.mul(&s);
    }
    (s1, s2, s3)
}
