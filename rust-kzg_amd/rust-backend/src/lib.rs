//! `rust-kzg-mi355x`: a rust-kzg backend whose hot trait methods run on an AMD MI355X.
//!
//! Element types are the blst backend's (`FsFr`, `FsFp`, `FsG2`, `FsPoly` are used as they are: nothing on the
//! hot path lives in them); the three types that own a hot method are thin wrappers:
//!
//! | this crate            | wraps                  | routed to the GPU                                             |
//! |-----------------------|------------------------|---------------------------------------------------------------|
//! | [`g1::MiG1`]          | `FsG1`                 | `G1LinComb::g1_lincomb` (-> `mult_pippenger[_prepared]`)       |
//! | [`fft_settings::MiFFTSettings`] | `FsFFTSettings` | `FFTFr::fft_fr`, `FFTG1::fft_g1`, `DASExtension::das_fft_extension` |
//! | [`kzg_settings::MiKZGSettings`] | — (own fields)  | `KZGSettings::new` (builds the device table), `commit_to_poly`, `compute_proof_single` |
//!
//! Mirrors blst/src/types/{g1,fft_settings,kzg_settings}.rs method for method; every other method delegates to
//! the wrapped blst type.  Not compiled in the build image: see Cargo.toml.
#![cfg_attr(not(feature = "std"), no_std)]
extern crate alloc;

pub mod backend;
pub mod fft_settings;
pub mod g1;
pub mod kzg_settings;

pub use backend::MiBackend;
pub use fft_settings::MiFFTSettings;
pub use g1::MiG1;
pub use kzg_settings::MiKZGSettings;
pub use rust_kzg_blst::types::{fp::FsFp, fr::FsFr, g2::FsG2, poly::FsPoly};
