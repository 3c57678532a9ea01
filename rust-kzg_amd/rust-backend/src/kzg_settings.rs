//! KZG settings for the MI355X backend.
//! Reference shape: blst/src/types/kzg_settings.rs:26-260 (`FsKZGSettings` and its `KZGSettings` impl); the
//! `sppark` arm of `new` (:109-123) is the model for the device table.
extern crate alloc;

use alloc::string::{String, ToString};
use alloc::sync::Arc;
use alloc::vec::Vec;

use blst::blst_p1_affine;
use kzg::{FFTFr, FFTSettings, Fr, G1Affine, G1Mul, G2Mul, KZGSettings, Poly, FFTG1, G1, G2};
use rust_kzg_blst::kzg_proofs::pairings_verify;
use rust_kzg_blst::types::fp::FsFp;
use rust_kzg_blst::types::fr::FsFr;
use rust_kzg_blst::types::g2::FsG2;
use rust_kzg_blst::types::poly::FsPoly;
use rust_kzg_mi355x_sys as sys;

use crate::fft_settings::MiFFTSettings;
use crate::g1::{g1_linear_combination, MiG1, MiG1Affine, MiG1ProjAddAffine, MiPrecomputation};

#[derive(Debug, Clone, Default)]
pub struct MiKZGSettings {
    pub fs: MiFFTSettings,
    pub g1_values_monomial: Vec<MiG1>,
    pub g1_values_lagrange_brp: Vec<MiG1>,
    pub g2_values_monomial: Vec<FsG2>,
    /// device-resident fixed-base tables behind ONE handle: over `g1_values_lagrange_brp` (`prepare_msm`, what
    /// `g1_lincomb` multiplies by) and over the rows of `x_ext_fft_columns` (`kzgamd_msm_attach_matrix`, what
    /// `g1_lincomb_batch` multiplies by) — the reference's `precompute(points, matrix)`
    pub precomputation: Option<Arc<MiPrecomputation>>,
    pub x_ext_fft_columns: Vec<Vec<MiG1>>,
    pub cell_size: usize,
    /// whether the FK20 matrix table was attached; `false` = it did not fit its budget (or the device refused) and
    /// `g1_lincomb_batch` runs row by row — correct, two orders of magnitude slower; callers that care can look
    pub matrix_table_attached: bool,
}

/// HBM each of the two fixed-base tables of ONE settings object may take.  The library's default (160 GB per table,
/// capped by what is free) is right for a process that owns a GPU and one settings object; the rust-kzg test and bench
/// suites create several `KZGSettings` side by side (and in parallel), and two tables at the default would leave a
/// second object with tiny tables or none.  24 GB buys window c = 13 (20 additions per scalar: 80 % of the default
/// table's commitments/s, DESIGN.md §6 "throughput per table budget"); 16 GB holds the 128 x 64 FK20 matrix at c = 13.
/// [`MiKZGSettings::set_table_budgets`] rebuilds the tables with other values (0 = the library's default).
pub const LAGRANGE_TABLE_BUDGET_BYTES: u64 = 24 << 30;
pub const MATRIX_TABLE_BUDGET_BYTES: u64 = 16 << 30;

fn prepare(points: &[MiG1], matrix: &[Vec<MiG1>], lagrange_budget: u64, matrix_budget: u64) -> (Option<Arc<MiPrecomputation>>, bool) {
    let mut affines: Vec<MiG1Affine> = alloc::vec![MiG1Affine::default(); points.len()];
    MiG1Affine::into_affines_loc(&mut affines, points);
    let raw = unsafe { core::slice::from_raw_parts(affines.as_ptr() as *const blst_p1_affine, affines.len()) };
    let mut cfg = sys::KzgAmdConfig::default();
    cfg.table_budget_bytes = lagrange_budget;
    let handle = sys::prepare_raw_with(raw, &cfg);
    if handle.is_null() {
        return (None, false);
    }
    let mut attached = false;
    // the matrix rows, flattened row-major (every row has the same length: cell_size)
    let rows = matrix.len();
    let cols = matrix.first().map_or(0, |r| r.len());
    if rows > 0 && cols > 0 && matrix.iter().all(|r| r.len() == cols) {
        let flat: Vec<MiG1> = matrix.iter().flat_map(|r| r.iter().copied()).collect();
        let mut flat_aff: Vec<MiG1Affine> = alloc::vec![MiG1Affine::default(); flat.len()];
        MiG1Affine::into_affines_loc(&mut flat_aff, &flat);
        let flat_raw = unsafe { core::slice::from_raw_parts(flat_aff.as_ptr() as *const blst_p1_affine, flat_aff.len()) };
        // a matrix that does not fit the HBM budget is not an error — g1_lincomb_batch then runs row by row — but it is
        // recorded (MiKZGSettings::matrix_table_attached) and, with KZGAMD_VERBOSE set, said on stderr
        let mut mcfg = sys::KzgAmdConfig::default();
        mcfg.table_budget_bytes = matrix_budget;
        match unsafe { sys::attach_matrix_raw(handle, flat_raw, rows, cols, Some(&mcfg)) } {
            Ok(()) => attached = true,
            Err(_e) => {
                #[cfg(feature = "std")]
                if std::env::var_os("KZGAMD_VERBOSE").is_some() {
                    eprintln!("rust-kzg-mi355x: FK20 matrix table not attached ({} x {}, budget {} bytes): {}", rows, cols, matrix_budget, _e);
                }
            }
        }
    }
    (Some(Arc::new(MiPrecomputation::from_ptr(handle))), attached)
}

impl MiKZGSettings {
    /// Rebuilds the two device tables with other HBM budgets (bytes per table; 0 = the library's default of 160 GB
    /// capped by the free HBM — for a process with ONE settings object on a GPU of its own).  The old tables are
    /// released first if this object is their last owner (otherwise when the last clone that shares them goes away).
    pub fn set_table_budgets(&mut self, lagrange_budget_bytes: u64, matrix_budget_bytes: u64) {
        // free the old tables first (they may hold most of the HBM) — if this object is their last owner
        if let Some(old) = self.precomputation.take() {
            if let Ok(old) = Arc::try_unwrap(old) {
                unsafe { sys::free_raw(old.table) };
            }
        }
        let (pre, attached) = prepare(&self.g1_values_lagrange_brp, &self.x_ext_fft_columns, lagrange_budget_bytes, matrix_budget_bytes);
        self.precomputation = pre;
        self.matrix_table_attached = attached;
    }
}

impl KZGSettings<FsFr, MiG1, FsG2, MiFFTSettings, FsPoly, FsFp, MiG1Affine, MiG1ProjAddAffine> for MiKZGSettings {
    /// blst/src/types/kzg_settings.rs:66-136.  The FK20 columns are computed the reference's way — k2 = 2 * n /
    /// cell_size G1-valued transforms of length k2 per offset — but through `FFTG1::fft_g1`, i.e. on the GPU.
    fn new(
        g1_monomial: &[MiG1],
        g1_lagrange_brp: &[MiG1],
        g2_monomial: &[FsG2],
        fft_settings: &MiFFTSettings,
        cell_size: usize,
    ) -> Result<Self, String> {
        if g1_monomial.len() != g1_lagrange_brp.len() {
            return Err("G1 point length mismatch".to_string());
        }
        let n = g1_monomial.len();
        let k = n / cell_size;
        let k2 = 2 * k;
        let mut x_ext_fft_columns = alloc::vec![alloc::vec![MiG1::default(); cell_size]; k2];
        for offset in 0..cell_size {
            // x = [ s^(n - cell_size - 1 - offset - i * cell_size) ]_{i < k - 1}, identity, zero-padded to k2
            let start = n - cell_size - 1 - offset;
            let mut x_ext = alloc::vec![MiG1::identity(); k2];
            for (i, slot) in x_ext.iter_mut().enumerate().take(k - 1) {
                *slot = g1_monomial[start - i * cell_size];
            }
            if k2 > 2 * n || !k2.is_power_of_two() {
                return Err("Invalid input size".to_string());
            }
            let column = fft_settings.fft_g1(&x_ext, false)?;
            for (row, value) in column.into_iter().enumerate() {
                x_ext_fft_columns[row][offset] = value;
            }
        }
        let (precomputation, matrix_table_attached) =
            prepare(g1_lagrange_brp, &x_ext_fft_columns, LAGRANGE_TABLE_BUDGET_BYTES, MATRIX_TABLE_BUDGET_BYTES);
        Ok(Self {
            g1_values_monomial: g1_monomial.to_vec(),
            g1_values_lagrange_brp: g1_lagrange_brp.to_vec(),
            g2_values_monomial: g2_monomial.to_vec(),
            fs: fft_settings.clone(),
            precomputation,
            x_ext_fft_columns,
            cell_size,
            matrix_table_attached,
        })
    }

    fn commit_to_poly(&self, poly: &FsPoly) -> Result<MiG1, String> {
        if poly.coeffs.len() > self.g1_values_monomial.len() {
            return Err(String::from("Polynomial is longer than secret g1"));
        }
        let mut out = MiG1::default();
        g1_linear_combination(&mut out, &self.g1_values_monomial, &poly.coeffs, poly.coeffs.len(), None);
        Ok(out)
    }

    fn compute_proof_single(&self, p: &FsPoly, x: &FsFr) -> Result<MiG1, String> {
        if p.coeffs.is_empty() {
            return Err(String::from("Polynomial must not be empty"));
        }
        // synthetic division by (X - x), highest coefficient first
        let mut q: Vec<FsFr> = p.coeffs[1..].to_vec();
        for i in (1..q.len()).rev() {
            let carry = q[i].mul(x);
            q[i - 1] = q[i - 1].add(&carry);
        }
        self.commit_to_poly(&FsPoly { coeffs: q })
    }

    fn check_proof_single(&self, com: &MiG1, proof: &MiG1, x: &FsFr, y: &FsFr) -> Result<bool, String> {
        // e(com - [y]G1, G2) == e(proof, [s]G2 - [x]G2); the pairing is CPU code in every backend
        let s_minus_x = self.g2_values_monomial[1].sub(&FsG2::generator().mul(x));
        let com_minus_y = com.sub(&MiG1::generator().mul(y));
        Ok(pairings_verify(&com_minus_y.0, &FsG2::generator(), &proof.0, &s_minus_x))
    }

    fn compute_proof_multi(&self, p: &FsPoly, x0: &FsFr, n: usize) -> Result<MiG1, String> {
        if p.coeffs.is_empty() {
            return Err(String::from("Polynomial must not be empty"));
        }
        if !n.is_power_of_two() {
            return Err(String::from("n must be a power of two"));
        }
        // divisor X^n - x0^n, quotient by the blst backend's polynomial division, commitment on the GPU
        let mut divisor = FsPoly { coeffs: alloc::vec![FsFr::zero(); n + 1] };
        divisor.coeffs[0] = x0.pow(n).negate();
        divisor.coeffs[n] = FsFr::one();
        let mut dividend = p.clone();
        let q = dividend.div(&divisor)?;
        self.commit_to_poly(&q)
    }

    fn check_proof_multi(&self, com: &MiG1, proof: &MiG1, x: &FsFr, ys: &[FsFr], n: usize) -> Result<bool, String> {
        if !n.is_power_of_two() {
            return Err(String::from("n is not a power of two"));
        }
        // interpolation polynomial of ys on the coset x * <w_n>: inverse NTT (GPU), then unscale by x^-i
        let mut interp = FsPoly { coeffs: self.fs.fft_fr(ys, true)? };
        let inv_x = x.inverse();
        let mut pw = inv_x;
        for c in interp.coeffs.iter_mut().skip(1) {
            *c = c.mul(&pw);
            pw = pw.mul(&inv_x);
        }
        let xn2 = FsG2::generator().mul(&x.pow(n));
        let xn_minus_yn = self.g2_values_monomial[n].sub(&xn2);
        let is1 = self.commit_to_poly(&interp)?;
        let commit_minus_interp = com.sub(&is1);
        Ok(pairings_verify(&commit_minus_interp.0, &FsG2::generator(), &proof.0, &xn_minus_yn))
    }

    fn get_roots_of_unity_at(&self, i: usize) -> FsFr {
        self.fs.get_roots_of_unity_at(i)
    }
    fn get_fft_settings(&self) -> &MiFFTSettings {
        &self.fs
    }
    fn get_g1_monomial(&self) -> &[MiG1] {
        &self.g1_values_monomial
    }
    fn get_g1_lagrange_brp(&self) -> &[MiG1] {
        &self.g1_values_lagrange_brp
    }
    fn get_g2_monomial(&self) -> &[FsG2] {
        &self.g2_values_monomial
    }
    fn get_precomputation(&self) -> Option<&MiPrecomputation> {
        self.precomputation.as_ref().map(|v| v.as_ref())
    }
    fn get_x_ext_fft_columns(&self) -> &[Vec<MiG1>] {
        &self.x_ext_fft_columns
    }
    fn get_cell_size(&self) -> usize {
        self.cell_size
    }
}

/// The device table is owned by the last clone of the settings.
impl Drop for MiKZGSettings {
    fn drop(&mut self) {
        if let Some(table) = self.precomputation.take() {
            if let Ok(table) = Arc::try_unwrap(table) {
                unsafe { sys::free_raw(table.table) };
            }
        }
    }
}
