//! G1 for the MI355X backend: `FsG1` with `g1_lincomb` on the GPU.
//! Reference shape: blst/src/types/g1.rs (type + trait impls), blst/src/kzg_proofs.rs:25-72 (the sppark branch of
//! g1_linear_combination this replaces).
extern crate alloc;

use alloc::string::String;
use alloc::vec::Vec;

use blst::{blst_fr, blst_p1, blst_p1_affine};
use kzg::msm::precompute::PrecomputationTable;
use kzg::{G1Affine, G1GetFp, G1LinComb, G1Mul, PairingVerify, G1};
use rust_kzg_blst::kzg_proofs::pairings_verify;
use rust_kzg_blst::types::fp::FsFp;
use rust_kzg_blst::types::fr::FsFr;
use rust_kzg_blst::types::g1::{FsG1, FsG1Affine, FsG1ProjAddAffine};
use rust_kzg_blst::types::g2::FsG2;
use rust_kzg_mi355x_sys as sys;

#[repr(transparent)]
#[derive(Debug, Default, Clone, Copy, PartialEq, Eq)]
pub struct MiG1(pub FsG1);

impl MiG1 {
    #[inline]
    pub fn from_blst(p: blst_p1) -> Self {
        MiG1(FsG1(p))
    }
    #[inline]
    pub fn slice_as_fs(points: &[MiG1]) -> &[FsG1] {
        // repr(transparent): same layout
        unsafe { core::slice::from_raw_parts(points.as_ptr() as *const FsG1, points.len()) }
    }
}

impl G1 for MiG1 {
    fn zero() -> Self {
        MiG1(FsG1::zero())
    }
    fn identity() -> Self {
        MiG1(FsG1::identity())
    }
    fn generator() -> Self {
        MiG1(FsG1::generator())
    }
    fn negative_generator() -> Self {
        MiG1(FsG1::negative_generator())
    }
    #[cfg(feature = "rand")]
    fn rand() -> Self {
        MiG1(FsG1::rand())
    }
    fn from_bytes(bytes: &[u8]) -> Result<Self, String> {
        FsG1::from_bytes(bytes).map(MiG1)
    }
    fn from_hex(hex: &str) -> Result<Self, String> {
        FsG1::from_hex(hex).map(MiG1)
    }
    fn to_bytes(&self) -> [u8; 48] {
        self.0.to_bytes()
    }
    fn add_or_dbl(&self, b: &Self) -> Self {
        MiG1(self.0.add_or_dbl(&b.0))
    }
    fn is_inf(&self) -> bool {
        self.0.is_inf()
    }
    fn is_valid(&self) -> bool {
        self.0.is_valid()
    }
    fn dbl(&self) -> Self {
        MiG1(self.0.dbl())
    }
    fn add(&self, b: &Self) -> Self {
        MiG1(self.0.add(&b.0))
    }
    fn sub(&self, b: &Self) -> Self {
        MiG1(self.0.sub(&b.0))
    }
    fn equals(&self, b: &Self) -> bool {
        self.0.equals(&b.0)
    }
    fn add_or_dbl_assign(&mut self, b: &Self) {
        self.0.add_or_dbl_assign(&b.0)
    }
    fn add_assign(&mut self, b: &Self) {
        self.0.add_assign(&b.0)
    }
    fn dbl_assign(&mut self) {
        self.0.dbl_assign()
    }
}

impl G1Mul<FsFr> for MiG1 {
    fn mul(&self, b: &FsFr) -> Self {
        MiG1(self.0.mul(b))
    }
}

impl G1GetFp<FsFp> for MiG1 {
    fn x(&self) -> &FsFp {
        self.0.x()
    }
    fn y(&self) -> &FsFp {
        self.0.y()
    }
    fn z(&self) -> &FsFp {
        self.0.z()
    }
    fn x_mut(&mut self) -> &mut FsFp {
        self.0.x_mut()
    }
    fn y_mut(&mut self) -> &mut FsFp {
        self.0.y_mut()
    }
    fn z_mut(&mut self) -> &mut FsFp {
        self.0.z_mut()
    }
}

impl PairingVerify<MiG1, FsG2> for MiG1 {
    /// The pairing stays on the CPU in every backend of the reference (blst/src/kzg_proofs.rs:73-100).
    fn verify(a1: &MiG1, a2: &FsG2, b1: &MiG1, b2: &FsG2) -> bool {
        pairings_verify(&a1.0, a2, &b1.0, b2)
    }
}

/// Affine form used by the generic MSM code paths (kzg::msm); the GPU path converts in bulk instead.
#[repr(transparent)]
#[derive(Debug, Default, Clone, Copy, PartialEq, Eq)]
pub struct MiG1Affine(pub FsG1Affine);

impl G1Affine<MiG1, FsFp> for MiG1Affine {
    fn zero() -> Self {
        MiG1Affine(<FsG1Affine as G1Affine<FsG1, FsFp>>::zero())
    }
    fn into_affine(g1: &MiG1) -> Self {
        MiG1Affine(FsG1Affine::into_affine(&g1.0))
    }
    fn into_affines_loc(out: &mut [Self], g1: &[MiG1]) {
        let out = unsafe { core::slice::from_raw_parts_mut(out.as_mut_ptr() as *mut FsG1Affine, out.len()) };
        FsG1Affine::into_affines_loc(out, MiG1::slice_as_fs(g1))
    }
    fn to_proj(&self) -> MiG1 {
        MiG1(self.0.to_proj())
    }
    fn x(&self) -> &FsFp {
        self.0.x()
    }
    fn y(&self) -> &FsFp {
        self.0.y()
    }
    fn is_infinity(&self) -> bool {
        self.0.is_infinity()
    }
    fn x_mut(&mut self) -> &mut FsFp {
        self.0.x_mut()
    }
    fn y_mut(&mut self) -> &mut FsFp {
        self.0.y_mut()
    }
}

pub type MiG1ProjAddAffine = FsG1ProjAddAffine;
pub type MiPrecomputation = PrecomputationTable<FsFr, MiG1, FsFp, MiG1Affine, MiG1ProjAddAffine>;

/// `g1_linear_combination` (blst/src/kzg_proofs.rs:25-72) with the GPU behind it.
///   * len < 8: plain sum of products on the CPU, exactly like the reference (:37-45) — the cut-off is part of
///     the reference's behaviour, and eight scalar multiplications are faster on a core than a PCIe round trip;
///   * a precomputation handle (built by `MiKZGSettings::new`): `mult_pippenger_prepared` on its device table;
///   * otherwise: `mult_pippenger` (variable-base engine; the bases must lie in G1, which every caller in
///     kzg/src guarantees — setup points and validated commitments / proofs).
pub fn g1_linear_combination(
    out: &mut MiG1,
    points: &[MiG1],
    scalars: &[FsFr],
    len: usize,
    precomputation: Option<&MiPrecomputation>,
) {
    if len < 8 {
        *out = MiG1::default();
        for i in 0..len {
            let tmp = points[i].mul(&scalars[i]);
            out.add_or_dbl_assign(&tmp);
        }
        return;
    }
    // scalars cross the boundary as Montgomery blst_fr (the sppark convention, kzg_proofs.rs:47-48)
    let scalars_raw = unsafe { core::slice::from_raw_parts(scalars.as_ptr() as *const blst_fr, len) };
    let result: blst_p1;
    if let Some(table) = precomputation {
        result = unsafe { sys::msm_prepared_raw(table.table, scalars_raw) }.expect("mult_pippenger_prepared");
    } else {
        // affine bases, infinity as (0, 0) (kzg_proofs.rs:53-57)
        let mut affines: Vec<MiG1Affine> = alloc::vec![MiG1Affine::default(); len];
        MiG1Affine::into_affines_loc(&mut affines, &points[..len]);
        let affines_raw = unsafe { core::slice::from_raw_parts(affines.as_ptr() as *const blst_p1_affine, len) };
        result = sys::msm(affines_raw, scalars_raw).expect("mult_pippenger");
    }
    *out = MiG1::from_blst(result);
}

impl G1LinComb<FsFr, FsFp, MiG1Affine, MiG1ProjAddAffine> for MiG1 {
    fn g1_lincomb(points: &[Self], scalars: &[FsFr], len: usize, precomputation: Option<&MiPrecomputation>) -> Self {
        let mut out = MiG1::default();
        g1_linear_combination(&mut out, points, scalars, len, precomputation);
        out
    }

    /// Batched form (kzg/src/lib.rs:159-181).  Its one caller is `compute_fk20_proofs` (kzg/src/das.rs:682-686),
    /// which passes the 128 rows of `x_ext_fft_columns` together with the settings' precomputation.  As in the
    /// reference (`precomputation.multiply_batch(scalars)`, kzg/src/msm/bgmw.rs:306-380) the table is the one
    /// `MiKZGSettings::new` built from those rows (`kzgamd_msm_attach_matrix`): all rows in ONE launch of the device's
    /// wide-table path (`kzgamd_mult_pippenger_matrix`).  Without a table — or when the matrix did not fit the HBM
    /// budget and was not attached — every row is a variable-base MSM over its points, like the trait's default.
    fn g1_lincomb_batch(
        points: &[Vec<Self>],
        scalars: &[Vec<FsFr>],
        precomputation: Option<&MiPrecomputation>,
    ) -> Result<Vec<Self>, String> {
        if points.len() != scalars.len() {
            return Err("Invalid batch size".into());
        }
        for (p, s) in points.iter().zip(scalars.iter()) {
            if p.len() != s.len() {
                return Err("Invalid point count length".into());
            }
        }
        if let Some(table) = precomputation {
            let rows = scalars.len();
            let cols = scalars.first().map_or(0, |r| r.len());
            if rows > 0 && cols > 0 && scalars.iter().all(|r| r.len() == cols) {
                let flat: Vec<FsFr> = scalars.iter().flat_map(|r| r.iter().copied()).collect();
                let flat_raw = unsafe { core::slice::from_raw_parts(flat.as_ptr() as *const blst_fr, flat.len()) };
                // Err: no matrix attached to this handle (or another shape): fall through to the row-by-row form
                if let Ok(sums) = unsafe { sys::msm_matrix_raw(table.table, flat_raw, rows) } {
                    return Ok(sums.into_iter().map(MiG1::from_blst).collect());
                }
            }
        }
        let mut result = Vec::with_capacity(points.len());
        for (p, s) in points.iter().zip(scalars.iter()) {
            result.push(Self::g1_lincomb(p, s, p.len(), None));
        }
        Ok(result)
    }
}
