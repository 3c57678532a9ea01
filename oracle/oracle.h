/*
 * TEST INFRASTRUCTURE — CPU oracle for the MI355X KZG hot path.
 *
 * A plain-C restatement of the algorithms the reference (grandinetech/rust-kzg,
 * blst backend) runs on the CPU for the MSM / NTT hot path and its EIP-4844
 * callers.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link or call this; the product library (rust-kzg_amd/csrc) never does.
 *
 * The field / curve arithmetic itself lives in the third-party `blst` crate
 * (pinned 0.3.16, reference Cargo.lock:530-533) whose sources are NOT under
 * /root/reference, so it is restated here from the published algorithms
 * (Montgomery CIOS; EFD add-2008-s / madd-2008-s XYZZ formulas; ZCash
 * compressed-point format) using the in-tree statements of the same constants
 * (zkcrypto/bls12_381/src/fp.rs:70-104, scalar.rs:75-173) and anchored on the
 * reference's call sites.  Parity is PINNED: tests/test_oracle_golden.py checks
 * this oracle against the c-kzg-4844 mainnet vectors and the hard-coded
 * known-answer constants the reference's own tests hold.
 *
 * All field elements are little-endian u64 limbs in Montgomery form, i.e.
 * bit-identical to blst_fp / blst_fr / blst_p1 / blst_p1_affine
 * (reference kzg/src/eth/c_bindings.rs:429-474).
 */
#ifndef KZG_ORACLE_H
#define KZG_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[6]; } ofp_t;                 /* blst_fp  */
typedef struct { uint64_t l[4]; } ofr_t;                 /* blst_fr  */
typedef struct { ofp_t x, y; } og1_affine_t;             /* blst_p1_affine; infinity = (0,0) */
typedef struct { ofp_t x, y, z; } og1_t;                 /* blst_p1 (Jacobian); infinity = Z==0 */
typedef struct { ofp_t x, y, zzz, zz; } og1_xyzz_t;      /* kzg/src/msm/pippenger_utils.rs:5-12 */

/* ---- Fp (fp.c) ---- */
void ofp_add(ofp_t *r, const ofp_t *a, const ofp_t *b);
void ofp_sub(ofp_t *r, const ofp_t *a, const ofp_t *b);
void ofp_neg(ofp_t *r, const ofp_t *a);
void ofp_mul(ofp_t *r, const ofp_t *a, const ofp_t *b);
void ofp_sqr(ofp_t *r, const ofp_t *a);
void ofp_inv(ofp_t *r, const ofp_t *a);
int  ofp_sqrt(ofp_t *r, const ofp_t *a);                 /* 1 if a is a square */
int  ofp_is_zero(const ofp_t *a);
int  ofp_eq(const ofp_t *a, const ofp_t *b);
void ofp_one(ofp_t *r);
int  ofp_from_be48(ofp_t *r, const uint8_t in[48]);      /* 0 if >= p */
void ofp_to_be48(uint8_t out[48], const ofp_t *a);
int  ofp_is_lex_largest(const ofp_t *a);                 /* a > (p-1)/2 */

/* ---- Fr (fr.c) ---- */
void ofr_add(ofr_t *r, const ofr_t *a, const ofr_t *b);
void ofr_sub(ofr_t *r, const ofr_t *a, const ofr_t *b);
void ofr_neg(ofr_t *r, const ofr_t *a);
void ofr_mul(ofr_t *r, const ofr_t *a, const ofr_t *b);
void ofr_sqr(ofr_t *r, const ofr_t *a);
void ofr_inv(ofr_t *r, const ofr_t *a);
void ofr_pow(ofr_t *r, const ofr_t *a, uint64_t e);
int  ofr_is_zero(const ofr_t *a);
int  ofr_is_one(const ofr_t *a);
int  ofr_eq(const ofr_t *a, const ofr_t *b);
void ofr_zero(ofr_t *r);
void ofr_one(ofr_t *r);
void ofr_from_u64(ofr_t *r, uint64_t v);
void ofr_from_u64_arr(ofr_t *r, const uint64_t v[4]);    /* canonical LE limbs -> Montgomery */
void ofr_to_u64_arr(uint64_t v[4], const ofr_t *a);      /* Montgomery -> canonical LE limbs */
int  ofr_from_be32(ofr_t *r, const uint8_t in[32]);      /* FsFr::from_bytes: 0 if >= r */
void ofr_from_be32_unchecked(ofr_t *r, const uint8_t in[32]); /* reduces mod r */
void ofr_to_be32(uint8_t out[32], const ofr_t *a);
void ofr_to_scalar_le(uint8_t out[32], const ofr_t *a);  /* FsFr::to_scalar */

/* ---- G1 (g1.c) ---- */
void og1_set_inf(og1_t *r);
int  og1_is_inf(const og1_t *a);
void og1_generator(og1_t *r);
void og1_from_affine(og1_t *r, const og1_affine_t *a);
void og1_to_affine(og1_affine_t *r, const og1_t *a);
void og1_add_or_dbl(og1_t *r, const og1_t *a, const og1_t *b);
void og1_dbl(og1_t *r, const og1_t *a);
void og1_neg(og1_t *r, const og1_t *a);
void og1_mul(og1_t *r, const og1_t *a, const ofr_t *s);  /* double-and-add */
int  og1_equal(const og1_t *a, const og1_t *b);          /* projective equivalence */
int  og1_affine_on_curve(const og1_affine_t *a);
int  og1_in_subgroup(const og1_t *a);
int  og1_uncompress(og1_affine_t *r, const uint8_t in[48]); /* 0 on invalid encoding */
void og1_compress(uint8_t out[48], const og1_t *a);
void og1_xyzz_dadd_affine(og1_xyzz_t *out, const og1_affine_t *p, int subtract);
void og1_xyzz_dadd(og1_xyzz_t *out, const og1_xyzz_t *p);
void og1_xyzz_to_jacobian(og1_t *out, const og1_xyzz_t *in);

/* ---- MSM (msm.c) ---- */
size_t opippenger_window_size(size_t npoints);
/* kzg/src/msm/tiling_pippenger_ops.rs:106-138: scalars are canonical 32-byte LE */
void omsm_tiling_pippenger(og1_t *out, const og1_affine_t *points, const uint8_t *scalars_le, size_t n);
/* g1_linear_combination (blst/src/kzg_proofs.rs:25-72 + kzg/src/msm/msm_impls.rs:114-148):
 * Jacobian points, Montgomery scalars; len<8 naive; infinity points filtered */
void og1_lincomb(og1_t *out, const og1_t *points, const ofr_t *scalars, size_t n);
/* same over affine points + Montgomery scalars (the sppark-boundary shape) */
void omsm_affine(og1_t *out, const og1_affine_t *points, const ofr_t *scalars, size_t n);
/* naive sum of double-and-add products */
void omsm_naive(og1_t *out, const og1_affine_t *points, const ofr_t *scalars, size_t n);
/* nthreads-way split of omsm_affine over point ranges (CPU baseline only) */
void omsm_affine_mt(og1_t *out, const og1_affine_t *points, const ofr_t *scalars, size_t n, int nthreads);

/* BGMW fixed-base MSM (kzg/src/msm/bgmw.rs): table[j * n + i] = affine(2^(window j) P_i), j < h */
typedef struct {
    size_t n, window, h;
    og1_affine_t *table;
} obgmw_table_t;
size_t obgmw_window_size(size_t npoints);
int  obgmw_table_new(obgmw_table_t *t, const og1_affine_t *points, size_t n);
void obgmw_table_free(obgmw_table_t *t);
void obgmw_multiply(og1_t *out, const obgmw_table_t *t, const uint8_t *scalars_le, size_t n);

/* ---- NTT (fft.c) ---- */
typedef struct {
    size_t max_width;
    ofr_t *roots_of_unity;          /* max_width + 1 */
    ofr_t *reverse_roots_of_unity;  /* max_width + 1 */
    ofr_t *brp_roots_of_unity;      /* max_width */
} offt_settings_t;
int  offt_settings_new(offt_settings_t *fs, unsigned scale);
void offt_settings_free(offt_settings_t *fs);
void oscale2_root_of_unity(uint64_t out[4], unsigned scale); /* canonical limbs */
/* 0 ok; 1 len > max_width; 2 not a power of two */
int  offt_fr(const offt_settings_t *fs, ofr_t *out, const ofr_t *in, size_t n, int inverse);
void offt_fr_slow(const offt_settings_t *fs, ofr_t *out, const ofr_t *in, size_t n);
/* 0 ok; 1 empty; 2 not a power of two; 3 too long */
int  odas_fft_extension(const offt_settings_t *fs, ofr_t *odds, const ofr_t *evens, size_t n);
void oreverse_bit_order(void *data, size_t elem_size, size_t n);
/* G1-valued NTT (blst/src/fft_g1.rs): 0 ok; 1 too long; 2 not a power of two */
int  offt_g1(const offt_settings_t *fs, og1_t *out, const og1_t *in, size_t n, int inverse);
void offt_g1_slow(const offt_settings_t *fs, og1_t *out, const og1_t *in, size_t n);

/* ---- SHA-256 (sha256.c) ---- */
void osha256(uint8_t out[32], const uint8_t *in, size_t len);

/* ---- EIP-4844 (eip4844.c) ---- */
#define O_FIELD_ELEMENTS_PER_BLOB 4096
#define O_BYTES_PER_BLOB (32 * 4096)
typedef struct {
    og1_affine_t *g1_lagrange_brp;  /* 4096, bit-reversed Lagrange setup (affine) */
    og1_affine_t *g1_monomial;      /* 4096 */
    uint8_t *g2_monomial_bytes;     /* 65 * 96 raw */
    offt_settings_t fs;             /* scale 13 */
    obgmw_table_t *bgmw;            /* over g1_lagrange_brp; built by the first oblob_to_kzg_commitment_bgmw call */
} osettings_t;
/* text format: kzg/src/eip_4844.rs:151-228.  0 ok, nonzero = BadArgs */
int  oload_trusted_setup_text(osettings_t *s, const char *text, size_t len);
void ofree_trusted_setup(osettings_t *s);
int  oblob_to_fr(ofr_t *out, const uint8_t *blob);                       /* 0 ok */
int  oblob_to_kzg_commitment(uint8_t out[48], const uint8_t *blob, const osettings_t *s);
/* the same through the BGMW table (the reference's default with feature `bgmw`; FsKZGSettings::new builds the table,
 * blst/src/types/kzg_settings.rs:109-123).  Builds the table on first use (~1 s); not thread-safe on that first call. */
int  oblob_to_kzg_commitment_bgmw(uint8_t out[48], const uint8_t *blob, osettings_t *s);
void ocompute_challenge(ofr_t *out, const ofr_t *blob_fr, const uint8_t commitment[48]);
int  oevaluate_polynomial_in_evaluation_form(ofr_t *out, const ofr_t *poly, const ofr_t *x, const osettings_t *s);
int  ocompute_kzg_proof(uint8_t proof[48], uint8_t y[32], const uint8_t *blob, const uint8_t z[32], const osettings_t *s);
int  ocompute_blob_kzg_proof(uint8_t proof[48], const uint8_t *blob, const uint8_t commitment[48], const osettings_t *s);
/* polynomial part of compute_cells (kzg/src/das.rs:244-292): ifft(brp(blob)) -> pad -> fft 8192 -> brp; out = 8192 x 32 B BE */
int  ocompute_cells(uint8_t *cells_out, const uint8_t *blob, const osettings_t *s);
/* KZG multiproof of cell k (0..127), by definition: quotient by X^64 - h_k^64, MSM over the monomial setup */
int  ocompute_cell_proof(uint8_t proof[48], const uint8_t *blob, size_t k, const osettings_t *s);

/* batched verification up to the pairing (kzg/src/eip_4844.rs:328-435): r-powers and the two G1 pairing inputs */
int  ocompute_r_powers(ofr_t *out, const uint8_t *commitments, const uint8_t *zs, const uint8_t *ys, const uint8_t *proofs, size_t n);
int  overify_kzg_proof_batch_g1(og1_t *proof_lincomb, og1_t *rhs, const uint8_t *commitments, const uint8_t *zs,
                                const uint8_t *ys, const uint8_t *proofs, size_t n);

#ifdef __cplusplus
}
#endif
#endif
