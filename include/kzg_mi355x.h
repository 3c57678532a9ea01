/*
 * kzg_mi355x — C ABI of the MI355X-native KZG hot path (libkzg_mi355x.so).
 *
 * Plain pointers and sizes only.  Every entry point names the reference interface it
 * replaces (grandinetech/rust-kzg @ 2025-12-12, paths relative to the repo root).
 * Field elements / points use blst's in-memory layout (kzg/src/eth/c_bindings.rs:429-474):
 * little-endian u64 limbs in Montgomery form.
 *
 * All functions need a gfx950 device; without one they fail (non-zero code / NULL handle)
 * — there is no CPU fallback in this library.
 */
#ifndef KZG_MI355X_H
#define KZG_MI355X_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

/* libkzg_mi355x_prefixed.so (python rust-kzg_amd/build.py --prefixed): the same library with every c-kzg-4844 name
 * exported as kzgamd_ckzg_<name>, for a process that also links the reference's own C bindings (e.g. to compare the
 * two backends in one test binary).  Define
 * KZG_MI355X_PREFIXED before including this header and keep writing the plain names. */
#ifdef KZG_MI355X_PREFIXED
#define load_trusted_setup kzgamd_ckzg_load_trusted_setup
#define load_trusted_setup_file kzgamd_ckzg_load_trusted_setup_file
#define free_trusted_setup kzgamd_ckzg_free_trusted_setup
#define blob_to_kzg_commitment kzgamd_ckzg_blob_to_kzg_commitment
#define compute_kzg_proof kzgamd_ckzg_compute_kzg_proof
#define compute_blob_kzg_proof kzgamd_ckzg_compute_blob_kzg_proof
#define verify_kzg_proof kzgamd_ckzg_verify_kzg_proof
#define verify_blob_kzg_proof kzgamd_ckzg_verify_blob_kzg_proof
#define verify_blob_kzg_proof_batch kzgamd_ckzg_verify_blob_kzg_proof_batch
#define compute_challenge kzgamd_ckzg_compute_challenge
#define bytes_to_kzg_commitment kzgamd_ckzg_bytes_to_kzg_commitment
#define bytes_from_bls_field kzgamd_ckzg_bytes_from_bls_field
#define compute_cells_and_kzg_proofs kzgamd_ckzg_compute_cells_and_kzg_proofs
#define recover_cells_and_kzg_proofs kzgamd_ckzg_recover_cells_and_kzg_proofs
#define verify_cell_kzg_proof_batch kzgamd_ckzg_verify_cell_kzg_proof_batch
#define compute_verify_cell_kzg_proof_batch_challenge kzgamd_ckzg_compute_verify_cell_kzg_proof_batch_challenge
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } blst_fr;
typedef struct { uint64_t l[6]; } blst_fp;
typedef struct { blst_fp x, y; } blst_p1_affine;     /* infinity = all-zero (blst/src/types/g1.rs:303-316) */
typedef struct { blst_fp x, y, z; } blst_p1;          /* Jacobian; infinity = Z == 0 */

/* sppark's error struct, returned by value (arkworks3-sppark-wlc/sppark/util/rusterror.h:15-27):
 * code 0 = ok; message is malloc'ed (caller frees) or NULL */
typedef struct { int code; char *message; } RustError;

/* ------------------------------------------------------------------------------------------
 * Configuration of a handle (new API; the reference's handles take none — its sppark context sizes itself).  Every
 * creating entry point has an _ex form that takes one; NULL means the defaults, which is what the plain names pass.
 * Values are read once, when the handle is created.  Set struct_size = sizeof(KzgAmdConfig) (kzgamd_config_init does).
 * ------------------------------------------------------------------------------------------ */
#define KZGAMD_NO_TABLES UINT64_MAX   /* table_budget_bytes: build no wide fixed-base table at all (bucket engine) */
typedef struct {
    uint32_t struct_size;          /* sizeof(KzgAmdConfig) of the caller's header */
    int32_t device;                /* GPU ordinal the handle lives on; -1 = the calling thread's current device */
    uint64_t table_budget_bytes;   /* HBM EACH fixed-base table of the handle may take (a settings object has up to three:
                                    * commitments, cell proofs, FK20 columns; tables after the first are built at first
                                    * use).  0 = default: 160 GB, capped by what is free less 12 GB of working room */
    const char *tuning;            /* NULL, or "key=value;key=value": the measured switches (kzgamd_tuning_keys lists
                                    * them).  An unknown key or a value out of range fails the call */
} KzgAmdConfig;
void kzgamd_config_init(KzgAmdConfig *cfg);   /* struct_size, device = -1, no budget, no tuning */
/* the tuning keys, one per line: "name default lo hi meaning"; returns a static string */
const char *kzgamd_tuning_keys(void);

/* ------------------------------------------------------------------------------------------
 * B1 — GPU MSM plug-in.  Same three symbols `rust-kzg-blst` binds under feature `sppark`
 * (blst-sppark/src/lib.rs:8-62, defined today by blst-sppark/cuda/pippenger.cu:23-38).
 * Scalars are blst_fr IN MONTGOMERY FORM (blst/src/kzg_proofs.rs:47-48); out is Jacobian.
 * Thread-safe: any number of threads may call on one handle (the reference shares it through an Arc between rayon
 * workers, kzg/src/msm/sppark.rs:24-44).  Concurrent mult_pippenger_prepared calls of the same length (up to 2^16
 * scalars) are combined into batched launches (up to three in flight); other calls serialise on an internal mutex.
 * Bases may be ANY points of the curve, as in the reference (FsG1::from_bytes does not test the subgroup,
 * blst/src/types/g1.rs:65-87): the result is the plain sum of k_i P_i.  The engines' endomorphism (GLV) split is an
 * identity of the r-torsion subgroup G1 only, so every handle tests its bases once at creation and runs unsplit if one
 * fails; mult_pippenger, whose handle lives for one call, tests up to 2^15 bases and runs unsplit beyond.
 * ------------------------------------------------------------------------------------------ */
void *prepare_msm(const blst_p1_affine points[], size_t npoints);
RustError mult_pippenger_prepared(void *msm, blst_p1 *out, size_t npoints, const blst_fr scalars[]);
RustError mult_pippenger(blst_p1 *out, const blst_p1_affine points[], size_t npoints, const blst_fr scalars[]);

/* Additions (the reference leaks the handle; batched shape follows the wlc variant's
 * mult_pippenger_faster_inf(ctx,out,npoints,batches,scalars), arkworks3-sppark-wlc/src/lib.rs:24-42):
 * scalars is nbatch x npoints, out is nbatch points. */
void free_msm(void *msm);
void *kzgamd_prepare_msm_ex(const blst_p1_affine points[], size_t npoints, const KzgAmdConfig *cfg);
RustError mult_pippenger_prepared_batch(void *msm, blst_p1 out[], size_t npoints, size_t nbatch,
                                        const blst_fr scalars[]);

/* Matrix handle: `rows` independent base sets of `cols` points each in ONE table (points[r * cols + c]) — what the
 * reference's precomputation holds for g1_lincomb_batch (BgmwTable.batch_points, kzg/src/msm/bgmw.rs:206-304; FK20's
 * 128 columns of 64 points, kzg/src/das.rs:682-686) and multiplies in multiply_batch (bgmw.rs:306-380):
 *   out[m * rows + r] = sum_c scalars[(m * rows + r) * cols + c] * points[r * cols + c],   m < nmat
 * in one launch for all rows (and all nmat scalar matrices: nmat = 1 is G1LinComb::g1_lincomb_batch,
 * kzg/src/lib.rs:156-181).  Scalars in Montgomery form, out Jacobian, as mult_pippenger_prepared.  The handle needs a
 * wide table (NULL when none fits cfg's budget: rows * cols * 2^9 * 13 slots of 128 B at the least); free_msm frees it;
 * thread-safe like every handle. */
void *kzgamd_prepare_msm_matrix(const blst_p1_affine points[], size_t rows, size_t cols, const KzgAmdConfig *cfg);
RustError kzgamd_mult_pippenger_matrix(void *msm, blst_p1 out[], const blst_fr scalars[], size_t nmat);
/* The matrix as a part of an existing handle (any handle of prepare_msm / kzgamd_settings_msm_handle): the reference's
 * PrecomputationTable is ONE object built from (points, matrix) (precompute(), kzg/src/msm/bgmw.rs:206-304) and the sppark
 * flavour of it has room for one pointer (kzg/src/msm/sppark.rs:5-22) — after this call kzgamd_mult_pippenger_matrix on
 * `msm` multiplies by the attached matrix, mult_pippenger_prepared by its own points as before; free_msm frees both.
 * cfg: the matrix table's budget and tuning (its device is the handle's).  A handle takes ONE matrix, once: a second
 * attach fails (code 1) and leaves the first in place, so a matrix call in flight never sees its table freed. */
RustError kzgamd_msm_attach_matrix(void *msm, const blst_p1_affine points[], size_t rows, size_t cols, const KzgAmdConfig *cfg);
/* rows and columns of the matrix a handle holds or has attached; 1 when it has none */
int kzgamd_msm_matrix_shape(void *msm, size_t *rows, size_t *cols);

/* Device-resident form used by the batched blob pipeline and bench.py: d_scalars / d_out are
 * device pointers, work is enqueued on `stream` (a hipStream_t, NULL = default stream) and NOT
 * synchronised.  scalars_mont != 0: blst_fr Montgomery limbs; 0: canonical little-endian 256-bit.
 * A handle keeps one workspace per stream it is used on, for up to eight streams: calls enqueued on the same
 * stream are serialised by stream order, calls on different streams may overlap on the GPU (the low-occupancy
 * tail of one batch under the accumulation of the next).  With more streams than workspaces the library makes a
 * stream wait (on the GPU) for the previous use of the workspace it is handed: still correct, no longer overlapped.
 * The host-buffer entry points above synchronise internally and may be called from any thread. */
RustError kzgamd_msm_prepared_batch_device(void *msm, void *d_out, const void *d_scalars, size_t npoints,
                                           size_t nbatch, int scalars_mont, void *stream);
/* Allocates now the workspace `stream` will use for nbatch MSMs of npoints scalars, so that the enqueue calls that
 * follow never call hipMalloc (which synchronises the device).  Without it the first enqueue on a new stream, or
 * with a larger shape, allocates lazily. */
RustError kzgamd_msm_reserve(void *msm, size_t npoints, size_t nbatch, void *stream);
/* the GPU a handle lives on (-1 for NULL) */
int kzgamd_msm_device(void *msm);
/* introspection for benches/tests: window bits, table rows, buckets of a handle */
int kzgamd_msm_info(void *msm, int *window_bits, int *rows, size_t *nbuckets, size_t *npoints);
/* non-zero if the handle holds the wide fixed-base table (rows x npoints x 2^(window_bits-1) affine multiples,
 * built when it fits the handle's table budget (KzgAmdConfig.table_budget_bytes; default 160 GB, capped by free HBM)) and so runs the gather-and-add
 * path: 1 = rows cover the 255-bit scalar (rows additions per scalar), 2 = GLV form, rows cover a 128-bit half
 * (2 x rows additions per scalar; chosen only when every base passed the r-torsion test at prepare time) */
int kzgamd_msm_uses_wide_table(void *msm);
/* HIP-event timing of the bucket-accumulation kernel (k_accum) and of the whole enqueue, recorded on
 * the launch stream; set(on) resets; get returns the number of enqueues averaged (-1 if none) */
int kzgamd_msm_set_profile(void *msm, int on);
int kzgamd_msm_get_profile(void *msm, float *accum_ms, float *total_ms);
/* Handle over DEVICE-resident bases (blst_p1_affine[npoints] in HBM); prepare != 0 builds the fixed-base
 * rows like prepare_msm, 0 gives the variable-base engine mult_pippenger uses. */
void *kzgamd_msm_create_device(const void *d_points_affine, size_t npoints, int prepare);
void *kzgamd_msm_create_device_ex(const void *d_points_affine, size_t npoints, int prepare, const KzgAmdConfig *cfg);
/* bench/test utility: npoints distinct G1 points h_i*G (h_i from splitmix64(seed,i), 248 bits) written
 * as blst_p1_affine into device memory */
RustError kzgamd_generate_points(void *d_out_affine, size_t npoints, uint64_t seed, void *stream);

/* ------------------------------------------------------------------------------------------
 * B2 — NTT plug-in.  Replaces FFTFr::fft_fr / DASExtension::das_fft_extension for FsFFTSettings
 * (kzg/src/lib.rs:421-431; blst/src/fft_fr.rs:112-165; blst/src/data_availability_sampling.rs:78-100).
 * Natural order in, natural order out, Montgomery blst_fr; inverse scales by n^-1.
 * Return codes mirror the reference's error conditions:
 *   ntt_fr:            0 ok, 1 "longer than the available max width", 2 "power-of-two length expected"
 *   das_fft_extension: 0 ok, 1 empty, 2 not a power of two, 3 longer than max width / 2
 *   fft_g1:            as ntt_fr
 *   negative: device error
 * Thread-safe: a handle may be shared between threads (FFTSettings is shared by reference in the reference); concurrent
 * ntt_fr / das_fft_extension calls of the same kind and length (up to 8192 elements) are combined into batched launches,
 * everything else serialises on the handle's mutex.
 * ------------------------------------------------------------------------------------------ */
void *kzgamd_ntt_new(unsigned scale);           /* FsFFTSettings::new(scale), blst/src/types/fft_settings.rs:30-58 */
void *kzgamd_ntt_new_ex(unsigned scale, const KzgAmdConfig *cfg);
void kzgamd_ntt_free(void *ctx);
int ntt_fr(void *ctx, blst_fr *out, const blst_fr *in, size_t n, int inverse);
int das_fft_extension(void *ctx, blst_fr *odds, const blst_fr *evens, size_t half_n);
/* G1-valued transform: FFTG1::fft_g1 for FsFFTSettings (kzg/src/lib.rs:433-435; blst/src/fft_g1.rs:54-83).
 * Jacobian blst_p1 in and out (any valid representation in; infinity = Z == 0), natural order, inverse
 * scales by n^-1.  Outputs equal the reference's as group elements (the Jacobian representative differs).
 * kzgamd_fft_g1_batch runs nbatch independent transforms of length n (contiguous) in the same launches —
 * the shape of the 64 x size-128 transforms of the FK20 setup (blst/src/types/kzg_settings.rs:84-101). */
int fft_g1(void *ctx, blst_p1 *out, const blst_p1 *in, size_t n, int inverse);
int kzgamd_fft_g1_batch(void *ctx, blst_p1 *out, const blst_p1 *in, size_t n, size_t nbatch, int inverse);
/* device-resident, batched (nbatch independent transforms of length n, contiguous), on `stream` */
int kzgamd_ntt_fr_device(void *ctx, void *d_out, const void *d_in, size_t n, size_t nbatch, int inverse, void *stream);
/* das_fft_extension on device-resident data: nbatch contiguous lists of half_n elements; d_scratch (half_n * nbatch
 * elements) must differ from d_odds and d_evens, which must not alias each other (-3).  Two half-size transforms:
 * the twist by the 2n-th roots and the n^-1 ride in their last-pass multiplications. */
int kzgamd_das_fft_extension_device(void *ctx, void *d_odds, const void *d_evens, void *d_scratch, size_t half_n,
                                    size_t nbatch, void *stream);
/* host copies of the settings arrays (FFTSettings getters, kzg/src/lib.rs:465-481); counts in elements */
int kzgamd_ntt_roots(void *ctx, blst_fr *roots /*W+1*/, blst_fr *reverse_roots /*W+1*/, blst_fr *brp_roots /*W*/);
/* The tile plan the NTT kernel runs for (kind, T) — host-only, no GPU needed (rust-kzg_amd/csrc/ntt_plan.h):
 * kind 0 = whole transform of 2^T <= 4096 points, 1 = first pass of a longer one, 2 = later pass; rounds[4*r..] =
 * {first stage, stages, barrier after, element bit}; tab[(r*1024 + thread)*4..] = {idxA, idxB, lds(idxA), lds(idxB)}.
 * Returns the number of rounds (<= 6), -1 on bad arguments. */
int kzgamd_ntt_plan_dump(int kind, int T, int *rounds /* 6 x 4 */, uint16_t *tab /* 6 x 1024 x 4 */);
/* The same for the one-pass DAS extension of lists of 2^T <= 4096 elements (inverse rounds, twist, forward rounds in
 * one tile): rounds[6*r..] = {first stage, stages, barrier after, element bit, flags (1 forward half, 2 unit
 * twiddles, 4 twist before the LDS store), LDS position bit of the element bit}.  Returns the rounds (<= 12). */
int kzgamd_ntt_das_plan_dump(int T, int *rounds /* 12 x 6 */, uint16_t *tab /* 12 x 1024 x 4 */);

/* ------------------------------------------------------------------------------------------
 * B3 — c-kzg-4844 surface for the proving path (blst/src/eip_4844.rs:160-530).
 * Types follow kzg/src/eth/c_bindings.rs:16-113.  Every failure maps to C_KZG_BADARGS like the
 * reference (blst/src/utils.rs:47-56).
 * ------------------------------------------------------------------------------------------ */
typedef enum { C_KZG_OK = 0, C_KZG_BADARGS = 1, C_KZG_ERROR = 2, C_KZG_MALLOC = 3 } C_KZG_RET;
#define BYTES_PER_BLOB 131072
#define FIELD_ELEMENTS_PER_BLOB 4096
typedef struct { uint8_t bytes[32]; } Bytes32;
typedef struct { uint8_t bytes[48]; } Bytes48;
typedef struct { uint8_t bytes[BYTES_PER_BLOB]; } Blob;
typedef Bytes48 KZGCommitment;
typedef Bytes48 KZGProof;

typedef struct { blst_fp fp[2]; } blst_fp2;
typedef struct { blst_fp2 x, y, z; } blst_p2;          /* Jacobian over Fp2; infinity = Z == 0 */

/* Same layout as the reference's CKZGSettings (kzg/src/eth/c_bindings.rs:55-108).  The host arrays are
 * owned by the library and freed by free_trusted_setup; the device-resident state (fixed-base MSM table)
 * is found through a registry keyed by g1_values_lagrange_brp, as the reference does for its tables
 * (kzg/src/eip_4844.rs:64-146).  tables / wbits / scratch_size stay empty like the reference's
 * (blst/src/eip_4844.rs:140-142). */
typedef struct {
    blst_fr *roots_of_unity;          /* 8193 */
    blst_fr *brp_roots_of_unity;      /* 8192 */
    blst_fr *reverse_roots_of_unity;  /* 8193 */
    blst_p1 *g1_values_monomial;      /* 4096 */
    blst_p1 *g1_values_lagrange_brp;  /* 4096 */
    blst_p2 *g2_values_monomial;      /* 65 */
    blst_p1 **x_ext_fft_columns;      /* 128 rows x 64 (FK20 columns, blst/src/types/kzg_settings.rs:84-101) */
    blst_p1_affine **tables;          /* NULL */
    size_t wbits;
    size_t scratch_size;
} CKZGSettings;

/* Exact c-kzg-4844 names and signatures, as exported by blst/src/eip_4844.rs. */
C_KZG_RET load_trusted_setup(CKZGSettings *out, const uint8_t *g1_monomial_bytes, uint64_t num_g1_monomial_bytes,
                             const uint8_t *g1_lagrange_bytes, uint64_t num_g1_lagrange_bytes,
                             const uint8_t *g2_monomial_bytes, uint64_t num_g2_monomial_bytes,
                             uint64_t precompute);                                             /* eip_4844.rs:180-222 */
C_KZG_RET load_trusted_setup_file(CKZGSettings *out, FILE *in);                                /* eip_4844.rs:227-269 */
void free_trusted_setup(CKZGSettings *s);                                                      /* eip_4844.rs:296-378 */
C_KZG_RET blob_to_kzg_commitment(KZGCommitment *out, const Blob *blob, const CKZGSettings *s); /* eip_4844.rs:163-175 */
C_KZG_RET compute_kzg_proof(KZGProof *proof_out, Bytes32 *y_out, const Blob *blob, const Bytes32 *z_bytes,
                            const CKZGSettings *s);                                            /* eip_4844.rs:476-496 */
C_KZG_RET compute_blob_kzg_proof(KZGProof *out, const Blob *blob, const Bytes48 *commitment_bytes,
                                 const CKZGSettings *s);                                       /* eip_4844.rs:274-291 */
/* Verification (eip_4844.rs:383-471).  The field work (challenge, evaluation) and, for batches, the G1 linear
 * combinations run on the GPU; the pairing check itself runs on the host, as in the reference (its GPU backend moves
 * only g1_lincomb; blst/src/kzg_proofs.rs:73-100 is CPU code).  *ok is the verdict; malformed input -> C_KZG_BADARGS. */
C_KZG_RET verify_kzg_proof(bool *ok, const Bytes48 *commitment_bytes, const Bytes32 *z_bytes, const Bytes32 *y_bytes,
                           const Bytes48 *proof_bytes, const CKZGSettings *s);                 /* eip_4844.rs:383-405 */
C_KZG_RET verify_blob_kzg_proof(bool *ok, const Blob *blob, const Bytes48 *commitment_bytes, const Bytes48 *proof_bytes,
                                const CKZGSettings *s);                                        /* eip_4844.rs:410-430 */
C_KZG_RET verify_blob_kzg_proof_batch(bool *ok, const Blob *blobs, const Bytes48 *commitments_bytes,
                                      const Bytes48 *proofs_bytes, size_t n, const CKZGSettings *s); /* eip_4844.rs:435-471 */
/* helpers the reference exports for the binding test-suites (eip_4844.rs:501-530) */
void compute_challenge(blst_fr *eval_challenge_out, const Blob *blob, const blst_p1 *commitment);
C_KZG_RET bytes_to_kzg_commitment(blst_p1 *out, const Bytes48 *b);
void bytes_from_bls_field(Bytes32 *out, const blst_fr *in);

/* Batched forms (new API; BASELINE.json configs[4] names a compute_blob_kzg_proof_batch the reference
 * does not have — its closest behaviour is a host loop, kzg-bench/src/benches/eip_4844.rs:56-60).
 * Per-blob results equal the single-blob calls; any invalid blob fails the whole call (BadArgs). */
C_KZG_RET kzgamd_blob_to_kzg_commitment_batch(KZGCommitment *out, const Blob *blobs, size_t n, const CKZGSettings *s);
C_KZG_RET kzgamd_compute_blob_kzg_proof_batch(KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n,
                                              const CKZGSettings *s);
/* Device-resident commit, enqueued on `stream` without synchronising: d_blobs = n x 131072 B,
 * d_out = n x 48 B, d_status = n x int32 (0 ok, 1 blob has an element >= r),
 * d_scratch = n x 131072 B of workspace. */
C_KZG_RET kzgamd_blob_to_kzg_commitment_device(void *d_out, void *d_status, void *d_scratch, const void *d_blobs,
                                               size_t n, const CKZGSettings *s, void *stream);
/* Device-resident compute_blob_kzg_proof (kzg/src/eip_4844.rs:541-584) for n blobs, enqueued on `stream` without
 * synchronising: d_blobs = n x 131072 B, d_commitments = n x 48 B, d_proofs = n x 48 B, d_status = n x int32
 * (non-zero: blob i has an element >= r or commitment i is not a valid G1 element — the reference fails the call;
 * here the other proofs of the batch stay valid), d_scratch = n x KZGAMD_PROOF_SCRATCH_BYTES of workspace.
 * The Fiat-Shamir SHA-256 runs on the device too (one lane per blob: ~9 ms of latency per batch and about a percent
 * of the chip — pipeline batches on a few streams); kzgamd_settings_reserve covers the MSM workspace. */
#define KZGAMD_PROOF_SCRATCH_BYTES (131072 + 64)
C_KZG_RET kzgamd_compute_blob_kzg_proof_device(void *d_proofs, void *d_status, void *d_scratch, const void *d_blobs,
                                               const void *d_commitments, size_t n, const CKZGSettings *s, void *stream);
/* EIP-7594: cells (128 x 2048 B) and cell proofs (128 x 48 B) of a blob, same name
 * and signature as the reference (kzg/src/eth/c_bindings.rs:356-372); either output may be NULL, not
 * both.  The first call that asks for proofs builds a second wide table over g1_values_monomial. */
/* compute_challenges_and_evaluate_polynomial (kzg/src/eip_4844.rs:690-719): the per-blob field work of
 * verify_blob_kzg_proof_batch (:736-832) — z_i = Fiat-Shamir challenge of (blob_i, commitment_i), y_i = p_i(z_i),
 * both 32-byte big-endian.  Commitments are validated (decode + subgroup) as validate_batched_input does.  The
 * pairing side (verify_kzg_proof_batch, :380-435) stays on the caller's CPU backend. */
C_KZG_RET kzgamd_compute_challenges_and_evaluate_batch(Bytes32 *zs_out, Bytes32 *ys_out, const Blob *blobs,
                                                       const Bytes48 *commitments, size_t n, const CKZGSettings *s);

/* verify_kzg_proof_batch (kzg/src/eip_4844.rs:380-435) up to the pairing: r = Fiat-Shamir scalar of the n tuples
 * (compute_r_powers, :328-378);  proof_lincomb = sum r^i proof_i;  rhs = sum r^i (C_i - [y_i]G) + sum r^i z_i proof_i.
 * Commitments and proofs are decoded and subgroup-checked on the GPU (validate_batched_input, :721-734), z_i / y_i must
 * be canonical (< r); any violation -> C_KZG_BADARGS.  The caller (its CPU backend) finishes with the one pairing check
 *     e(proof_lincomb, g2_values_monomial[1]) == e(rhs, G2_generator).
 * n == 0 yields two points at infinity.  kzgamd_verify_blob_kzg_proof_batch_g1 is verify_blob_kzg_proof_batch
 * (:736-832) up to the same point: it derives z_i, y_i from the blobs first. */
C_KZG_RET kzgamd_verify_kzg_proof_batch_g1(blst_p1 *proof_lincomb_out, blst_p1 *rhs_out, const Bytes48 *commitments,
                                           const Bytes32 *zs, const Bytes32 *ys, const Bytes48 *proofs, size_t n,
                                           const CKZGSettings *s);
C_KZG_RET kzgamd_verify_blob_kzg_proof_batch_g1(blst_p1 *proof_lincomb_out, blst_p1 *rhs_out, const Blob *blobs,
                                                const Bytes48 *commitments, const Bytes48 *proofs, size_t n,
                                                const CKZGSettings *s);

/* Host-only helpers over blst_p2 and the pairing (no GPU involved): pairings_verify = blst/src/kzg_proofs.rs:73-100
 * (1 if e(a1,a2) == e(b1,b2), 0 if not, -1 on NULL); uncompress = FsG2::from_bytes (0 ok, 1 invalid encoding /
 * not on the curve); mult takes a Montgomery blst_fr like G2Mul (blst/src/types/g2.rs:24-39). */
int kzgamd_pairings_verify(const blst_p1 *a1, const blst_p2 *a2, const blst_p1 *b1, const blst_p2 *b2);
int kzgamd_p2_uncompress(blst_p2 *out, const uint8_t in[96]);
void kzgamd_p2_compress(uint8_t out[96], const blst_p2 *in);
void kzgamd_p2_generator(blst_p2 *out);
void kzgamd_p2_mult(blst_p2 *out, const blst_p2 *in, const blst_fr *scalar);
void kzgamd_p2_add(blst_p2 *out, const blst_p2 *a, const blst_p2 *b);

typedef struct { uint8_t bytes[2048]; } Cell;
C_KZG_RET compute_cells_and_kzg_proofs(Cell *cells, KZGProof *proofs, const Blob *blob, const CKZGSettings *s);
C_KZG_RET kzgamd_compute_cells_and_kzg_proofs_batch(Cell *cells, KZGProof *proofs, const Blob *blobs, size_t n,
                                                    const CKZGSettings *s);
/* EIP-7594 recovery and cell verification (kzg/src/eth/c_bindings.rs:202-355 -> kzg/src/das.rs:101-243, 294-389).
 * recover: num_cells in [64, 128] cells with strictly ascending cell_indices < 128 -> all 128 cells and (unless
 * recovered_proofs is NULL) their 128 proofs; the five 8192-point transforms, coset shifts and inversions run on the GPU.
 * verify: ok = every (commitment, cell index, cell, proof) tuple is consistent; num_cells = 0 is true; decoding,
 * subgroup checks and the linear combinations (one two-row MSM) on the GPU, one pairing check on the host.
 * Every failure of the reference (bad encoding, element >= r, index >= 128, point outside G1, too few / unordered
 * cells) is C_KZG_BADARGS, as there. */
C_KZG_RET recover_cells_and_kzg_proofs(Cell *recovered_cells, KZGProof *recovered_proofs, const uint64_t *cell_indices,
                                       const Cell *cells, uint64_t num_cells, const CKZGSettings *s);
C_KZG_RET verify_cell_kzg_proof_batch(bool *ok, const Bytes48 *commitments_bytes, const uint64_t *cell_indices,
                                      const Cell *cells, const Bytes48 *proofs_bytes, uint64_t num_cells,
                                      const CKZGSettings *s);
/* blst/src/eip_7594.rs:35-97: the Fiat-Shamir scalar of a cell batch over DEDUPLICATED commitments (Montgomery blst_fr) */
C_KZG_RET compute_verify_cell_kzg_proof_batch_challenge(blst_fr *challenge_out, const Bytes48 *commitment_bytes,
                                                        uint64_t num_commitments, const uint64_t *commitment_indices,
                                                        const uint64_t *cell_indices, const Cell *cells,
                                                        const Bytes48 *proofs_bytes, uint64_t num_cells);
/* load_trusted_setup / load_trusted_setup_file with a configuration (device, table budget, tuning); same outputs and
 * failures, plus C_KZG_BADARGS for a malformed configuration */
C_KZG_RET kzgamd_load_trusted_setup_ex(CKZGSettings *out, const uint8_t *g1_monomial_bytes, uint64_t num_g1_monomial_bytes,
                                       const uint8_t *g1_lagrange_bytes, uint64_t num_g1_lagrange_bytes,
                                       const uint8_t *g2_monomial_bytes, uint64_t num_g2_monomial_bytes,
                                       uint64_t precompute, const KzgAmdConfig *cfg);
C_KZG_RET kzgamd_load_trusted_setup_file_ex(CKZGSettings *out, FILE *in, const KzgAmdConfig *cfg);
/* the prepared-MSM handle behind a settings object (for kzgamd_msm_* calls) */
void *kzgamd_settings_msm_handle(const CKZGSettings *s);
/* the GPU a settings object lives on (-1 if unknown) */
int kzgamd_settings_device(const CKZGSettings *s);
/* pre-allocates the per-stream workspace kzgamd_blob_to_kzg_commitment_device needs for batches of up to n blobs
 * on `stream` (see kzgamd_msm_reserve) */
C_KZG_RET kzgamd_settings_reserve(const CKZGSettings *s, size_t n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU from ONE process, inside the library (rust-kzg_amd/csrc/multi.hip).  The reference's callers are
 * one Rust process that parallelises inside itself over groups of blobs (kzg/src/eip_4844.rs:770-816) around one shared
 * precomputation handle (kzg/src/msm/sppark.rs:24-44); its GPU path is single-device
 * (arkworks3-sppark-wlc/sppark/msm/pippenger.cuh:573-575).  Here: s[0..ndev) are settings objects loaded from the
 * same trusted setup, normally one per GPU (two on one GPU work too — with a table budget that lets both fit — and are how the
 * path is tested on a one-GPU box).  Blob i of the batch goes to s[k] for the k with lo_k <= i < hi_k, contiguous slabs
 * whose sizes differ by at most one; one host thread per settings object drives that object's single-device pipeline;
 * results land in place in out[].  Nothing is exchanged between devices (no collective): blobs are independent and
 * every device has its own replica of the tables.  Per-blob results equal the single-device calls; any failing slab
 * fails the call (C_KZG_BADARGS, like the reference).  n < ndev uses the first n objects.
 * ------------------------------------------------------------------------------------------ */
/* the slab [*lo, *hi) of part k when n items are cut into `parts` contiguous slabs whose sizes differ by at most one
 * (what every *_multi entry point uses; with n < parts the first n parts get one item).  Returns 0, 1 on bad arguments.
 * No GPU involved. */
int kzgamd_shard_range(size_t n, size_t parts, size_t k, size_t *lo, size_t *hi);
/* one CKZGSettings per entry of devices[] (NULL = GPUs 0 .. ndev-1) from one setup file, loaded in parallel; all or
 * nothing: on failure every out[d] is left empty.  The caller's current device is left alone. */
C_KZG_RET kzgamd_load_trusted_setup_file_multi(CKZGSettings out[], const int devices[], size_t ndev, FILE *in);
/* Page-locks (hipHostRegister, portable across devices) / releases a caller's buffer: copies between it and any device
 * are then direct DMA instead of passing through the runtime's staging of pageable memory (one CPU pass over the data
 * per device less — with several GPUs fed from one process that staging shares the host's memory bandwidth).  Optional:
 * every entry point takes pageable buffers.  0 ok, 1 failed (nothing changed). */
int kzgamd_pin_host_buffer(void *p, size_t bytes);
int kzgamd_unpin_host_buffer(void *p);
/* the same with a configuration for every object (cfg->device is ignored: devices[] places them); two objects on one
 * GPU need an explicit table_budget_bytes that lets both fit */
C_KZG_RET kzgamd_load_trusted_setup_file_multi_ex(CKZGSettings out[], const int devices[], size_t ndev, FILE *in,
                                                  const KzgAmdConfig *cfg);
void kzgamd_free_trusted_setup_multi(CKZGSettings s[], size_t ndev);
C_KZG_RET kzgamd_blob_to_kzg_commitment_batch_multi(KZGCommitment *out, const Blob *blobs, size_t n,
                                                    const CKZGSettings *const s[], size_t ndev);
C_KZG_RET kzgamd_compute_blob_kzg_proof_batch_multi(KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n,
                                                    const CKZGSettings *const s[], size_t ndev);
C_KZG_RET kzgamd_compute_cells_and_kzg_proofs_batch_multi(Cell *cells, KZGProof *proofs, const Blob *blobs, size_t n,
                                                          const CKZGSettings *const s[], size_t ndev);
/* the reference verifies a large batch as groups and ANDs the verdicts (kzg/src/eip_4844.rs:770-816); the groups here
 * are the devices' slabs: one pairing check per device */
C_KZG_RET kzgamd_verify_blob_kzg_proof_batch_multi(bool *ok, const Blob *blobs, const Bytes48 *commitments,
                                                   const Bytes48 *proofs, size_t n, const CKZGSettings *const s[],
                                                   size_t ndev);
/* One large MSM sharded by index range: msm[d] was prepared (prepare_msm after kzgamd_set_device(d), or
 * kzgamd_msm_create_device) over points[offsets[d] .. offsets[d+1]); scalars is the whole array (offsets[ndev]
 * elements, Montgomery blst_fr as mult_pippenger_prepared).  Every device sums its slice, the ndev 144-byte partials are
 * added on the host (kzgamd_g1_sum).  An empty slice (offsets[d] == offsets[d+1]) may have a NULL handle. */
RustError kzgamd_mult_pippenger_prepared_multi(void *const msm[], size_t ndev, blst_p1 *out, const size_t offsets[],
                                               const blst_fr scalars[]);

/* Multi-GPU combine step of one large MSM sharded by index range over G ranks: every rank computes
 * its partial with mult_pippenger / the device entry points over its slice of (points, scalars); the G
 * 144-byte Jacobian partials are all-gathered and summed locally with this host-side helper (a G1 addition is
 * not a reduction operator RCCL offers).  The reference has no multi-GPU path; its single-GPU call is
 * blst/src/kzg_proofs.rs:47-61. */
void kzgamd_g1_sum(blst_p1 *out, const blst_p1 in[], size_t n);

/* library / device info */
int kzgamd_device_count(void);
/* Multi-GPU from one process (the reference's sppark path is single-GPU, blst/src/kzg_proofs.rs:47-61).
 * kzgamd_set_device selects the GPU for the calling thread (hipSetDevice): handles created afterwards — prepare_msm,
 * kzgamd_msm_create_device, kzgamd_ntt_new, load_trusted_setup(_file) — live on that GPU.  Every later call on a
 * handle, host-buffer or device-resident form, switches to the handle's GPU by itself and restores the caller's
 * current device on return; device pointers and streams passed to a *_device entry point must belong to the
 * handle's GPU.  kzgamd_set_device returns 0, or 1 if the runtime refuses the index; kzgamd_get_device returns the
 * calling thread's current device, -1 on error. */
int kzgamd_set_device(int device);
int kzgamd_get_device(void);
/* What a settings object got of the HBM-sized tables it asked for (they are built on first use and shrink, or are left
 * out, when HBM is short — e.g. a second CKZGSettings on the same GPU): for which = 0 the commitment / proof table
 * over g1_values_lagrange_brp, 1 the cell-proof table of single blobs over g1_values_monomial, 2 the FK20 table over
 * x_ext_fft_columns — window bits, rows and the kzgamd_msm_uses_wide_table() code of the handle (0 = no wide table:
 * the bucket engine / the direct form runs instead; results are the same, only slower).  Returns 0, 1 when that
 * table has not been built (yet), -1 on bad arguments. */
int kzgamd_settings_table_info(const CKZGSettings *s, int which, int *window_bits, int *rows, int *wide_table);
const char *kzgamd_version(void);

#ifdef __cplusplus
}
#endif
#endif
