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
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

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
 * B1 — GPU MSM plug-in.  Same three symbols `rust-kzg-blst` binds under feature `sppark`
 * (blst-sppark/src/lib.rs:8-62, defined today by blst-sppark/cuda/pippenger.cu:23-38).
 * Scalars are blst_fr IN MONTGOMERY FORM (blst/src/kzg_proofs.rs:47-48); out is Jacobian.
 * Thread-safe: calls on one handle serialise on an internal mutex.
 * ------------------------------------------------------------------------------------------ */
void *prepare_msm(const blst_p1_affine points[], size_t npoints);
RustError mult_pippenger_prepared(void *msm, blst_p1 *out, size_t npoints, const blst_fr scalars[]);
RustError mult_pippenger(blst_p1 *out, const blst_p1_affine points[], size_t npoints, const blst_fr scalars[]);

/* Additions (the reference leaks the handle; batched shape follows the wlc variant's
 * mult_pippenger_faster_inf(ctx,out,npoints,batches,scalars), arkworks3-sppark-wlc/src/lib.rs:24-42):
 * scalars is nbatch x npoints, out is nbatch points. */
void free_msm(void *msm);
RustError mult_pippenger_prepared_batch(void *msm, blst_p1 out[], size_t npoints, size_t nbatch,
                                        const blst_fr scalars[]);

/* Device-resident form used by the batched blob pipeline and bench.py: d_scalars / d_out are
 * device pointers, work is enqueued on `stream` (a hipStream_t, NULL = default stream) and NOT
 * synchronised.  scalars_mont != 0: blst_fr Montgomery limbs; 0: canonical little-endian 256-bit. */
RustError kzgamd_msm_prepared_batch_device(void *msm, void *d_out, const void *d_scalars, size_t npoints,
                                           size_t nbatch, int scalars_mont, void *stream);
/* introspection for benches/tests: window bits, table rows, buckets of a handle */
int kzgamd_msm_info(void *msm, int *window_bits, int *rows, size_t *nbuckets, size_t *npoints);

/* ------------------------------------------------------------------------------------------
 * B2 — NTT plug-in.  Replaces FFTFr::fft_fr / DASExtension::das_fft_extension for FsFFTSettings
 * (kzg/src/lib.rs:421-431; blst/src/fft_fr.rs:112-165; blst/src/data_availability_sampling.rs:78-100).
 * Natural order in, natural order out, Montgomery blst_fr; inverse scales by n^-1.
 * Return codes mirror the reference's error conditions:
 *   ntt_fr:            0 ok, 1 "longer than the available max width", 2 "power-of-two length expected"
 *   das_fft_extension: 0 ok, 1 empty, 2 not a power of two, 3 longer than max width / 2
 *   negative: device error
 * ------------------------------------------------------------------------------------------ */
void *kzgamd_ntt_new(unsigned scale);           /* FsFFTSettings::new(scale), blst/src/types/fft_settings.rs:30-58 */
void kzgamd_ntt_free(void *ctx);
int ntt_fr(void *ctx, blst_fr *out, const blst_fr *in, size_t n, int inverse);
int das_fft_extension(void *ctx, blst_fr *odds, const blst_fr *evens, size_t half_n);
/* device-resident, batched (nbatch independent transforms of length n, contiguous), on `stream` */
int kzgamd_ntt_fr_device(void *ctx, void *d_out, const void *d_in, size_t n, size_t nbatch, int inverse, void *stream);
/* host copies of the settings arrays (FFTSettings getters, kzg/src/lib.rs:465-481); counts in elements */
int kzgamd_ntt_roots(void *ctx, blst_fr *roots /*W+1*/, blst_fr *reverse_roots /*W+1*/, blst_fr *brp_roots /*W*/);

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

/* Opaque settings: device-resident fixed-base tables + roots (the reference keeps the GPU table
 * behind an opaque pointer too: kzg/src/msm/sppark.rs:5-22). */
typedef struct KzgAmdSettings KzgAmdSettings;

C_KZG_RET kzgamd_load_trusted_setup(KzgAmdSettings **out, const uint8_t *g1_monomial_bytes, size_t n1m,
                                    const uint8_t *g1_lagrange_bytes, size_t n1l,
                                    const uint8_t *g2_monomial_bytes, size_t n2);           /* eip_4844.rs:180-222 */
C_KZG_RET kzgamd_load_trusted_setup_file(KzgAmdSettings **out, FILE *in);                     /* eip_4844.rs:227-269 */
void kzgamd_free_trusted_setup(KzgAmdSettings *s);                                             /* eip_4844.rs:296-378 */
C_KZG_RET kzgamd_blob_to_kzg_commitment(KZGCommitment *out, const Blob *blob, const KzgAmdSettings *s); /* :163-175 */
C_KZG_RET kzgamd_compute_kzg_proof(KZGProof *proof_out, Bytes32 *y_out, const Blob *blob, const Bytes32 *z_bytes,
                                   const KzgAmdSettings *s);                                   /* :476-496 */
C_KZG_RET kzgamd_compute_blob_kzg_proof(KZGProof *out, const Blob *blob, const Bytes48 *commitment_bytes,
                                        const KzgAmdSettings *s);                              /* :274-291 */
C_KZG_RET kzgamd_compute_challenge(Bytes32 *out, const Blob *blob, const Bytes48 *commitment_bytes); /* :501-514 */
/* batched forms (new API, BASELINE.json configs[4]); per-blob results equal the single calls */
C_KZG_RET kzgamd_blob_to_kzg_commitment_batch(KZGCommitment *out, const Blob *blobs, size_t n, const KzgAmdSettings *s);
C_KZG_RET kzgamd_compute_blob_kzg_proof_batch(KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n,
                                              const KzgAmdSettings *s);
/* device-resident commit: d_blobs = n x 131072 bytes, d_out = n x 48 bytes, d_status = n x int32 (0 ok, 1 bad blob) */
C_KZG_RET kzgamd_blob_to_kzg_commitment_device(void *d_out, void *d_status, const void *d_blobs, size_t n,
                                               const KzgAmdSettings *s, void *stream);

/* library / device info */
int kzgamd_device_count(void);
const char *kzgamd_version(void);

#ifdef __cplusplus
}
#endif
#endif
