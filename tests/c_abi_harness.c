/*
 * A native consumer of the C-ABI: plain C99 compiled against include/kzg_mi355x.h and linked with
 * libkzg_mi355x.so — what the reference does when it links its staticlib into the c-kzg-4844 bindings
 * (run-c-kzg-4844-tests.sh:36-57).  It replays test vectors through the header's own prototypes and struct layouts,
 * so the HEADER (not the ctypes mirror in rust-kzg_amd/__init__.py) is what the vectors pin.
 *
 *   c_abi_harness layout                      struct sizes / offsets only (no GPU needed)
 *   c_abi_harness run <setup.txt> <records>   replay a record file (needs a GPU); prints one line per op kind
 *
 * Record file (written by tests/test_c_abi_harness.py from the fixtures under tests/golden and the oracle): a sequence of
 *   u32 op, u32 nfields, then per field u64 length + bytes       (little-endian)
 * The last field of a record is the expectation; an EMPTY expectation means "the call must fail".
 * Test infrastructure; never part of the product.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kzg_mi355x.h"

enum {
    OP_COMMIT = 1,       /* blob | commitment */
    OP_PROOF = 2,        /* blob, z | proof ++ y */
    OP_BLOB_PROOF = 3,   /* blob, commitment | proof */
    OP_VERIFY = 4,       /* commitment, z, y, proof | verdict byte */
    OP_VERIFY_BLOB = 5,  /* blob, commitment, proof | verdict byte */
    OP_VERIFY_BATCH = 6, /* blobs, commitments, proofs | verdict byte */
    OP_CELLS = 7,        /* blob | sha256(cells) ++ sha256(proofs) ++ proof 0 ++ proof 127 (what the reference's vectors hold) */
    OP_RECOVER = 8,      /* indices (u64), cells | cells ++ sha256(proofs) */
    OP_VERIFY_CELLS = 9, /* commitments, indices (u64), cells, proofs | verdict byte */
    OP_CELL_CHALLENGE = 10, /* commitments, commitment indices, cell indices, cells, proofs | 32 bytes big-endian */
    OP_NTT = 11,         /* flag byte (inverse), blst_fr[] | blst_fr[] */
    OP_DAS = 12,         /* blst_fr[] evens | blst_fr[] odds */
    OP_MSM = 13,         /* blst_p1_affine[], blst_fr[] | 48-byte compressed sum */
    OP_FFT_G1 = 14,      /* flag byte, blst_p1[] | 48-byte compressed outputs */
    OP_CHALLENGE = 15,   /* blob, commitment | 32 bytes big-endian */
    OP_LOAD_BYTES = 16,  /* g1 monomial, g1 lagrange, g2 monomial, blob | commitment  (load_trusted_setup, byte form) */
    OP_COMMIT_BATCH = 17,/* blobs | commitments (batch, and batch_multi over two settings objects) */
    OP_PROOF_BATCH = 18, /* blobs, commitments | proofs */
    OP_G1_SUM = 19,      /* blst_p1[] | 48-byte compressed sum */
    OP_MATRIX = 20       /* blst_p1_affine[rows*cols], {rows, cols, nmat} (u64), blst_fr[nmat*rows*cols] | nmat*rows compressed sums */
};
#define MAX_OP 21
#define MAX_FIELDS 8

typedef struct {
    uint64_t len;
    unsigned char *p;
} field_t;

static int failures = 0;
static int count_ok[MAX_OP], count_fail[MAX_OP];

static void check(int op, int cond, const char *what, long rec) {
    if (cond) {
        count_ok[op]++;
    } else {
        count_fail[op]++;
        failures++;
        fprintf(stderr, "MISMATCH op %d record %ld: %s\n", op, rec, what);
    }
}

/* group equality through the C-ABI alone: e(a, G2) == e(b, G2) <=> a == b */
static int p1_equal(const blst_p1 *a, const blst_p1 *b) {
    blst_p2 g2;
    kzgamd_p2_generator(&g2);
    return kzgamd_pairings_verify(a, &g2, b, &g2) == 1;
}

static int p1_equals_compressed(const blst_p1 *a, const unsigned char *c48) {
    Bytes48 b;
    blst_p1 want;
    memcpy(b.bytes, c48, 48);
    if (bytes_to_kzg_commitment(&want, &b) != C_KZG_OK) return 0;
    return p1_equal(a, &want);
}

/* SHA-256 (FIPS 180-4), for the digests the reference's cell vectors are stored as */
static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256(unsigned char out[32], const unsigned char *msg, size_t len) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t total = ((len + 9 + 63) / 64) * 64, off;
    unsigned char *buf = (unsigned char *)calloc(total, 1);
    int i;
    if (!buf) exit(2);
    memcpy(buf, msg, len);
    buf[len] = 0x80;
    for (i = 0; i < 8; ++i) buf[total - 1 - i] = (unsigned char)(((uint64_t)len * 8) >> (8 * i));
    for (off = 0; off < total; off += 64) {
        uint32_t w[64], a[8], t1, t2;
        for (i = 0; i < 16; ++i)
            w[i] = ((uint32_t)buf[off + 4 * i] << 24) | ((uint32_t)buf[off + 4 * i + 1] << 16) | ((uint32_t)buf[off + 4 * i + 2] << 8) |
                   buf[off + 4 * i + 3];
        for (i = 16; i < 64; ++i)
            w[i] = w[i - 16] + (rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
                   (rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10));
        memcpy(a, h, sizeof a);
        for (i = 0; i < 64; ++i) {
            t1 = a[7] + (rotr(a[4], 6) ^ rotr(a[4], 11) ^ rotr(a[4], 25)) + ((a[4] & a[5]) ^ (~a[4] & a[6])) + K[i] + w[i];
            t2 = (rotr(a[0], 2) ^ rotr(a[0], 13) ^ rotr(a[0], 22)) + ((a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]));
            a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
        }
        for (i = 0; i < 8; ++i) h[i] += a[i];
    }
    for (i = 0; i < 8; ++i) {
        out[4 * i] = (unsigned char)(h[i] >> 24);
        out[4 * i + 1] = (unsigned char)(h[i] >> 16);
        out[4 * i + 2] = (unsigned char)(h[i] >> 8);
        out[4 * i + 3] = (unsigned char)h[i];
    }
    free(buf);
}

static int layout(void) {
    int bad = 0;
#define EXPECT(cond)                                          \
    do {                                                      \
        if (!(cond)) {                                        \
            fprintf(stderr, "layout: %s is false\n", #cond); \
            bad = 1;                                          \
        }                                                     \
    } while (0)
    /* kzg/src/eth/c_bindings.rs:16-113, 429-474 */
    EXPECT(sizeof(blst_fr) == 32);
    EXPECT(sizeof(blst_fp) == 48);
    EXPECT(sizeof(blst_p1_affine) == 96);
    EXPECT(sizeof(blst_p1) == 144);
    EXPECT(sizeof(blst_fp2) == 96);
    EXPECT(sizeof(blst_p2) == 288);
    EXPECT(sizeof(Bytes32) == 32 && sizeof(Bytes48) == 48);
    EXPECT(sizeof(Blob) == 131072 && sizeof(Cell) == 2048);
    EXPECT(sizeof(KZGCommitment) == 48 && sizeof(KZGProof) == 48);
    EXPECT(offsetof(CKZGSettings, roots_of_unity) == 0);
    EXPECT(offsetof(CKZGSettings, brp_roots_of_unity) == sizeof(void *));
    EXPECT(offsetof(CKZGSettings, reverse_roots_of_unity) == 2 * sizeof(void *));
    EXPECT(offsetof(CKZGSettings, g1_values_monomial) == 3 * sizeof(void *));
    EXPECT(offsetof(CKZGSettings, g1_values_lagrange_brp) == 4 * sizeof(void *));
    EXPECT(offsetof(CKZGSettings, g2_values_monomial) == 5 * sizeof(void *));
    EXPECT(offsetof(CKZGSettings, x_ext_fft_columns) == 6 * sizeof(void *));
    EXPECT(offsetof(CKZGSettings, tables) == 7 * sizeof(void *));
    EXPECT(offsetof(CKZGSettings, wbits) == 8 * sizeof(void *));
    EXPECT(offsetof(CKZGSettings, scratch_size) == 8 * sizeof(void *) + sizeof(size_t));
    EXPECT(sizeof(CKZGSettings) == 8 * sizeof(void *) + 2 * sizeof(size_t));
    EXPECT(C_KZG_OK == 0 && C_KZG_BADARGS == 1 && C_KZG_ERROR == 2 && C_KZG_MALLOC == 3);
    EXPECT(sizeof(RustError) == 2 * sizeof(void *) && offsetof(RustError, message) == sizeof(void *));
    EXPECT(sizeof(bool) == 1);
    EXPECT(KZGAMD_PROOF_SCRATCH_BYTES == 131072 + 64);
#undef EXPECT
    /* argument validation that needs no device */
    {
        CKZGSettings empty;
        KZGCommitment c;
        Blob *blob = (Blob *)calloc(1, sizeof(Blob));
        memset(&empty, 0, sizeof empty);
        if (!blob) return 2;
        if (blob_to_kzg_commitment(&c, blob, &empty) != C_KZG_BADARGS) bad = 1;
        if (blob_to_kzg_commitment(NULL, blob, &empty) != C_KZG_BADARGS) bad = 1;
        free_trusted_setup(&empty); /* never loaded: a no-op */
        free_trusted_setup(NULL);
        free(blob);
    }
    printf("layout %s, library %s\n", bad ? "MISMATCH" : "ok", kzgamd_version());
    return bad;
}

static int read_record(FILE *f, uint32_t *op, uint32_t *nf, field_t fields[MAX_FIELDS]) {
    uint32_t hdr[2];
    uint32_t i;
    if (fread(hdr, 4, 2, f) != 2) return 0;
    *op = hdr[0];
    *nf = hdr[1];
    if (*nf > MAX_FIELDS) {
        fprintf(stderr, "bad record\n");
        exit(2);
    }
    for (i = 0; i < *nf; ++i) {
        if (fread(&fields[i].len, 8, 1, f) != 1) exit(2);
        fields[i].p = (unsigned char *)malloc(fields[i].len ? fields[i].len : 1);
        if (!fields[i].p) exit(2);
        if (fields[i].len && fread(fields[i].p, 1, fields[i].len, f) != fields[i].len) exit(2);
    }
    return 1;
}

static int run(const char *setup_path, const char *records_path) {
    CKZGSettings s, s2;
    const CKZGSettings *both[2];
    FILE *f;
    void *ntt;
    uint32_t op, nf, i;
    field_t fl[MAX_FIELDS];
    long rec = 0;
    int have_s2 = 0;
    KzgAmdConfig cfg;

    if (kzgamd_device_count() < 1) {
        fprintf(stderr, "c_abi_harness: no GPU visible; the library has no CPU fallback\n");
        return 3;
    }
    /* two settings objects live side by side on the one GPU: 40 GB per table each, through the configuration */
    kzgamd_config_init(&cfg);
    cfg.table_budget_bytes = 40000000000ull;
    f = fopen(setup_path, "r");
    if (!f) return 2;
    if (kzgamd_load_trusted_setup_file_ex(&s, f, &cfg) != C_KZG_OK) {
        fprintf(stderr, "kzgamd_load_trusted_setup_file_ex failed\n");
        return 2;
    }
    fclose(f);
    {   /* a malformed configuration is refused: unknown tuning key, value out of range, wrong struct size */
        KzgAmdConfig bad = cfg;
        CKZGSettings sb;
        bad.tuning = "no_such_key=1";
        f = fopen(setup_path, "r");
        if (!f || kzgamd_load_trusted_setup_file_ex(&sb, f, &bad) != C_KZG_BADARGS || sb.g1_values_lagrange_brp != NULL) {
            fprintf(stderr, "an unknown tuning key was accepted\n");
            return 2;
        }
        fclose(f);
        bad.tuning = "spl=99";
        if (kzgamd_ntt_new_ex(4, &bad) != NULL) {
            fprintf(stderr, "an out-of-range tuning value was accepted\n");
            return 2;
        }
        bad = cfg;
        bad.struct_size = 3;
        if (kzgamd_ntt_new_ex(4, &bad) != NULL) {
            fprintf(stderr, "a truncated KzgAmdConfig was accepted\n");
            return 2;
        }
        if (strstr(kzgamd_tuning_keys(), "g1_wide_max 4096 ") == NULL) {
            fprintf(stderr, "kzgamd_tuning_keys does not list g1_wide_max\n");
            return 2;
        }
    }
    if (!s.roots_of_unity || !s.g1_values_lagrange_brp || !s.g2_values_monomial || !s.x_ext_fft_columns || s.tables) {
        fprintf(stderr, "CKZGSettings not populated as the reference's\n");
        return 2;
    }
    ntt = kzgamd_ntt_new(16);
    if (!ntt) return 2;
    f = fopen(records_path, "rb");
    if (!f) return 2;
    while (read_record(f, &op, &nf, fl)) {
        const field_t *ex = &fl[nf - 1];
        ++rec;
        switch (op) {
        case OP_COMMIT: {
            KZGCommitment c;
            C_KZG_RET rc = blob_to_kzg_commitment(&c, (const Blob *)fl[0].p, &s);
            if (ex->len == 0) check(op, rc == C_KZG_BADARGS, "expected BADARGS", rec);
            else check(op, rc == C_KZG_OK && memcmp(c.bytes, ex->p, 48) == 0, "commitment", rec);
            break;
        }
        case OP_PROOF: {
            KZGProof p;
            Bytes32 y;
            C_KZG_RET rc = compute_kzg_proof(&p, &y, (const Blob *)fl[0].p, (const Bytes32 *)fl[1].p, &s);
            if (ex->len == 0) check(op, rc == C_KZG_BADARGS, "expected BADARGS", rec);
            else check(op, rc == C_KZG_OK && memcmp(p.bytes, ex->p, 48) == 0 && memcmp(y.bytes, ex->p + 48, 32) == 0, "proof, y", rec);
            break;
        }
        case OP_BLOB_PROOF: {
            KZGProof p;
            C_KZG_RET rc = compute_blob_kzg_proof(&p, (const Blob *)fl[0].p, (const Bytes48 *)fl[1].p, &s);
            if (ex->len == 0) check(op, rc == C_KZG_BADARGS, "expected BADARGS", rec);
            else check(op, rc == C_KZG_OK && memcmp(p.bytes, ex->p, 48) == 0, "blob proof", rec);
            break;
        }
        case OP_VERIFY: {
            bool ok = false;
            C_KZG_RET rc = verify_kzg_proof(&ok, (const Bytes48 *)fl[0].p, (const Bytes32 *)fl[1].p, (const Bytes32 *)fl[2].p,
                                            (const Bytes48 *)fl[3].p, &s);
            if (ex->len == 0) check(op, rc == C_KZG_BADARGS, "expected BADARGS", rec);
            else check(op, rc == C_KZG_OK && ok == (ex->p[0] != 0), "verdict", rec);
            break;
        }
        case OP_VERIFY_BLOB: {
            bool ok = false;
            C_KZG_RET rc = verify_blob_kzg_proof(&ok, (const Blob *)fl[0].p, (const Bytes48 *)fl[1].p, (const Bytes48 *)fl[2].p, &s);
            if (ex->len == 0) check(op, rc == C_KZG_BADARGS, "expected BADARGS", rec);
            else check(op, rc == C_KZG_OK && ok == (ex->p[0] != 0), "verdict", rec);
            break;
        }
        case OP_VERIFY_BATCH: {
            bool ok = false;
            size_t n = (size_t)(fl[0].len / sizeof(Blob));
            C_KZG_RET rc = verify_blob_kzg_proof_batch(&ok, (const Blob *)fl[0].p, (const Bytes48 *)fl[1].p,
                                                       (const Bytes48 *)fl[2].p, n, &s);
            if (ex->len == 0) check(op, rc == C_KZG_BADARGS, "expected BADARGS", rec);
            else check(op, rc == C_KZG_OK && ok == (ex->p[0] != 0), "verdict", rec);
            break;
        }
        case OP_CELLS: {
            Cell *cells = (Cell *)malloc(128 * sizeof(Cell));
            KZGProof proofs[128];
            C_KZG_RET rc = compute_cells_and_kzg_proofs(cells, proofs, (const Blob *)fl[0].p, &s);
            if (ex->len == 0) check(op, rc == C_KZG_BADARGS, "expected BADARGS", rec);
            else {
                unsigned char dc[32], dp[32];
                sha256(dc, (const unsigned char *)cells, 128 * sizeof(Cell));
                sha256(dp, (const unsigned char *)proofs, 128 * 48);
                check(op, rc == C_KZG_OK && memcmp(dc, ex->p, 32) == 0 && memcmp(dp, ex->p + 32, 32) == 0 &&
                              memcmp(proofs[0].bytes, ex->p + 64, 48) == 0 && memcmp(proofs[127].bytes, ex->p + 112, 48) == 0,
                      "cells, proofs", rec);
            }
            free(cells);
            break;
        }
        case OP_RECOVER: {
            Cell *cells = (Cell *)malloc(128 * sizeof(Cell));
            KZGProof proofs[128];
            uint64_t n = fl[0].len / 8;
            C_KZG_RET rc = recover_cells_and_kzg_proofs(cells, proofs, (const uint64_t *)fl[0].p, (const Cell *)fl[1].p, n, &s);
            if (ex->len == 0) check(op, rc == C_KZG_BADARGS, "expected BADARGS", rec);
            else {
                unsigned char dp[32];
                sha256(dp, (const unsigned char *)proofs, 128 * 48);
                check(op, rc == C_KZG_OK && memcmp(cells, ex->p, 128 * sizeof(Cell)) == 0 &&
                              memcmp(dp, ex->p + 128 * sizeof(Cell), 32) == 0, "recovered cells, proofs", rec);
            }
            free(cells);
            break;
        }
        case OP_VERIFY_CELLS: {
            bool ok = false;
            uint64_t n = fl[1].len / 8;
            C_KZG_RET rc = verify_cell_kzg_proof_batch(&ok, (const Bytes48 *)fl[0].p, (const uint64_t *)fl[1].p,
                                                       (const Cell *)fl[2].p, (const Bytes48 *)fl[3].p, n, &s);
            if (ex->len == 0) check(op, rc == C_KZG_BADARGS, "expected BADARGS", rec);
            else check(op, rc == C_KZG_OK && ok == (ex->p[0] != 0), "verdict", rec);
            break;
        }
        case OP_CELL_CHALLENGE: {
            blst_fr ch;
            Bytes32 be;
            C_KZG_RET rc = compute_verify_cell_kzg_proof_batch_challenge(
                &ch, (const Bytes48 *)fl[0].p, fl[0].len / 48, (const uint64_t *)fl[1].p, (const uint64_t *)fl[2].p,
                (const Cell *)fl[3].p, (const Bytes48 *)fl[4].p, fl[2].len / 8);
            bytes_from_bls_field(&be, &ch);
            check(op, rc == C_KZG_OK && memcmp(be.bytes, ex->p, 32) == 0, "cell batch challenge", rec);
            break;
        }
        case OP_NTT: {
            size_t n = (size_t)(fl[1].len / sizeof(blst_fr));
            blst_fr *out = (blst_fr *)malloc(fl[1].len ? fl[1].len : 1);
            int rc = ntt_fr(ntt, out, (const blst_fr *)fl[1].p, n, fl[0].p[0]);
            if (ex->len == 0) check(op, rc > 0, "expected a reference error code", rec);
            else check(op, rc == 0 && memcmp(out, ex->p, fl[1].len) == 0, "ntt_fr", rec);
            free(out);
            break;
        }
        case OP_DAS: {
            size_t n = (size_t)(fl[0].len / sizeof(blst_fr));
            blst_fr *out = (blst_fr *)malloc(fl[0].len ? fl[0].len : 1);
            int rc = das_fft_extension(ntt, out, (const blst_fr *)fl[0].p, n);
            if (ex->len == 0) check(op, rc > 0, "expected a reference error code", rec);
            else check(op, rc == 0 && memcmp(out, ex->p, fl[0].len) == 0, "das_fft_extension", rec);
            free(out);
            break;
        }
        case OP_MSM: {
            /* the three sppark symbols (blst-sppark/src/lib.rs:8-62) + free_msm + the batched form */
            size_t n = (size_t)(fl[0].len / sizeof(blst_p1_affine));
            const blst_p1_affine *pts = (const blst_p1_affine *)fl[0].p;
            const blst_fr *sc = (const blst_fr *)fl[1].p;
            blst_p1 a, b, c2[2];
            RustError e1 = mult_pippenger(&a, pts, n, sc);
            void *h = prepare_msm(pts, n);
            RustError e2, e3;
            blst_fr *twice = (blst_fr *)malloc(2 * fl[1].len + 1);
            memcpy(twice, sc, fl[1].len);
            memcpy((unsigned char *)twice + fl[1].len, sc, fl[1].len);
            e2 = mult_pippenger_prepared(h, &b, n, sc);
            e3 = mult_pippenger_prepared_batch(h, c2, n, 2, twice);
            check(op, e1.code == 0 && e1.message == NULL && p1_equals_compressed(&a, ex->p), "mult_pippenger", rec);
            check(op, h != NULL && e2.code == 0 && p1_equals_compressed(&b, ex->p), "mult_pippenger_prepared", rec);
            check(op, e3.code == 0 && p1_equals_compressed(&c2[0], ex->p) && p1_equals_compressed(&c2[1], ex->p),
                  "mult_pippenger_prepared_batch", rec);
            {
                int wb = 0, rows = 0;
                size_t nb = 0, np = 0;
                check(op, kzgamd_msm_info(h, &wb, &rows, &nb, &np) == 0 && np == n && kzgamd_msm_device(h) == kzgamd_get_device(),
                      "kzgamd_msm_info", rec);
            }
            {   /* the sharded form over two handles on this GPU */
                size_t offs[3];
                void *hs[2];
                blst_p1 m;
                RustError e4;
                offs[0] = 0;
                offs[1] = n / 3;
                offs[2] = n;
                hs[0] = prepare_msm(pts, offs[1]);
                hs[1] = prepare_msm(pts + offs[1], n - offs[1]);
                e4 = kzgamd_mult_pippenger_prepared_multi(hs, 2, &m, offs, sc);
                check(op, e4.code == 0 && p1_equals_compressed(&m, ex->p), "kzgamd_mult_pippenger_prepared_multi", rec);
                free_msm(hs[0]);
                free_msm(hs[1]);
            }
            free_msm(h);
            free(twice);
            break;
        }
        case OP_FFT_G1: {
            size_t n = (size_t)(fl[1].len / sizeof(blst_p1)), k;
            blst_p1 *out = (blst_p1 *)malloc(fl[1].len ? fl[1].len : 1);
            int rc = fft_g1(ntt, out, (const blst_p1 *)fl[1].p, n, fl[0].p[0]);
            int good = rc == 0;
            for (k = 0; good && k < n; ++k) good = p1_equals_compressed(&out[k], ex->p + 48 * k);
            check(op, good, "fft_g1", rec);
            free(out);
            break;
        }
        case OP_CHALLENGE: {
            blst_p1 c;
            blst_fr z;
            Bytes32 be;
            C_KZG_RET rc = bytes_to_kzg_commitment(&c, (const Bytes48 *)fl[1].p);
            compute_challenge(&z, (const Blob *)fl[0].p, &c);
            bytes_from_bls_field(&be, &z);
            check(op, rc == C_KZG_OK && memcmp(be.bytes, ex->p, 32) == 0, "compute_challenge", rec);
            break;
        }
        case OP_LOAD_BYTES: {
            KZGCommitment c;
            C_KZG_RET rc = kzgamd_load_trusted_setup_ex(&s2, fl[0].p, fl[0].len, fl[1].p, fl[1].len, fl[2].p, fl[2].len, 0, &cfg);
            check(op, rc == C_KZG_OK, "kzgamd_load_trusted_setup_ex", rec);
            if (rc == C_KZG_OK) {
                int wb = 0, rows = 0, wide = 0;
                have_s2 = 1;
                check(op, blob_to_kzg_commitment(&c, (const Blob *)fl[3].p, &s2) == C_KZG_OK && memcmp(c.bytes, ex->p, 48) == 0,
                      "commitment on the byte-loaded settings", rec);
                check(op, kzgamd_settings_device(&s2) == kzgamd_get_device() && kzgamd_settings_msm_handle(&s2) != NULL &&
                              kzgamd_settings_table_info(&s2, 0, &wb, &rows, &wide) == 0 && rows > 0,
                      "settings introspection", rec);
            }
            break;
        }
        case OP_COMMIT_BATCH: {
            size_t n = (size_t)(fl[0].len / sizeof(Blob));
            KZGCommitment *out = (KZGCommitment *)calloc(n ? n : 1, 48);
            C_KZG_RET rc = kzgamd_blob_to_kzg_commitment_batch(out, (const Blob *)fl[0].p, n, &s);
            check(op, rc == C_KZG_OK && memcmp(out, ex->p, 48 * n) == 0, "kzgamd_blob_to_kzg_commitment_batch", rec);
            if (have_s2) {
                both[0] = &s;
                both[1] = &s2;
                memset(out, 0, 48 * n);
                rc = kzgamd_blob_to_kzg_commitment_batch_multi(out, (const Blob *)fl[0].p, n, both, 2);
                check(op, rc == C_KZG_OK && memcmp(out, ex->p, 48 * n) == 0, "kzgamd_blob_to_kzg_commitment_batch_multi", rec);
            }
            free(out);
            break;
        }
        case OP_PROOF_BATCH: {
            size_t n = (size_t)(fl[0].len / sizeof(Blob));
            KZGProof *out = (KZGProof *)calloc(n ? n : 1, 48);
            C_KZG_RET rc = kzgamd_compute_blob_kzg_proof_batch(out, (const Blob *)fl[0].p, (const Bytes48 *)fl[1].p, n, &s);
            check(op, rc == C_KZG_OK && memcmp(out, ex->p, 48 * n) == 0, "kzgamd_compute_blob_kzg_proof_batch", rec);
            if (have_s2) {
                bool ok = false;
                both[0] = &s;
                both[1] = &s2;
                memset(out, 0, 48 * n);
                rc = kzgamd_compute_blob_kzg_proof_batch_multi(out, (const Blob *)fl[0].p, (const Bytes48 *)fl[1].p, n, both, 2);
                check(op, rc == C_KZG_OK && memcmp(out, ex->p, 48 * n) == 0, "kzgamd_compute_blob_kzg_proof_batch_multi", rec);
                rc = kzgamd_verify_blob_kzg_proof_batch_multi(&ok, (const Blob *)fl[0].p, (const Bytes48 *)fl[1].p, out, n, both, 2);
                check(op, rc == C_KZG_OK && ok, "kzgamd_verify_blob_kzg_proof_batch_multi", rec);
            }
            free(out);
            break;
        }
        case OP_G1_SUM: {
            blst_p1 sum;
            kzgamd_g1_sum(&sum, (const blst_p1 *)fl[0].p, (size_t)(fl[0].len / sizeof(blst_p1)));
            check(op, p1_equals_compressed(&sum, ex->p), "kzgamd_g1_sum", rec);
            break;
        }
        case OP_MATRIX: {
            /* the table behind g1_lincomb_batch (kzg/src/lib.rs:156-181, kzg/src/msm/bgmw.rs:306-380) */
            const uint64_t *dims = (const uint64_t *)fl[1].p;
            const size_t rows = (size_t)dims[0], cols = (size_t)dims[1], nmat = (size_t)dims[2];
            KzgAmdConfig mc;
            void *h;
            blst_p1 *out = (blst_p1 *)malloc(nmat * rows * sizeof(blst_p1));
            RustError e;
            size_t k;
            int good;
            kzgamd_config_init(&mc);
            mc.table_budget_bytes = 4000000000ull;
            h = kzgamd_prepare_msm_matrix((const blst_p1_affine *)fl[0].p, rows, cols, &mc);
            check(op, h != NULL, "kzgamd_prepare_msm_matrix", rec);
            if (h) {
                e = kzgamd_mult_pippenger_matrix(h, out, (const blst_fr *)fl[2].p, nmat);
                good = e.code == 0;
                for (k = 0; good && k < nmat * rows; ++k) good = p1_equals_compressed(&out[k], ex->p + 48 * k);
                check(op, good, "kzgamd_mult_pippenger_matrix", rec);
                /* a plain prepared handle is not a matrix handle */
                e = kzgamd_mult_pippenger_matrix(kzgamd_settings_msm_handle(&s), out, (const blst_fr *)fl[2].p, 1);
                check(op, e.code != 0, "matrix call on a plain handle is refused", rec);
                free(e.message);
                free_msm(h);
            }
            free(out);
            break;
        }
        default:
            fprintf(stderr, "unknown op %u\n", op);
            return 2;
        }
        for (i = 0; i < nf; ++i) free(fl[i].p);
    }
    fclose(f);
    kzgamd_ntt_free(ntt);
    if (have_s2) free_trusted_setup(&s2);
    free_trusted_setup(&s);
    if (s.g1_values_lagrange_brp != NULL || s.roots_of_unity != NULL) {
        fprintf(stderr, "free_trusted_setup left pointers behind\n");
        failures++;
    }
    free_trusted_setup(&s); /* a second free is safe (c_bindings.rs:490-544) */
    for (i = 1; i < MAX_OP; ++i)
        if (count_ok[i] || count_fail[i]) printf("op %u: %d ok, %d failed\n", i, count_ok[i], count_fail[i]);
    printf("c_abi_harness: %ld records, %d failures\n", rec, failures);
    return failures ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "layout") == 0) return layout();
    if (argc >= 4 && strcmp(argv[1], "run") == 0) return run(argv[2], argv[3]);
    fprintf(stderr, "usage: %s layout | run <trusted_setup.txt> <records.bin>\n", argv[0]);
    return 2;
}
