/*
 * TEST INFRASTRUCTURE (see oracle.h).  The EIP-4844 callers of the hot path,
 * restated from kzg/src/eip_4844.rs: trusted-setup text parser (:151-228),
 * blob_to_kzg_commitment (:256-314), compute_kzg_proof (:437-519),
 * compute_blob_kzg_proof (:541-563), fr_batch_inv (:882-914),
 * compute_challenge (:920-945), evaluate_polynomial_in_evaluation_form (:954-1003),
 * load_trusted_setup_rust (:1022-1086; the pairing-based Lagrange-form check
 * :1005-1020 is outside this path and not restated), and the polynomial half
 * of compute_cells (kzg/src/das.rs:244-292, :618-629).
 */
#include "oracle.h"
#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#define N O_FIELD_ELEMENTS_PER_BLOB

/* ---- trusted setup text: two decimal counts, then whitespace-separated hex bytes ---- */
static int scan_number(const char *s, size_t len, size_t *off, size_t *out) {
    while (*off < len && isspace((unsigned char)s[*off])) ++*off;
    if (*off >= len) return 0;
    size_t start = *off;
    while (*off < len && isdigit((unsigned char)s[*off])) ++*off;
    if (*off >= len) return 0; /* the reference requires a terminating non-digit */
    if (*off == start) return 0;
    size_t v = 0;
    for (size_t i = start; i < *off; ++i) {
        if (v > (SIZE_MAX - 9) / 10) return 0;
        v = v * 10 + (size_t)(s[i] - '0');
    }
    *out = v;
    return 1;
}

static int hexval(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

/* one byte = 1 or 2 hex digits (eip_4844.rs:188-213) */
static int scan_hex_byte(const char *s, size_t len, size_t *off, uint8_t *out) {
    while (*off < len && isspace((unsigned char)s[*off])) ++*off;
    if (*off >= len) return 0;
    int hi = hexval((unsigned char)s[*off]);
    if (hi < 0) return 0;
    if (*off + 1 < len && hexval((unsigned char)s[*off + 1]) >= 0) {
        *out = (uint8_t)(hi * 16 + hexval((unsigned char)s[*off + 1]));
        *off += 2;
    } else {
        *out = (uint8_t)hi;
        *off += 1;
    }
    return 1;
}

int oload_trusted_setup_text(osettings_t *s, const char *text, size_t len) {
    memset(s, 0, sizeof *s);
    size_t off = 0, n1 = 0, n2 = 0;
    if (!scan_number(text, len, &off, &n1) || n1 != N) return 1;
    if (!scan_number(text, len, &off, &n2) || n2 != 65) return 1;
    uint8_t *g1_lagrange = malloc(48 * N), *g1_monomial = malloc(48 * N);
    s->g2_monomial_bytes = malloc(96 * 65);
    int ok = 1;
    for (size_t i = 0; ok && i < 48 * N; ++i) ok = scan_hex_byte(text, len, &off, &g1_lagrange[i]);
    for (size_t i = 0; ok && i < 96 * 65; ++i) ok = scan_hex_byte(text, len, &off, &s->g2_monomial_bytes[i]);
    for (size_t i = 0; ok && i < 48 * N; ++i) ok = scan_hex_byte(text, len, &off, &g1_monomial[i]);
    s->g1_lagrange_brp = malloc(N * sizeof(og1_affine_t));
    s->g1_monomial = malloc(N * sizeof(og1_affine_t));
    for (size_t i = 0; ok && i < N; ++i) ok = og1_uncompress(&s->g1_monomial[i], g1_monomial + 48 * i);
    for (size_t i = 0; ok && i < N; ++i) ok = og1_uncompress(&s->g1_lagrange_brp[i], g1_lagrange + 48 * i);
    free(g1_lagrange);
    free(g1_monomial);
    if (ok) {
        oreverse_bit_order(s->g1_lagrange_brp, sizeof(og1_affine_t), N); /* eip_4844.rs:1070 */
        ok = offt_settings_new(&s->fs, 13) == 0;                         /* :1072-1077 */
    }
    if (!ok) {
        ofree_trusted_setup(s);
        return 1;
    }
    return 0;
}

void ofree_trusted_setup(osettings_t *s) {
    if (s->bgmw) {
        obgmw_table_free(s->bgmw);
        free(s->bgmw);
    }
    free(s->g1_lagrange_brp);
    free(s->g1_monomial);
    free(s->g2_monomial_bytes);
    if (s->fs.roots_of_unity) offt_settings_free(&s->fs);
    memset(s, 0, sizeof *s);
}

/* bytes_to_blob, eip_4844.rs:867-880 */
int oblob_to_fr(ofr_t *out, const uint8_t *blob) {
    for (size_t i = 0; i < N; ++i)
        if (!ofr_from_be32(&out[i], blob + 32 * i)) return 1;
    return 0;
}

static void lincomb_setup(og1_t *out, const ofr_t *scalars, const osettings_t *s) {
    omsm_affine(out, s->g1_lagrange_brp, scalars, N);
}

int oblob_to_kzg_commitment(uint8_t out[48], const uint8_t *blob, const osettings_t *s) {
    ofr_t *poly = malloc(N * sizeof *poly);
    if (oblob_to_fr(poly, blob)) {
        free(poly);
        return 1;
    }
    og1_t c;
    lincomb_setup(&c, poly, s);
    og1_compress(out, &c);
    free(poly);
    return 0;
}

int oblob_to_kzg_commitment_bgmw(uint8_t out[48], const uint8_t *blob, osettings_t *s) {
    if (!s->bgmw) {
        obgmw_table_t *t = calloc(1, sizeof *t);
        if (!t || obgmw_table_new(t, s->g1_lagrange_brp, N)) return 1;
        s->bgmw = t;
    }
    ofr_t *poly = malloc(N * sizeof *poly);
    if (oblob_to_fr(poly, blob)) {
        free(poly);
        return 1;
    }
    uint8_t *le = malloc(32 * N);
    for (size_t i = 0; i < N; ++i) ofr_to_scalar_le(le + 32 * i, &poly[i]);
    og1_t c;
    obgmw_multiply(&c, s->bgmw, le, N);
    og1_compress(out, &c);
    free(le);
    free(poly);
    return 0;
}

/* fr_batch_inv, eip_4844.rs:882-914 */
static int fr_batch_inv(ofr_t *out, const ofr_t *a, size_t len) {
    ofr_t acc;
    ofr_one(&acc);
    for (size_t i = 0; i < len; ++i) {
        out[i] = acc;
        ofr_mul(&acc, &acc, &a[i]);
    }
    if (ofr_is_zero(&acc)) return 1;
    ofr_inv(&acc, &acc);
    for (size_t i = len; i-- > 0;) {
        ofr_mul(&out[i], &out[i], &acc);
        ofr_mul(&acc, &acc, &a[i]);
    }
    return 0;
}

/* compute_challenge_rust, eip_4844.rs:920-945 */
void ocompute_challenge(ofr_t *out, const ofr_t *blob_fr, const uint8_t commitment[48]) {
    size_t sz = 16 + 16 + O_BYTES_PER_BLOB + 48;
    uint8_t *buf = calloc(sz, 1), h[32];
    memcpy(buf, "FSBLOBVERIFY_V1_", 16);
    uint64_t n = N;
    for (int i = 0; i < 8; ++i) buf[24 + 7 - i] = (uint8_t)(n >> (8 * i));
    for (size_t i = 0; i < N; ++i) ofr_to_be32(buf + 32 + 32 * i, &blob_fr[i]);
    memcpy(buf + 32 + O_BYTES_PER_BLOB, commitment, 48);
    osha256(h, buf, sz);
    ofr_from_be32_unchecked(out, h);
    free(buf);
}

/* eip_4844.rs:954-1003 */
int oevaluate_polynomial_in_evaluation_form(ofr_t *out, const ofr_t *p, const ofr_t *x, const osettings_t *s) {
    const ofr_t *roots = s->fs.brp_roots_of_unity;
    ofr_t *inv_in = malloc(N * sizeof(ofr_t)), *inv = malloc(N * sizeof(ofr_t));
    for (size_t i = 0; i < N; ++i) {
        if (ofr_eq(x, &roots[i])) {
            *out = p[i];
            free(inv_in);
            free(inv);
            return 0;
        }
        ofr_sub(&inv_in[i], x, &roots[i]);
    }
    if (fr_batch_inv(inv, inv_in, N)) {
        free(inv_in);
        free(inv);
        return 1;
    }
    ofr_t acc, tmp;
    ofr_zero(&acc);
    for (size_t i = 0; i < N; ++i) {
        ofr_mul(&tmp, &inv[i], &roots[i]);
        ofr_mul(&tmp, &tmp, &p[i]);
        ofr_add(&acc, &acc, &tmp);
    }
    ofr_from_u64(&tmp, N);
    ofr_inv(&tmp, &tmp);
    ofr_mul(&acc, &acc, &tmp); /* out.div(N) */
    ofr_pow(&tmp, x, N);
    ofr_t one;
    ofr_one(&one);
    ofr_sub(&tmp, &tmp, &one);
    ofr_mul(out, &acc, &tmp);
    free(inv_in);
    free(inv);
    return 0;
}

/* compute_kzg_proof_rust, eip_4844.rs:437-519 */
static int compute_kzg_proof_fr(og1_t *proof, ofr_t *y_out, const ofr_t *poly, const ofr_t *z, const osettings_t *s) {
    const ofr_t *roots = s->fs.brp_roots_of_unity;
    ofr_t y;
    if (oevaluate_polynomial_in_evaluation_form(&y, poly, z, s)) return 1;
    ofr_t *q = calloc(N, sizeof(ofr_t)), *inv_in = calloc(N, sizeof(ofr_t)), *inv = calloc(N, sizeof(ofr_t));
    size_t m = 0;
    for (size_t i = 0; i < N; ++i) {
        if (ofr_eq(z, &roots[i])) {
            m = i + 1;
            ofr_one(&inv_in[i]);
            continue;
        }
        ofr_sub(&q[i], &poly[i], &y);
        ofr_sub(&inv_in[i], &roots[i], z);
    }
    int rc = fr_batch_inv(inv, inv_in, N);
    if (!rc) {
        for (size_t i = 0; i < N; ++i) ofr_mul(&q[i], &q[i], &inv[i]);
        if (m != 0) {
            m -= 1;
            ofr_zero(&q[m]);
            for (size_t i = 0; i < N; ++i) {
                if (i == m) continue;
                ofr_t tmp;
                ofr_sub(&tmp, z, &roots[i]);
                ofr_mul(&inv_in[i], &tmp, z);
            }
            rc = fr_batch_inv(inv, inv_in, N);
            for (size_t i = 0; !rc && i < N; ++i) {
                if (i == m) continue;
                ofr_t tmp;
                ofr_sub(&tmp, &poly[i], &y);
                ofr_mul(&tmp, &tmp, &roots[i]);
                ofr_mul(&tmp, &tmp, &inv[i]);
                ofr_add(&q[m], &q[m], &tmp);
            }
        }
    }
    if (!rc) {
        lincomb_setup(proof, q, s);
        *y_out = y;
    }
    free(q);
    free(inv_in);
    free(inv);
    return rc;
}

int ocompute_kzg_proof(uint8_t proof[48], uint8_t y[32], const uint8_t *blob, const uint8_t z[32], const osettings_t *s) {
    ofr_t *poly = malloc(N * sizeof *poly), zf, yf;
    og1_t pr;
    int rc = oblob_to_fr(poly, blob);
    if (!rc) rc = !ofr_from_be32(&zf, z);
    if (!rc) rc = compute_kzg_proof_fr(&pr, &yf, poly, &zf, s);
    if (!rc) {
        og1_compress(proof, &pr);
        ofr_to_be32(y, &yf);
    }
    free(poly);
    return rc;
}

/* compute_blob_kzg_proof_raw/_rust, eip_4844.rs:541-584 */
int ocompute_blob_kzg_proof(uint8_t proof[48], const uint8_t *blob, const uint8_t commitment[48], const osettings_t *s) {
    ofr_t *poly = malloc(N * sizeof *poly), zf, yf;
    og1_t pr, c;
    og1_affine_t ca;
    int rc = oblob_to_fr(poly, blob);
    if (!rc) rc = !og1_uncompress(&ca, commitment);
    if (!rc) {
        og1_from_affine(&c, &ca);
        if (!og1_is_inf(&c) && !og1_in_subgroup(&c)) rc = 1;
    }
    if (!rc) {
        uint8_t cbytes[48];
        og1_compress(cbytes, &c); /* commitment.to_bytes() */
        ocompute_challenge(&zf, poly, cbytes);
        rc = compute_kzg_proof_fr(&pr, &yf, poly, &zf, s);
    }
    if (!rc) og1_compress(proof, &pr);
    free(poly);
    return rc;
}

/* cells of compute_cells_and_kzg_proofs (kzg/src/das.rs:258-279): monomial form by
 * inverse NTT of the bit-reversed blob, zero-extend to 8192, forward NTT, bit-reverse */
int ocompute_cells(uint8_t *cells_out, const uint8_t *blob, const osettings_t *s) {
    ofr_t *poly = malloc(N * sizeof(ofr_t)), *mono = calloc(2 * N, sizeof(ofr_t)), *ext = malloc(2 * N * sizeof(ofr_t));
    int rc = oblob_to_fr(poly, blob);
    if (!rc) {
        oreverse_bit_order(poly, sizeof(ofr_t), N);
        rc = offt_fr(&s->fs, mono, poly, N, 1);
    }
    if (!rc) rc = offt_fr(&s->fs, ext, mono, 2 * N, 0);
    if (!rc) {
        oreverse_bit_order(ext, sizeof(ofr_t), 2 * N);
        for (size_t i = 0; i < 2 * N; ++i) ofr_to_be32(cells_out + 32 * i, &ext[i]);
    }
    free(poly);
    free(mono);
    free(ext);
    return rc;
}

/* KZG multiproof of cell k by its definition (the value the reference's FK20 path,
 * kzg/src/das.rs:660-696, arrives at): proof_k = [q_k(tau)] with
 * q_k(X) = p(X) div (X^64 - a_k), a_k = h_k^64, h_k = w_8192^brp7(k) the coset shift of cell k.
 * Coefficients by the division recurrence q_j = p_{j+64} + a_k q_{j+64}; commitment by MSM over the
 * monomial setup.  One 4096-point MSM per proof, so tests ask for a few k only. */
static size_t brp_bits(size_t v, unsigned bits) {
    size_t r = 0;
    for (unsigned b = 0; b < bits; ++b)
        if (v & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
    return r;
}

int ocompute_cell_proof(uint8_t proof[48], const uint8_t *blob, size_t k, const osettings_t *s) {
    if (k >= 128) return 1;
    ofr_t *poly = malloc(N * sizeof(ofr_t)), *mono = malloc(N * sizeof(ofr_t)), *q = calloc(N, sizeof(ofr_t));
    int rc = oblob_to_fr(poly, blob);
    if (!rc) {
        oreverse_bit_order(poly, sizeof(ofr_t), N);
        rc = offt_fr(&s->fs, mono, poly, N, 1);
    }
    if (!rc) {
        const ofr_t *a = &s->fs.roots_of_unity[64 * brp_bits(k, 7)];
        for (size_t j = N - 64; j-- > 0;) {
            ofr_t t;
            if (j + 64 < N - 64) ofr_mul(&t, a, &q[j + 64]);
            else ofr_zero(&t);
            ofr_add(&q[j], &mono[j + 64], &t);
        }
        og1_t pr;
        omsm_affine(&pr, s->g1_monomial, q, N);
        og1_compress(proof, &pr);
    }
    free(poly);
    free(mono);
    free(q);
    return rc;
}

/* ---- batched verification, G1 half ----
 * compute_r_powers (kzg/src/eip_4844.rs:328-378) and the three linear combinations of
 * verify_kzg_proof_batch (:380-435), up to the pairing:  returns
 *     proof_lincomb = sum r^i * proof_i
 *     rhs           = sum r^i * (C_i - [y_i]G)  +  sum (r^i z_i) * proof_i
 * i.e. the two G1 inputs of the final  e(proof_lincomb, [tau]G2) == e(rhs, G2)  check.  Inputs are validated the way
 * verify_blob_kzg_proof_batch's callers do (bytes_to_* + validate_batched_input, :721-734): 1 = BadArgs. */
int ocompute_r_powers(ofr_t *out, const uint8_t *commitments, const uint8_t *zs, const uint8_t *ys, const uint8_t *proofs, size_t n) {
    size_t sz = 32 + n * (48 + 32 + 32 + 48), off = 32;
    uint8_t *buf = calloc(sz, 1), h[32];
    memcpy(buf, "RCKZGBATCH___V1_", 16);
    uint64_t fe = N, nn = n;
    for (int i = 0; i < 8; ++i) {
        buf[16 + 7 - i] = (uint8_t)(fe >> (8 * i));
        buf[24 + 7 - i] = (uint8_t)(nn >> (8 * i));
    }
    for (size_t i = 0; i < n; ++i) {
        memcpy(buf + off, commitments + 48 * i, 48); /* G1::to_bytes of a decoded valid point = its input bytes */
        off += 48;
        memcpy(buf + off, zs + 32 * i, 32);
        off += 32;
        memcpy(buf + off, ys + 32 * i, 32);
        off += 32;
        memcpy(buf + off, proofs + 48 * i, 48);
        off += 48;
    }
    osha256(h, buf, sz);
    free(buf);
    ofr_t r;
    ofr_from_be32_unchecked(&r, h);
    /* compute_powers (:309-326) */
    if (n > 0) ofr_one(&out[0]);
    for (size_t i = 1; i < n; ++i) ofr_mul(&out[i], &out[i - 1], &r);
    return 0;
}

int overify_kzg_proof_batch_g1(og1_t *proof_lincomb, og1_t *rhs, const uint8_t *commitments, const uint8_t *zs,
                               const uint8_t *ys, const uint8_t *proofs, size_t n) {
    og1_t *c = malloc((n + 1) * sizeof(og1_t)), *p = malloc((n + 1) * sizeof(og1_t)), *cmy = malloc((n + 1) * sizeof(og1_t));
    ofr_t *z = malloc((n + 1) * sizeof(ofr_t)), *y = malloc((n + 1) * sizeof(ofr_t)), *rp = malloc((n + 1) * sizeof(ofr_t)),
          *rz = malloc((n + 1) * sizeof(ofr_t));
    int rc = 0;
    for (size_t i = 0; !rc && i < n; ++i) {
        og1_affine_t a;
        if (!og1_uncompress(&a, commitments + 48 * i)) rc = 1;
        else {
            og1_from_affine(&c[i], &a);
            if (!og1_is_inf(&c[i]) && !og1_in_subgroup(&c[i])) rc = 1;
        }
        if (!rc && !og1_uncompress(&a, proofs + 48 * i)) rc = 1;
        if (!rc) {
            og1_from_affine(&p[i], &a);
            if (!og1_is_inf(&p[i]) && !og1_in_subgroup(&p[i])) rc = 1;
        }
        if (!rc && (!ofr_from_be32(&z[i], zs + 32 * i) || !ofr_from_be32(&y[i], ys + 32 * i))) rc = 1;
    }
    if (!rc) {
        ocompute_r_powers(rp, commitments, zs, ys, proofs, n);
        og1_lincomb(proof_lincomb, p, rp, n);
        og1_t g, t;
        og1_generator(&g);
        for (size_t i = 0; i < n; ++i) {
            og1_mul(&t, &g, &y[i]);          /* [y_i] */
            og1_neg(&t, &t);
            og1_add_or_dbl(&cmy[i], &c[i], &t); /* C_i - [y_i] */
            ofr_mul(&rz[i], &rp[i], &z[i]);
        }
        og1_t a, b;
        og1_lincomb(&a, p, rz, n);
        og1_lincomb(&b, cmy, rp, n);
        og1_add_or_dbl(rhs, &b, &a);
    }
    free(c);
    free(p);
    free(cmy);
    free(z);
    free(y);
    free(rp);
    free(rz);
    return rc;
}
