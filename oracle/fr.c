/*
 * TEST INFRASTRUCTURE (see oracle.h).  Fr = GF(r), r the 255-bit BLS12-381 group
 * order; 4 x u64 Montgomery limbs, R = 2^256 — the representation behind the
 * reference's FsFr (blst/src/types/fr.rs:18-278, FFI into blst_fr_*).
 * Constants as stated in-tree at zkcrypto/bls12_381/src/scalar.rs:75-173.
 */
#include "oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

static const uint64_t RM[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull,
                               0x73eda753299d7d48ull};
static const uint64_t R_INV = 0xfffffffeffffffffull; /* -r^-1 mod 2^64, scalar.rs:156-157 */
static const uint64_t ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull,
                                0x1824b159acc5056full};
static const uint64_t R2[4] = {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full,
                               0x0748d9d99f59ff11ull};
static const uint64_t R_MINUS_2[4] = {0xfffffffeffffffffull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull,
                                      0x73eda753299d7d48ull};

static inline int geq_r(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > RM[i]) return 1;
        if (a[i] < RM[i]) return 0;
    }
    return 1;
}

static inline void sub_r(uint64_t a[4]) {
    u128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - RM[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}

void ofr_add(ofr_t *r, const ofr_t *a, const ofr_t *b) {
    u128 c = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq_r(t)) sub_r(t);
    memcpy(r->l, t, sizeof t);
}

void ofr_sub(ofr_t *r, const ofr_t *a, const ofr_t *b) {
    u128 borrow = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        t[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)t[i] + RM[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, sizeof t);
}

int ofr_is_zero(const ofr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
int ofr_is_one(const ofr_t *a) { return memcmp(a->l, ONE, sizeof ONE) == 0; }
int ofr_eq(const ofr_t *a, const ofr_t *b) { return memcmp(a->l, b->l, sizeof a->l) == 0; }
void ofr_zero(ofr_t *r) { memset(r, 0, sizeof *r); }
void ofr_one(ofr_t *r) { memcpy(r->l, ONE, sizeof ONE); }

void ofr_neg(ofr_t *r, const ofr_t *a) {
    ofr_t z;
    ofr_zero(&z);
    ofr_sub(r, &z, a);
}

void ofr_mul(ofr_t *r, const ofr_t *a, const ofr_t *b) {
    uint64_t t[6] = {0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * R_INV;
        c = (u128)m * RM[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * RM[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || geq_r(t)) sub_r(t);
    memcpy(r->l, t, 4 * sizeof(uint64_t));
}

void ofr_sqr(ofr_t *r, const ofr_t *a) { ofr_mul(r, a, a); }

static void fr_pow_limbs(ofr_t *r, const ofr_t *a, const uint64_t *e, int n) {
    ofr_t acc, base = *a;
    ofr_one(&acc);
    for (int i = 0; i < n; ++i)
        for (int b = 0; b < 64; ++b) {
            if ((e[i] >> b) & 1) ofr_mul(&acc, &acc, &base);
            ofr_sqr(&base, &base);
        }
    *r = acc;
}

/* FsFr::inverse / eucl_inverse (fr.rs:218-238): same value, a^(r-2); 0 -> 0 */
void ofr_inv(ofr_t *r, const ofr_t *a) { fr_pow_limbs(r, a, R_MINUS_2, 4); }

/* FsFr::pow (fr.rs:240-257) */
void ofr_pow(ofr_t *r, const ofr_t *a, uint64_t e) { fr_pow_limbs(r, a, &e, 1); }

/* canonical limbs -> Montgomery: FsFr::from_u64_arr (fr.rs:114-121) */
void ofr_from_u64_arr(ofr_t *r, const uint64_t v[4]) {
    ofr_t raw, r2;
    memcpy(raw.l, v, sizeof raw.l);
    memcpy(r2.l, R2, sizeof R2);
    ofr_mul(r, &raw, &r2);
}

void ofr_from_u64(ofr_t *r, uint64_t v) {
    uint64_t a[4] = {v, 0, 0, 0};
    ofr_from_u64_arr(r, a);
}

/* Montgomery -> canonical limbs: FsFr::to_u64_arr (fr.rs:138-145) */
void ofr_to_u64_arr(uint64_t v[4], const ofr_t *a) {
    ofr_t one_raw = {{1, 0, 0, 0}}, t;
    ofr_mul(&t, a, &one_raw);
    memcpy(v, t.l, sizeof t.l);
}

static void be32_to_limbs(uint64_t raw[4], const uint8_t in[32]) {
    for (int i = 0; i < 4; ++i) {
        uint64_t w = 0;
        for (int j = 0; j < 8; ++j) w = (w << 8) | in[(3 - i) * 8 + j];
        raw[i] = w;
    }
}

/* FsFr::from_bytes (fr.rs:64-86): big-endian, values >= r rejected */
int ofr_from_be32(ofr_t *r, const uint8_t in[32]) {
    uint64_t raw[4];
    be32_to_limbs(raw, in);
    if (geq_r(raw)) return 0;
    ofr_from_u64_arr(r, raw);
    return 1;
}

/* FsFr::from_bytes_unchecked (fr.rs:88-107): any 256-bit value, reduced mod r.
 * Montgomery-multiplying the raw value by R^2 is a reduction: the CIOS bound
 * only needs one operand < r. */
void ofr_from_be32_unchecked(ofr_t *r, const uint8_t in[32]) {
    uint64_t raw[4];
    be32_to_limbs(raw, in);
    ofr_t a, r2;
    memcpy(a.l, raw, sizeof raw);
    memcpy(r2.l, R2, sizeof R2);
    ofr_mul(r, &r2, &a);
}

void ofr_to_be32(uint8_t out[32], const ofr_t *a) {
    uint64_t raw[4];
    ofr_to_u64_arr(raw, a);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) out[(3 - i) * 8 + j] = (uint8_t)(raw[i] >> (56 - 8 * j));
}

/* FsFr::to_scalar (fr.rs:271-277): canonical little-endian bytes */
void ofr_to_scalar_le(uint8_t out[32], const ofr_t *a) {
    uint64_t raw[4];
    ofr_to_u64_arr(raw, a);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) out[i * 8 + j] = (uint8_t)(raw[i] >> (8 * j));
}
