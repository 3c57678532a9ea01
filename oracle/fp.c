/*
 * TEST INFRASTRUCTURE (see oracle.h).  Fp = GF(p), p the 381-bit BLS12-381 base
 * prime; 6 x u64 Montgomery limbs, R = 2^384 — the representation behind the
 * reference's FsFp (blst/src/types/fp.rs:5-97, FFI into blst 0.3.16's
 * blst_fp_{add,sub,mul,sqr,cneg,inverse}).  Constants as stated in-tree at
 * zkcrypto/bls12_381/src/fp.rs:70-104 (MODULUS, INV, R, R2).
 */
#include "oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

static const uint64_t P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                              0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t P_INV = 0x89f3fffcfffcfffdull; /* -p^-1 mod 2^64, fp.rs:80-81 */
static const uint64_t ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                                0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
static const uint64_t R2[6] = {0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull,
                               0x67eb88a9939d83c0ull, 0x9a793e85b519952dull, 0x11988fe592cae3aaull};
/* (p-1)/2, the "lexicographically largest" threshold, fp.rs:285-290 */
static const uint64_t P_HALF[6] = {0xdcff7fffffffd555ull, 0x0f55ffff58a9ffffull, 0xb39869507b587b12ull,
                                   0xb23ba5c279c2895full, 0x258dd3db21a5d66bull, 0x0d0088f51cbff34dull};
/* (p+1)/4, fp.rs:331-338 */
static const uint64_t P_SQRT_EXP[6] = {0xee7fbfffffffeaabull, 0x07aaffffac54ffffull, 0xd9cc34a83dac3d89ull,
                                       0xd91dd2e13ce144afull, 0x92c6e9ed90d2eb35ull, 0x0680447a8e5ff9a6ull};
/* p-2, fp.rs:347-359 */
static const uint64_t P_MINUS_2[6] = {0xb9feffffffffaaa9ull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                      0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};

static inline int geq_p(const uint64_t a[6]) {
    for (int i = 5; i >= 0; --i) {
        if (a[i] > P[i]) return 1;
        if (a[i] < P[i]) return 0;
    }
    return 1;
}

static inline void sub_p(uint64_t a[6]) {
    u128 borrow = 0;
    for (int i = 0; i < 6; ++i) {
        u128 d = (u128)a[i] - P[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}

void ofp_add(ofp_t *r, const ofp_t *a, const ofp_t *b) {
    u128 c = 0;
    uint64_t t[6];
    for (int i = 0; i < 6; ++i) {
        c += (u128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq_p(t)) sub_p(t);
    memcpy(r->l, t, sizeof t);
}

void ofp_sub(ofp_t *r, const ofp_t *a, const ofp_t *b) {
    u128 borrow = 0;
    uint64_t t[6];
    for (int i = 0; i < 6; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        t[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 6; ++i) {
            c += (u128)t[i] + P[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, sizeof t);
}

int ofp_is_zero(const ofp_t *a) {
    uint64_t acc = 0;
    for (int i = 0; i < 6; ++i) acc |= a->l[i];
    return acc == 0;
}

int ofp_eq(const ofp_t *a, const ofp_t *b) { return memcmp(a->l, b->l, sizeof a->l) == 0; }

void ofp_neg(ofp_t *r, const ofp_t *a) {
    if (ofp_is_zero(a)) {
        memset(r, 0, sizeof *r);
        return;
    }
    u128 borrow = 0;
    for (int i = 0; i < 6; ++i) {
        u128 d = (u128)P[i] - a->l[i] - borrow;
        r->l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}

/* Montgomery product, coarsely-integrated operand scanning (Koc et al.);
 * same result as the reference's FsFp::mul_fp -> blst_fp_mul. */
void ofp_mul(ofp_t *r, const ofp_t *a, const ofp_t *b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 6; ++i) {
        u128 c = 0;
        for (int j = 0; j < 6; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (uint64_t)c;
        t[7] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * P_INV;
        c = (u128)m * P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 6; ++j) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = t[7] + (uint64_t)(c >> 64);
    }
    if (t[6] || geq_p(t)) sub_p(t);
    memcpy(r->l, t, 6 * sizeof(uint64_t));
}

void ofp_sqr(ofp_t *r, const ofp_t *a) { ofp_mul(r, a, a); }

void ofp_one(ofp_t *r) { memcpy(r->l, ONE, sizeof ONE); }

static void fp_pow(ofp_t *r, const ofp_t *a, const uint64_t e[6]) {
    ofp_t acc, base = *a;
    ofp_one(&acc);
    for (int i = 0; i < 6; ++i)
        for (int b = 0; b < 64; ++b) {
            if ((e[i] >> b) & 1) ofp_mul(&acc, &acc, &base);
            ofp_sqr(&base, &base);
        }
    *r = acc;
}

/* a^(p-2); blst uses a different inversion algorithm with the same value */
void ofp_inv(ofp_t *r, const ofp_t *a) { fp_pow(r, a, P_MINUS_2); }

/* p = 3 mod 4: candidate root a^((p+1)/4) */
int ofp_sqrt(ofp_t *r, const ofp_t *a) {
    ofp_t s, chk;
    fp_pow(&s, a, P_SQRT_EXP);
    ofp_sqr(&chk, &s);
    *r = s;
    return ofp_eq(&chk, a);
}

static void fp_from_mont(uint64_t out[6], const ofp_t *a) {
    ofp_t one_raw = {{1, 0, 0, 0, 0, 0}}, t;
    ofp_mul(&t, a, &one_raw);
    memcpy(out, t.l, sizeof t.l);
}

int ofp_from_be48(ofp_t *r, const uint8_t in[48]) {
    ofp_t raw, r2;
    for (int i = 0; i < 6; ++i) {
        uint64_t w = 0;
        for (int j = 0; j < 8; ++j) w = (w << 8) | in[(5 - i) * 8 + j];
        raw.l[i] = w;
    }
    if (geq_p(raw.l)) return 0;
    memcpy(r2.l, R2, sizeof R2);
    ofp_mul(r, &raw, &r2);
    return 1;
}

void ofp_to_be48(uint8_t out[48], const ofp_t *a) {
    uint64_t raw[6];
    fp_from_mont(raw, a);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 8; ++j) out[(5 - i) * 8 + j] = (uint8_t)(raw[i] >> (56 - 8 * j));
}

int ofp_is_lex_largest(const ofp_t *a) {
    uint64_t raw[6];
    fp_from_mont(raw, a);
    for (int i = 5; i >= 0; --i) {
        if (raw[i] > P_HALF[i]) return 1;
        if (raw[i] < P_HALF[i]) return 0;
    }
    return 0;
}
