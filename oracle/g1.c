/*
 * TEST INFRASTRUCTURE (see oracle.h).  BLS12-381 G1: y^2 = x^3 + 4 over Fp.
 * Restates what the reference gets from blst through FsG1 / FsG1Affine
 * (blst/src/types/g1.rs:28-441) and the XYZZ bucket arithmetic it carries
 * in-tree (kzg/src/msm/pippenger_utils.rs:84-210).
 */
#include "oracle.h"
#include <string.h>

static const ofp_t B4 = {{0xaa270000000cfff3ull, 0x53cc0032fc34000aull, 0x478fe97a6b0a807full,
                          0xb1d37ebee6ba24d7ull, 0x8ec9733bbf78ab2full, 0x09d645513d83de7eull}};
/* generator, Montgomery limbs as in blst/src/consts.rs:52-84 */
static const ofp_t GX = {{0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull,
                          0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull}};
static const ofp_t GY = {{0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull,
                          0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull}};

void og1_set_inf(og1_t *r) { memset(r, 0, sizeof *r); }
int og1_is_inf(const og1_t *a) { return ofp_is_zero(&a->z); }

void og1_generator(og1_t *r) {
    r->x = GX;
    r->y = GY;
    ofp_one(&r->z);
}

static int affine_is_inf(const og1_affine_t *a) { return ofp_is_zero(&a->x) && ofp_is_zero(&a->y); }

/* blst_p1_from_affine: (0,0) -> infinity */
void og1_from_affine(og1_t *r, const og1_affine_t *a) {
    if (affine_is_inf(a)) {
        og1_set_inf(r);
        return;
    }
    r->x = a->x;
    r->y = a->y;
    ofp_one(&r->z);
}

/* blst_p1_to_affine: (X/Z^2, Y/Z^3); infinity -> (0,0) */
void og1_to_affine(og1_affine_t *r, const og1_t *a) {
    if (og1_is_inf(a)) {
        memset(r, 0, sizeof *r);
        return;
    }
    ofp_t zi, zi2, zi3;
    ofp_inv(&zi, &a->z);
    ofp_sqr(&zi2, &zi);
    ofp_mul(&zi3, &zi2, &zi);
    ofp_mul(&r->x, &a->x, &zi2);
    ofp_mul(&r->y, &a->y, &zi3);
}

/* dbl-2009-l (a = 0) */
void og1_dbl(og1_t *r, const og1_t *p) {
    if (og1_is_inf(p)) {
        og1_set_inf(r);
        return;
    }
    ofp_t A, B, C, D, E, F, t, x3, y3, z3;
    ofp_sqr(&A, &p->x);
    ofp_sqr(&B, &p->y);
    ofp_sqr(&C, &B);
    ofp_add(&t, &p->x, &B);
    ofp_sqr(&t, &t);
    ofp_sub(&t, &t, &A);
    ofp_sub(&t, &t, &C);
    ofp_add(&D, &t, &t);
    ofp_add(&E, &A, &A);
    ofp_add(&E, &E, &A);
    ofp_sqr(&F, &E);
    ofp_sub(&x3, &F, &D);
    ofp_sub(&x3, &x3, &D);
    ofp_sub(&t, &D, &x3);
    ofp_mul(&y3, &E, &t);
    ofp_add(&C, &C, &C);
    ofp_add(&C, &C, &C);
    ofp_add(&C, &C, &C);
    ofp_sub(&y3, &y3, &C);
    ofp_mul(&z3, &p->y, &p->z);
    ofp_add(&z3, &z3, &z3);
    r->x = x3;
    r->y = y3;
    r->z = z3;
}

/* blst_p1_add_or_double: add-2007-bl with the exceptional cases handled */
void og1_add_or_dbl(og1_t *r, const og1_t *a, const og1_t *b) {
    if (og1_is_inf(a)) {
        *r = *b;
        return;
    }
    if (og1_is_inf(b)) {
        *r = *a;
        return;
    }
    ofp_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    ofp_sqr(&z1z1, &a->z);
    ofp_sqr(&z2z2, &b->z);
    ofp_mul(&u1, &a->x, &z2z2);
    ofp_mul(&u2, &b->x, &z1z1);
    ofp_mul(&s1, &a->y, &b->z);
    ofp_mul(&s1, &s1, &z2z2);
    ofp_mul(&s2, &b->y, &a->z);
    ofp_mul(&s2, &s2, &z1z1);
    ofp_sub(&h, &u2, &u1);
    ofp_sub(&rr, &s2, &s1);
    if (ofp_is_zero(&h)) {
        if (ofp_is_zero(&rr)) {
            og1_dbl(r, a);
        } else {
            og1_set_inf(r);
        }
        return;
    }
    ofp_add(&rr, &rr, &rr);
    ofp_add(&i, &h, &h);
    ofp_sqr(&i, &i);
    ofp_mul(&j, &h, &i);
    ofp_mul(&v, &u1, &i);
    ofp_sqr(&x3, &rr);
    ofp_sub(&x3, &x3, &j);
    ofp_sub(&x3, &x3, &v);
    ofp_sub(&x3, &x3, &v);
    ofp_sub(&t, &v, &x3);
    ofp_mul(&y3, &rr, &t);
    ofp_mul(&t, &s1, &j);
    ofp_add(&t, &t, &t);
    ofp_sub(&y3, &y3, &t);
    ofp_add(&z3, &a->z, &b->z);
    ofp_sqr(&z3, &z3);
    ofp_sub(&z3, &z3, &z1z1);
    ofp_sub(&z3, &z3, &z2z2);
    ofp_mul(&z3, &z3, &h);
    r->x = x3;
    r->y = y3;
    r->z = z3;
}

void og1_neg(og1_t *r, const og1_t *a) {
    r->x = a->x;
    ofp_neg(&r->y, &a->y);
    r->z = a->z;
}

/* FsG1::mul (g1.rs:242-273): plain double-and-add over the canonical scalar */
void og1_mul(og1_t *r, const og1_t *a, const ofr_t *s) {
    uint8_t le[32];
    ofr_to_scalar_le(le, s);
    og1_t acc;
    og1_set_inf(&acc);
    for (int bit = 254; bit >= 0; --bit) {
        og1_dbl(&acc, &acc);
        if ((le[bit >> 3] >> (bit & 7)) & 1) og1_add_or_dbl(&acc, &acc, a);
    }
    *r = acc;
}

/* blst_p1_is_equal: X1 Z2^2 == X2 Z1^2 and Y1 Z2^3 == Y2 Z1^3 (g1.rs:147-149) */
int og1_equal(const og1_t *a, const og1_t *b) {
    int ia = og1_is_inf(a), ib = og1_is_inf(b);
    if (ia || ib) return ia && ib;
    ofp_t z1z1, z2z2, l, r;
    ofp_sqr(&z1z1, &a->z);
    ofp_sqr(&z2z2, &b->z);
    ofp_mul(&l, &a->x, &z2z2);
    ofp_mul(&r, &b->x, &z1z1);
    if (!ofp_eq(&l, &r)) return 0;
    ofp_mul(&z1z1, &z1z1, &a->z);
    ofp_mul(&z2z2, &z2z2, &b->z);
    ofp_mul(&l, &a->y, &z2z2);
    ofp_mul(&r, &b->y, &z1z1);
    return ofp_eq(&l, &r);
}

int og1_affine_on_curve(const og1_affine_t *a) {
    if (affine_is_inf(a)) return 1;
    ofp_t l, r;
    ofp_sqr(&l, &a->y);
    ofp_sqr(&r, &a->x);
    ofp_mul(&r, &r, &a->x);
    ofp_add(&r, &r, &B4);
    return ofp_eq(&l, &r);
}

/* blst_p1_in_g1 (g1.rs:114-119): membership in the order-r subgroup; here by [r]P == inf */
int og1_in_subgroup(const og1_t *a) {
    static const uint64_t RM[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull,
                                   0x73eda753299d7d48ull};
    if (og1_is_inf(a)) return 1;
    og1_t acc;
    og1_set_inf(&acc);
    for (int bit = 254; bit >= 0; --bit) {
        og1_dbl(&acc, &acc);
        if ((RM[bit >> 6] >> (bit & 63)) & 1) og1_add_or_dbl(&acc, &acc, a);
    }
    return og1_is_inf(&acc);
}

/* ZCash compressed G1 (blst_p1_uncompress; format as in zkcrypto/bls12_381/src/g1.rs:337-392).
 * On-curve by construction; NO subgroup check (blst/src/types/g1.rs:65-87). */
int og1_uncompress(og1_affine_t *r, const uint8_t in[48]) {
    int compressed = (in[0] >> 7) & 1, infinity = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    if (!compressed) return 0;
    uint8_t tmp[48];
    memcpy(tmp, in, 48);
    tmp[0] &= 0x1f;
    if (infinity) {
        if (sort) return 0;
        for (int i = 0; i < 48; ++i)
            if (tmp[i]) return 0;
        memset(r, 0, sizeof *r);
        return 1;
    }
    ofp_t x, y, y2;
    if (!ofp_from_be48(&x, tmp)) return 0;
    ofp_sqr(&y2, &x);
    ofp_mul(&y2, &y2, &x);
    ofp_add(&y2, &y2, &B4);
    if (!ofp_sqrt(&y, &y2)) return 0;
    if (ofp_is_lex_largest(&y) != sort) ofp_neg(&y, &y);
    r->x = x;
    r->y = y;
    return 1;
}

/* blst_p1_compress (g1.rs:94-100) */
void og1_compress(uint8_t out[48], const og1_t *a) {
    if (og1_is_inf(a)) {
        memset(out, 0, 48);
        out[0] = 0xc0;
        return;
    }
    og1_affine_t af;
    og1_to_affine(&af, a);
    ofp_to_be48(out, &af.x);
    out[0] |= 0x80;
    if (ofp_is_lex_largest(&af.y)) out[0] |= 0x20;
}

static int xyzz_is_inf(const og1_xyzz_t *p) { return ofp_is_zero(&p->zzz) && ofp_is_zero(&p->zz); }

/* p1_dadd_affine, kzg/src/msm/pippenger_utils.rs:90-157 (EFD madd-2008-s + exceptional cases) */
void og1_xyzz_dadd_affine(og1_xyzz_t *out, const og1_affine_t *p2, int subtract) {
    if (affine_is_inf(p2)) return;
    if (xyzz_is_inf(out)) {
        out->x = p2->x;
        out->y = p2->y;
        ofp_one(&out->zzz);
        if (subtract) ofp_neg(&out->zzz, &out->zzz);
        ofp_one(&out->zz);
        return;
    }
    ofp_t p, r, pp, ppp, q, t;
    ofp_mul(&p, &p2->x, &out->zz);
    ofp_mul(&r, &p2->y, &out->zzz);
    if (subtract) ofp_neg(&r, &r);
    ofp_sub(&p, &p, &out->x);
    ofp_sub(&r, &r, &out->y);
    if (!ofp_is_zero(&p)) {
        ofp_sqr(&pp, &p);
        ofp_mul(&ppp, &pp, &p);
        ofp_mul(&q, &out->x, &pp);
        ofp_sqr(&out->x, &r);
        ofp_add(&t, &q, &q);
        ofp_sub(&out->x, &out->x, &ppp);
        ofp_sub(&out->x, &out->x, &t);
        ofp_sub(&q, &q, &out->x);
        ofp_mul(&q, &q, &r);
        ofp_mul(&out->y, &out->y, &ppp);
        ofp_sub(&out->y, &q, &out->y);
        ofp_mul(&out->zz, &out->zz, &pp);
        ofp_mul(&out->zzz, &out->zzz, &ppp);
    } else if (ofp_is_zero(&r)) {
        /* doubling of the affine point (mdbl-2008-s-1) */
        ofp_t u, s, m;
        ofp_add(&u, &p2->y, &p2->y);
        ofp_sqr(&out->zz, &u);
        ofp_mul(&out->zzz, &out->zz, &u);
        ofp_mul(&s, &p2->x, &out->zz);
        ofp_sqr(&m, &p2->x);
        ofp_add(&t, &m, &m);
        ofp_add(&m, &t, &m);
        ofp_sqr(&out->x, &m);
        ofp_add(&u, &s, &s);
        ofp_sub(&out->x, &out->x, &u);
        ofp_mul(&out->y, &out->zzz, &p2->y);
        ofp_sub(&s, &s, &out->x);
        ofp_mul(&s, &s, &m);
        ofp_sub(&out->y, &s, &out->y);
        if (subtract) ofp_neg(&out->zzz, &out->zzz);
    } else {
        memset(&out->zzz, 0, sizeof out->zzz);
        memset(&out->zz, 0, sizeof out->zz);
    }
}

/* p1_dadd, pippenger_utils.rs:159-210 (EFD add-2008-s + exceptional cases) */
void og1_xyzz_dadd(og1_xyzz_t *out, const og1_xyzz_t *p2) {
    if (xyzz_is_inf(p2)) return;
    if (xyzz_is_inf(out)) {
        *out = *p2;
        return;
    }
    ofp_t u, s, p, r, pp, ppp, q, t;
    ofp_mul(&u, &out->x, &p2->zz);
    ofp_mul(&s, &out->y, &p2->zzz);
    ofp_mul(&p, &p2->x, &out->zz);
    ofp_mul(&r, &p2->y, &out->zzz);
    ofp_sub(&p, &p, &u);
    ofp_sub(&r, &r, &s);
    if (!ofp_is_zero(&p)) {
        ofp_sqr(&pp, &p);
        ofp_mul(&ppp, &pp, &p);
        ofp_mul(&q, &u, &pp);
        ofp_sqr(&out->x, &r);
        ofp_add(&t, &q, &q);
        ofp_sub(&out->x, &out->x, &ppp);
        ofp_sub(&out->x, &out->x, &t);
        ofp_sub(&q, &q, &out->x);
        ofp_mul(&q, &q, &r);
        ofp_mul(&out->y, &s, &ppp);
        ofp_sub(&out->y, &q, &out->y);
        ofp_mul(&out->zz, &out->zz, &p2->zz);
        ofp_mul(&out->zzz, &out->zzz, &p2->zzz);
        ofp_mul(&out->zz, &out->zz, &pp);
        ofp_mul(&out->zzz, &out->zzz, &ppp);
    } else if (ofp_is_zero(&r)) {
        /* doubling (dbl-2008-s-1) */
        ofp_t v, w, m;
        ofp_add(&u, &out->y, &out->y);
        ofp_sqr(&v, &u);
        ofp_mul(&w, &v, &u);
        ofp_mul(&s, &out->x, &v);
        ofp_sqr(&m, &out->x);
        ofp_add(&t, &m, &m);
        ofp_add(&m, &t, &m);
        ofp_sqr(&out->x, &m);
        ofp_add(&u, &s, &s);
        ofp_sub(&out->x, &out->x, &u);
        ofp_mul(&out->y, &w, &out->y);
        ofp_sub(&s, &s, &out->x);
        ofp_mul(&s, &s, &m);
        ofp_sub(&out->y, &s, &out->y);
        ofp_mul(&out->zz, &out->zz, &v);
        ofp_mul(&out->zzz, &out->zzz, &w);
    } else {
        memset(&out->zzz, 0, sizeof out->zzz);
        memset(&out->zz, 0, sizeof out->zz);
    }
}

/* p1_to_jacobian, pippenger_utils.rs:84-88 */
void og1_xyzz_to_jacobian(og1_t *out, const og1_xyzz_t *in) {
    ofp_mul(&out->x, &in->x, &in->zz);
    ofp_mul(&out->y, &in->y, &in->zzz);
    out->z = in->zz;
}
