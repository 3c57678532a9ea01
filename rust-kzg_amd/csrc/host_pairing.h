// Host-side G2 and pairing arithmetic for the verify_* half of the c-kzg-4844 surface.
//
// The reference keeps the pairing on the CPU whatever backend computes the MSMs (blst's Miller loop behind
// PairingVerify, blst/src/kzg_proofs.rs:73-100; with the sppark feature only g1_lincomb moves to the GPU).  This
// library does the same: the field work and the G1 linear combinations of verification run on the GPU, the final
//     e(a1, a2) == e(b1, b2)
// check — two Miller loops and one final exponentiation per call, whatever the batch size — runs here.  It is not
// a fallback for anything the GPU path computes, and it shares no code with the CPU checker of the test-suite.
//
// Construction (textbook; ~0.7 ms per check on one core of the GPU box with the cached line tables below):
//   tower      Fp2 = Fp[u]/(u^2+1),  Fp6 = Fp2[v]/(v^3 - xi), xi = 1+u,  Fp12 = Fp6[w]/(w^2 - v)
//   twist      E'(Fp2): y^2 = x^3 + 4 xi  (M-type);  (x', y') -> (x' w^-2, y' w^-3) lands on E(Fp12): y^2 = x^3 + 4
//   Miller     optimal-ate loop over |x| = 0xd201000000010000 with affine arithmetic on E', line through T evaluated
//              at P = (xP, yP) in G1 and scaled by w^3 (a constant of the subfield Fp4, killed by the final
//              exponentiation):   l = (lambda' x'_T - y'_T) + (-lambda' xP) v + yP (v w)
//   final exp  easy part by conjugation / inversion / Frobenius, hard part by the five-exponentiation chain in x;
//              Frobenius constants xi^(k(p-1)/6) computed at first use
// Bilinearity and non-degeneracy are what the callers rely on (product-of-pairings == 1 checks); both are tested
// on the CPU against the reference's verify_kzg_proof vectors (tests/test_pairing_cpu.py).
#pragma once
#include <stdint.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "ff.hip.h"
#include "host_fp64.h"
#include "host_g1.h"

namespace kzgamd {
namespace pairing {

using ff::Fp;

// ---------------------------------------------------------------- Fp2
struct Fp2 {
    Fp c0, c1;
};
inline Fp2 f2_zero() { return {Fp::zero(), Fp::zero()}; }
inline Fp2 f2_one() { return {Fp::one(), Fp::zero()}; }
inline bool f2_is_zero(const Fp2& a) { return a.c0.is_zero() && a.c1.is_zero(); }
inline bool f2_eq(const Fp2& a, const Fp2& b) { return a.c0 == b.c0 && a.c1 == b.c1; }
inline Fp2 f2_add(const Fp2& a, const Fp2& b) { return {hfp::add(a.c0, b.c0), hfp::add(a.c1, b.c1)}; }
inline Fp2 f2_sub(const Fp2& a, const Fp2& b) { return {hfp::sub(a.c0, b.c0), hfp::sub(a.c1, b.c1)}; }
inline Fp2 f2_neg(const Fp2& a) { return {hfp::neg(a.c0), hfp::neg(a.c1)}; }
inline Fp2 f2_dbl(const Fp2& a) { return {hfp::dbl(a.c0), hfp::dbl(a.c1)}; }
inline Fp2 f2_conj(const Fp2& a) { return {a.c0, hfp::neg(a.c1)}; }
inline Fp2 f2_mul(const Fp2& a, const Fp2& b) {  // Karatsuba: 3 Fp multiplications
    const Fp t0 = hfp::mul(a.c0, b.c0), t1 = hfp::mul(a.c1, b.c1);
    const Fp t2 = hfp::mul(hfp::add(a.c0, a.c1), hfp::add(b.c0, b.c1));
    return {hfp::sub(t0, t1), hfp::sub(hfp::sub(t2, t0), t1)};
}
inline Fp2 f2_sqr(const Fp2& a) {  // (a0+a1)(a0-a1), 2 a0 a1
    const Fp t = hfp::mul(a.c0, a.c1);
    return {hfp::mul(hfp::add(a.c0, a.c1), hfp::sub(a.c0, a.c1)), hfp::dbl(t)};
}
inline Fp2 f2_mul_fp(const Fp2& a, const Fp& b) { return {hfp::mul(a.c0, b), hfp::mul(a.c1, b)}; }
inline Fp2 f2_mul_xi(const Fp2& a) { return {hfp::sub(a.c0, a.c1), hfp::add(a.c0, a.c1)}; }  // * (1 + u)
inline Fp2 f2_inv(const Fp2& a) {  // conj(a) / (a0^2 + a1^2)
    const Fp n = ff::inverse_bgcd(hfp::add(hfp::sqr(a.c0), hfp::sqr(a.c1)));
    return {hfp::mul(a.c0, n), hfp::neg(hfp::mul(a.c1, n))};
}
inline Fp2 f2_pow(const Fp2& a, const uint32_t* e, int nlimbs) {
    Fp2 r = f2_one();
    bool started = false;
    for (int i = nlimbs * 32 - 1; i >= 0; --i) {
        if (started) r = f2_sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            r = started ? f2_mul(r, a) : a;
            started = true;
        }
    }
    return r;
}
// square root for p = 3 mod 4 (Adj–Rodriguez-Henriquez, alg. 9); false if `a` is not a square
inline bool f2_sqrt(Fp2& out, const Fp2& a) {
    static const uint32_t EXP_PM3_4[12] = {0xffffeaaau, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                           0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
    static const uint32_t EXP_PM1_2[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                           0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
    if (f2_is_zero(a)) {
        out = a;
        return true;
    }
    const Fp2 a1 = f2_pow(a, EXP_PM3_4, 12);
    const Fp2 alpha = f2_mul(f2_sqr(a1), a);
    const Fp2 x0 = f2_mul(a1, a);
    const Fp2 minus_one = f2_neg(f2_one());
    Fp2 x;
    if (f2_eq(alpha, minus_one)) {
        x = {hfp::neg(x0.c1), x0.c0};  // u * x0
    } else {
        const Fp2 b = f2_pow(f2_add(f2_one(), alpha), EXP_PM1_2, 12);
        x = f2_mul(b, x0);
    }
    if (!f2_eq(f2_sqr(x), a)) return false;
    out = x;
    return true;
}

// ---------------------------------------------------------------- Fp6 = Fp2[v]/(v^3 - xi)
struct Fp6 {
    Fp2 c0, c1, c2;
};
inline Fp6 f6_zero() { return {f2_zero(), f2_zero(), f2_zero()}; }
inline Fp6 f6_one() { return {f2_one(), f2_zero(), f2_zero()}; }
inline Fp6 f6_add(const Fp6& a, const Fp6& b) { return {f2_add(a.c0, b.c0), f2_add(a.c1, b.c1), f2_add(a.c2, b.c2)}; }
inline Fp6 f6_sub(const Fp6& a, const Fp6& b) { return {f2_sub(a.c0, b.c0), f2_sub(a.c1, b.c1), f2_sub(a.c2, b.c2)}; }
inline Fp6 f6_neg(const Fp6& a) { return {f2_neg(a.c0), f2_neg(a.c1), f2_neg(a.c2)}; }
inline Fp6 f6_mul_v(const Fp6& a) { return {f2_mul_xi(a.c2), a.c0, a.c1}; }
inline Fp6 f6_mul(const Fp6& a, const Fp6& b) {  // Karatsuba: 6 Fp2 multiplications
    const Fp2 t0 = f2_mul(a.c0, b.c0), t1 = f2_mul(a.c1, b.c1), t2 = f2_mul(a.c2, b.c2);
    const Fp2 s12 = f2_sub(f2_sub(f2_mul(f2_add(a.c1, a.c2), f2_add(b.c1, b.c2)), t1), t2);  // a1b2 + a2b1
    const Fp2 s01 = f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c1), f2_add(b.c0, b.c1)), t0), t1);  // a0b1 + a1b0
    const Fp2 s02 = f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c2), f2_add(b.c0, b.c2)), t0), t2);  // a0b2 + a2b0
    return {f2_add(t0, f2_mul_xi(s12)), f2_add(s01, f2_mul_xi(t2)), f2_add(s02, t1)};
}
inline Fp6 f6_inv(const Fp6& a) {
    const Fp2 A = f2_sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
    const Fp2 B = f2_sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
    const Fp2 C = f2_sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
    const Fp2 F = f2_add(f2_mul(a.c0, A), f2_mul_xi(f2_add(f2_mul(a.c2, B), f2_mul(a.c1, C))));
    const Fp2 Fi = f2_inv(F);
    return {f2_mul(A, Fi), f2_mul(B, Fi), f2_mul(C, Fi)};
}

// ---------------------------------------------------------------- Fp12 = Fp6[w]/(w^2 - v)
struct Fp12 {
    Fp6 c0, c1;
};
inline Fp12 f12_one() { return {f6_one(), f6_zero()}; }
inline Fp12 f12_mul(const Fp12& a, const Fp12& b) {
    const Fp6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
    const Fp6 m = f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1));
    return {f6_add(t0, f6_mul_v(t1)), f6_sub(f6_sub(m, t0), t1)};
}
// (a + b w)^2 = (a + b)(a + v b) - ab - v ab  +  2ab w : two Fp6 multiplications
inline Fp12 f12_sqr(const Fp12& a) {
    const Fp6 ab = f6_mul(a.c0, a.c1);
    const Fp6 t = f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_v(a.c1)));
    return {f6_sub(f6_sub(t, ab), f6_mul_v(ab)), f6_add(ab, ab)};
}
// Squaring in the cyclotomic subgroup (Granger-Scott, "Faster squaring in the cyclotomic subgroup of sixth degree
// extensions", PKC 2010, for the tower 2-3-2): three Fp4 squarings, 18 Fp multiplications against 36.
// Only valid for elements of norm one over Fp6, i.e. after the easy part of the final exponentiation.
inline Fp12 f12_cyclotomic_sqr(const Fp12& f) {
    auto fp4_sqr = [](const Fp2& a, const Fp2& b, Fp2& r0, Fp2& r1) {  // (a + b y)^2, y^2 = xi
        const Fp2 ab = f2_mul(a, b);
        r0 = f2_sub(f2_sub(f2_mul(f2_add(a, b), f2_add(a, f2_mul_xi(b))), ab), f2_mul_xi(ab));
        r1 = f2_dbl(ab);
    };
    const Fp2 &z0 = f.c0.c0, &z4 = f.c0.c1, &z3 = f.c0.c2, &z2 = f.c1.c0, &z1 = f.c1.c1, &z5 = f.c1.c2;
    Fp2 t0, t1, t2, t3, t4, t5;
    fp4_sqr(z0, z1, t0, t1);
    fp4_sqr(z2, z3, t2, t3);
    fp4_sqr(z4, z5, t4, t5);
    auto three_minus_two = [](const Fp2& t, const Fp2& z) { return f2_add(f2_dbl(f2_sub(t, z)), t); };  // 3t - 2z
    auto three_plus_two = [](const Fp2& t, const Fp2& z) { return f2_add(f2_dbl(f2_add(t, z)), t); };    // 3t + 2z
    Fp12 r;
    r.c0.c0 = three_minus_two(t0, z0);
    r.c1.c1 = three_plus_two(t1, z1);
    r.c1.c0 = three_plus_two(f2_mul_xi(t5), z2);
    r.c0.c2 = three_minus_two(t4, z3);
    r.c0.c1 = three_minus_two(t2, z4);
    r.c1.c2 = three_plus_two(t3, z5);
    return r;
}
// f * (a + b v + c v w) for a line value: a, b in Fp2, c in Fp
inline Fp12 f12_mul_line(const Fp12& f, const Fp2& a, const Fp2& b, const Fp& c) {
    auto mul_ab0 = [](const Fp6& x, const Fp2& a2, const Fp2& b2) -> Fp6 {  // x * (a2 + b2 v)
        const Fp2 t0 = f2_mul(x.c0, a2), t1 = f2_mul(x.c1, b2);
        const Fp2 m01 = f2_sub(f2_sub(f2_mul(f2_add(x.c0, x.c1), f2_add(a2, b2)), t0), t1);  // x0 b + x1 a
        return {f2_add(t0, f2_mul_xi(f2_mul(x.c2, b2))), m01, f2_add(t1, f2_mul(x.c2, a2))};
    };
    const Fp6 t0 = mul_ab0(f.c0, a, b);
    // f1 * (c v) = (xi x2 c, x0 c, x1 c)
    const Fp6 t1 = {f2_mul_xi(f2_mul_fp(f.c1.c2, c)), f2_mul_fp(f.c1.c0, c), f2_mul_fp(f.c1.c1, c)};
    const Fp2 bc = {hfp::add(b.c0, c), b.c1};
    const Fp6 m = mul_ab0(f6_add(f.c0, f.c1), a, bc);
    return {f6_add(t0, f6_mul_v(t1)), f6_sub(f6_sub(m, t0), t1)};
}
inline Fp12 f12_conj(const Fp12& a) { return {a.c0, f6_neg(a.c1)}; }  // = a^(p^6)
inline Fp12 f12_inv(const Fp12& a) {
    const Fp6 d = f6_inv(f6_sub(f6_mul(a.c0, a.c0), f6_mul_v(f6_mul(a.c1, a.c1))));
    return {f6_mul(a.c0, d), f6_neg(f6_mul(a.c1, d))};
}
inline bool f12_is_one(const Fp12& a) {
    return f2_eq(a.c0.c0, f2_one()) && f2_is_zero(a.c0.c1) && f2_is_zero(a.c0.c2) && f2_is_zero(a.c1.c0) &&
           f2_is_zero(a.c1.c1) && f2_is_zero(a.c1.c2);
}

// ---------------------------------------------------------------- G2 on the twist E'(Fp2): y^2 = x^3 + 4(1+u)
struct G2Affine {
    Fp2 x, y;
    bool inf;
};
struct G2Jac {  // blst_p2 layout: x, y, z as Fp2 in Montgomery form; infinity <=> z == 0
    Fp2 x, y, z;
};
static_assert(sizeof(G2Jac) == sizeof(blst_p2), "blst_p2 layout");

inline Fp fp_from_plain(const uint32_t v[12]) {
    Fp a;
    for (int i = 0; i < 12; ++i) a.v[i] = v[i];
    return hfp::to_mont(a);
}
inline Fp2 b_twist() {  // 4 (1 + u)
    Fp four = Fp::zero();
    four.v[0] = 4;
    four = hfp::to_mont(four);
    return {four, four};
}
inline G2Jac g2_generator() {
    static const uint32_t X0[12] = {0xc121bdb8u, 0xd48056c8u, 0xa805bbefu, 0x0bac0326u, 0x7ae3d177u, 0xb4510b64u,
                                    0xfa403b02u, 0xc6e47ad4u, 0x2dc51051u, 0x26080527u, 0xf08f0a91u, 0x024aa2b2u};
    static const uint32_t X1[12] = {0x5d042b7eu, 0xe5ac7d05u, 0x13945d57u, 0x334cf112u, 0xdc7f5049u, 0xb5da61bbu,
                                    0x9920b61au, 0x596bd0d0u, 0x88274f65u, 0x7dacd3a0u, 0x52719f60u, 0x13e02b60u};
    static const uint32_t Y0[12] = {0x08b82801u, 0xe1935486u, 0x3baca289u, 0x923ac9ccu, 0x5160d12cu, 0x6d429a69u,
                                    0x8cbdd3a7u, 0xadfd9baau, 0xda2e351au, 0x8cc9cdc6u, 0x727d6e11u, 0x0ce5d527u};
    static const uint32_t Y1[12] = {0xf05f79beu, 0xaaa9075fu, 0x5cec1da1u, 0x3f370d27u, 0x572e99abu, 0x267492abu,
                                    0x85a763afu, 0xcb3e287eu, 0x2bc28b99u, 0x32acd2b0u, 0x2ea734ccu, 0x0606c4a0u};
    return {{fp_from_plain(X0), fp_from_plain(X1)}, {fp_from_plain(Y0), fp_from_plain(Y1)}, f2_one()};
}
inline G2Jac g2_inf() { return {f2_zero(), f2_zero(), f2_zero()}; }
inline bool g2_is_inf(const G2Jac& p) { return f2_is_zero(p.z); }
inline G2Jac g2_dbl(const G2Jac& p) {  // dbl-2009-l over Fp2
    if (g2_is_inf(p)) return p;
    const Fp2 A = f2_sqr(p.x), B = f2_sqr(p.y), C = f2_sqr(B);
    const Fp2 t = f2_sub(f2_sub(f2_sqr(f2_add(p.x, B)), A), C);
    const Fp2 D = f2_dbl(t), E = f2_add(f2_dbl(A), A), F = f2_sqr(E);
    G2Jac r;
    r.x = f2_sub(f2_sub(F, D), D);
    r.y = f2_sub(f2_mul(E, f2_sub(D, r.x)), f2_dbl(f2_dbl(f2_dbl(C))));
    r.z = f2_dbl(f2_mul(p.y, p.z));
    return r;
}
inline G2Jac g2_add(const G2Jac& a, const G2Jac& b) {  // add-2007-bl with the exceptional cases
    if (g2_is_inf(a)) return b;
    if (g2_is_inf(b)) return a;
    const Fp2 z1z1 = f2_sqr(a.z), z2z2 = f2_sqr(b.z);
    const Fp2 u1 = f2_mul(a.x, z2z2), u2 = f2_mul(b.x, z1z1);
    const Fp2 s1 = f2_mul(f2_mul(a.y, b.z), z2z2), s2 = f2_mul(f2_mul(b.y, a.z), z1z1);
    const Fp2 h = f2_sub(u2, u1);
    Fp2 rr = f2_sub(s2, s1);
    if (f2_is_zero(h)) return f2_is_zero(rr) ? g2_dbl(a) : g2_inf();
    rr = f2_dbl(rr);
    const Fp2 i = f2_sqr(f2_dbl(h)), j = f2_mul(h, i), v = f2_mul(u1, i);
    G2Jac r;
    r.x = f2_sub(f2_sub(f2_sub(f2_sqr(rr), j), v), v);
    r.y = f2_sub(f2_mul(rr, f2_sub(v, r.x)), f2_dbl(f2_mul(s1, j)));
    r.z = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(a.z, b.z)), z1z1), z2z2), h);
    return r;
}
inline G2Jac g2_neg(const G2Jac& a) { return {a.x, f2_neg(a.y), a.z}; }
// k = canonical (non-Montgomery) little-endian 256-bit scalar
inline G2Jac g2_mul(const G2Jac& p, const uint32_t k[8]) {
    G2Jac acc = g2_inf();
    for (int bit = 255; bit >= 0; --bit) {
        acc = g2_dbl(acc);
        if ((k[bit >> 5] >> (bit & 31)) & 1) acc = g2_add(acc, p);
    }
    return acc;
}
inline G2Affine g2_to_affine(const G2Jac& p) {
    if (g2_is_inf(p)) return {f2_zero(), f2_zero(), true};
    const Fp2 zi = f2_inv(p.z), zi2 = f2_sqr(zi);
    return {f2_mul(p.x, zi2), f2_mul(p.y, f2_mul(zi2, zi)), false};
}
inline bool g2_equal(const G2Jac& a, const G2Jac& b) {
    const bool ia = g2_is_inf(a), ib = g2_is_inf(b);
    if (ia || ib) return ia && ib;
    const Fp2 z1z1 = f2_sqr(a.z), z2z2 = f2_sqr(b.z);
    if (!f2_eq(f2_mul(a.x, z2z2), f2_mul(b.x, z1z1))) return false;
    return f2_eq(f2_mul(a.y, f2_mul(z2z2, b.z)), f2_mul(b.y, f2_mul(z1z1, a.z)));
}
// y is "lexicographically largest": compare the imaginary part first, the real part when it is zero
inline bool f2_lex_largest(const Fp2& y) {
    const Fp c1 = hfp::from_mont(y.c1);
    if (!c1.is_zero()) return host_fp_lex_largest(c1);
    return host_fp_lex_largest(hfp::from_mont(y.c0));
}
// blst_p2_uncompress + blst_p2_from_affine (FsG2::from_bytes, blst/src/types/g2.rs:52-75): ZCash format, 96 bytes =
// x.c1 (flags in the top three bits) | x.c0, big-endian; the point is checked to be on the curve, not in the subgroup
inline bool g2_uncompress(G2Jac& out, const uint8_t in[96]) {
    const bool compressed = (in[0] >> 7) & 1, infinity = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    if (!compressed) return false;
    uint8_t tmp[48];
    memcpy(tmp, in, 48);
    tmp[0] &= 0x1f;
    bool lt1 = false, lt0 = false;
    const Fp x1 = host_fp_from_be48(tmp, &lt1), x0 = host_fp_from_be48(in + 48, &lt0);
    if (infinity) {
        if (sort || !x1.is_zero() || !x0.is_zero()) return false;
        out = g2_inf();
        return true;
    }
    if (!lt1 || !lt0) return false;
    const Fp2 x = {hfp::to_mont(x0), hfp::to_mont(x1)};
    const Fp2 y2 = f2_add(f2_mul(f2_sqr(x), x), b_twist());
    Fp2 y;
    if (!f2_sqrt(y, y2)) return false;
    if (f2_lex_largest(y) != sort) y = f2_neg(y);
    out = {x, y, f2_one()};
    return true;
}
inline void g2_compress(uint8_t out[96], const G2Jac& p) {
    if (g2_is_inf(p)) {
        memset(out, 0, 96);
        out[0] = 0xc0;
        return;
    }
    const G2Affine a = g2_to_affine(p);
    host_fp_to_be48(out, hfp::from_mont(a.x.c1));
    host_fp_to_be48(out + 48, hfp::from_mont(a.x.c0));
    out[0] |= 0x80;
    if (f2_lex_largest(a.y)) out[0] |= 0x20;
}

// ---------------------------------------------------------------- Miller loop and final exponentiation
// The G2 side of a Miller loop does not depend on the G1 point: for Q on the twist (affine) the walk T = Q, 2Q, ...
// over the bits of |x| gives, per doubling / addition step, the slope lambda and c = lambda x_T - y_T; the line through
// the step evaluated at P = (xP, yP) is  c + (-lambda xP) v + yP (v w).  KZG verification only ever pairs with two
// G2 points per setup ([1]G2 and [tau]G2), so the tables are cached (prepared_lines) and a pairing check does no G2
// arithmetic and no inversion at all.
struct LineTable {
    bool inf = true;
    std::vector<Fp2> lambda, c;  // 63 doublings + 5 additions, in loop order
};
constexpr uint64_t BLS_X_ABS = 0xd201000000010000ull;
inline LineTable g2_line_table(const G2Affine& Q) {
    LineTable t;
    t.inf = Q.inf;
    if (Q.inf) return t;
    Fp2 tx = Q.x, ty = Q.y;
    for (int bit = 62; bit >= 0; --bit) {
        // doubling step: lambda = 3 x^2 / (2 y)   (y != 0: the order of T is odd)
        const Fp2 x2 = f2_sqr(tx);
        const Fp2 lambda = f2_mul(f2_add(f2_dbl(x2), x2), f2_inv(f2_dbl(ty)));
        t.lambda.push_back(lambda);
        t.c.push_back(f2_sub(f2_mul(lambda, tx), ty));
        const Fp2 nx = f2_sub(f2_sqr(lambda), f2_dbl(tx));
        ty = f2_sub(f2_mul(lambda, f2_sub(tx, nx)), ty);
        tx = nx;
        if ((BLS_X_ABS >> bit) & 1) {
            // addition step T + Q (T != +-Q inside the loop: T = kQ with 1 < k < r - 1)
            const Fp2 lam = f2_mul(f2_sub(Q.y, ty), f2_inv(f2_sub(Q.x, tx)));
            t.lambda.push_back(lam);
            t.c.push_back(f2_sub(f2_mul(lam, tx), ty));
            const Fp2 ax = f2_sub(f2_sub(f2_sqr(lam), tx), Q.x);
            ty = f2_sub(f2_mul(lam, f2_sub(tx, ax)), ty);
            tx = ax;
        }
    }
    return t;
}
// the table of a G2 point, from a small cache keyed by the point's 288 bytes (blst_p2 layout)
inline std::shared_ptr<const LineTable> prepared_lines(const G2Jac& q) {
    struct Entry {
        G2Jac key;
        std::shared_ptr<const LineTable> tab;
    };
    static std::mutex mu;
    static std::vector<Entry> cache;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Entry& e : cache)
            if (memcmp(&e.key, &q, sizeof q) == 0) return e.tab;
    }
    auto tab = std::make_shared<const LineTable>(g2_line_table(g2_to_affine(q)));
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() >= 8) cache.erase(cache.begin());
    cache.push_back({q, tab});
    return tab;
}
// prod_k f_{|x|,Q_k}(P_k), conjugated (the BLS parameter is negative): one shared squaring per bit, one sparse
// multiplication per pair and step.  P_k = (xP[k], yP[k]) affine, Montgomery; pairs with an infinite side are skipped.
inline Fp12 miller_loop_multi(const LineTable* const* tabs, const Fp* xP, const Fp* yP, const bool* p_inf, int n) {
    Fp12 f = f12_one();
    size_t idx = 0;
    auto step = [&]() {
        for (int k = 0; k < n; ++k) {
            if (p_inf[k] || tabs[k]->inf) continue;
            f = f12_mul_line(f, tabs[k]->c[idx], f2_neg(f2_mul_fp(tabs[k]->lambda[idx], xP[k])), yP[k]);
        }
        ++idx;
    };
    for (int bit = 62; bit >= 0; --bit) {
        f = f12_sqr(f);
        step();
        if ((BLS_X_ABS >> bit) & 1) step();
    }
    return f12_conj(f);
}
inline Fp12 miller_loop(const G2Affine& Q, const Fp& xP, const Fp& yP, bool p_inf) {
    const LineTable t = g2_line_table(Q);
    const LineTable* tp = &t;
    return miller_loop_multi(&tp, &xP, &yP, &p_inf, 1);
}
// Frobenius f -> f^p in the tower: conjugation on every Fp2 coefficient, times xi^(k (p-1)/6) for the coefficient of
// w^k (v = w^2): the six constants are computed once, by exponentiation, from xi = 1 + u.
struct FrobeniusConstants {
    Fp2 g[6];  // g[k] = xi^(k (p - 1) / 6)
    FrobeniusConstants() {
        static const uint32_t EXP_PM1_6[12] = {0xfffff1c7u, 0x49aa7fffu, 0x72e35555u, 0x051caaaau, 0xd3c82906u, 0xe688231au,
                                               0x7deb831fu, 0xe613e1ebu, 0xb5e1f223u, 0x0c849bf3u, 0x5eeaa66fu, 0x045582fcu};
        const Fp2 xi = {Fp::one(), Fp::one()};
        g[0] = f2_one();
        g[1] = f2_pow(xi, EXP_PM1_6, 12);
        for (int k = 2; k < 6; ++k) g[k] = f2_mul(g[k - 1], g[1]);
    }
};
inline const FrobeniusConstants& frobenius_constants() {
    static const FrobeniusConstants c;
    return c;
}
inline Fp12 f12_frobenius(const Fp12& a) {
    const FrobeniusConstants& k = frobenius_constants();
    Fp12 r;
    r.c0 = {f2_conj(a.c0.c0), f2_mul(f2_conj(a.c0.c1), k.g[2]), f2_mul(f2_conj(a.c0.c2), k.g[4])};   // 1, v, v^2
    r.c1 = {f2_mul(f2_conj(a.c1.c0), k.g[1]), f2_mul(f2_conj(a.c1.c1), k.g[3]), f2_mul(f2_conj(a.c1.c2), k.g[5])};  // w, vw, v^2 w
    return r;
}
// conj(f^|x|) = f^x for the (negative) BLS parameter, f in the cyclotomic subgroup (where inversion = conjugation)
inline Fp12 f12_exp_x(const Fp12& f) {
    const uint64_t X = 0xd201000000010000ull;
    Fp12 r = f;
    for (int bit = 62; bit >= 0; --bit) {
        r = f12_cyclotomic_sqr(r);
        if ((X >> bit) & 1) r = f12_mul(r, f);
    }
    return f12_conj(r);
}
// f^((p^12 - 1) / r * 3): easy part (p^6 - 1)(p^2 + 1) by conjugation, one inversion and two Frobenius maps; hard part
// by the addition chain in the BLS parameter of Hayashida-Hayasaka-Teruya (five exponentiations by x, as laid out in
// "Guide to Pairing-Based Cryptography", alg. 5.5.4, and in zkcrypto/bls12_381/src/pairings.rs:134-170).  The extra
// factor 3 is coprime to r: "== 1" is unaffected.
inline Fp12 final_exponentiation(const Fp12& f) {
    Fp12 t2 = f12_mul(f12_conj(f), f12_inv(f));                       // f^(p^6 - 1)
    t2 = f12_mul(f12_frobenius(f12_frobenius(t2)), t2);               // ^(p^2 + 1): now in the cyclotomic subgroup
    Fp12 t1 = f12_conj(f12_cyclotomic_sqr(t2));
    Fp12 t3 = f12_exp_x(t2);
    Fp12 t4 = f12_cyclotomic_sqr(t3);
    Fp12 t5 = f12_mul(t1, t3);
    t1 = f12_exp_x(t5);
    Fp12 t0 = f12_exp_x(t1);
    Fp12 t6 = f12_mul(f12_exp_x(t0), t4);
    t4 = f12_exp_x(t6);
    t5 = f12_conj(t5);
    t4 = f12_mul(t4, f12_mul(t5, t2));
    t5 = f12_conj(t2);
    t1 = f12_mul(t1, t2);
    t1 = f12_frobenius(f12_frobenius(f12_frobenius(t1)));
    t6 = f12_frobenius(f12_mul(t6, t5));
    t3 = f12_frobenius(f12_frobenius(f12_mul(t3, t0)));
    return f12_mul(f12_mul(f12_mul(t3, t1), t6), t4);
}

// pairings_verify (blst/src/kzg_proofs.rs:73-100):  e(a1, a2) == e(b1, b2)
inline bool pairings_verify(const blst_p1* a1, const blst_p2* a2, const blst_p1* b1, const blst_p2* b2) {
    G2Jac A2, B2;
    memcpy(&A2, a2, sizeof A2);
    memcpy(&B2, b2, sizeof B2);
    auto g1_affine = [](const blst_p1* p, Fp& x, Fp& y) -> bool {  // returns "is infinity"
        const Fp* P = reinterpret_cast<const Fp*>(p);
        if (P[2].is_zero()) return true;
        const Fp zi = ff::inverse_bgcd(P[2]), zi2 = hfp::sqr(zi);
        x = hfp::mul(P[0], zi2);
        y = hfp::mul(P[1], hfp::mul(zi2, zi));
        return false;
    };
    Fp ax, ay, bx, by;
    const bool ainf = g1_affine(a1, ax, ay), binf = g1_affine(b1, bx, by);
    if (!ainf) ay = hfp::neg(ay);  // e(-a1, a2) * e(b1, b2) == 1
    const std::shared_ptr<const LineTable> ta = prepared_lines(A2), tb = prepared_lines(B2);
    const LineTable* tabs[2] = {ta.get(), tb.get()};
    const Fp xs[2] = {ax, bx}, ys[2] = {ay, by};
    const bool infs[2] = {ainf, binf};
    return f12_is_one(final_exponentiation(miller_loop_multi(tabs, xs, ys, infs, 2)));
}

}  // namespace pairing
}  // namespace kzgamd
