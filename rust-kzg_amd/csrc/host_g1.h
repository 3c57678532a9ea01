// Host-side G1 (de)serialisation on blst-layout values, for the few single-point conversions the
// c-kzg helpers need (compute_challenge takes a blst_p1; bytes_to_kzg_commitment returns one).
// Bulk work goes through the device kernels in g1_io.hip.h instead.
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/kzg_mi355x.h"
#include "ff.hip.h"
#include "host_fp64.h"

namespace kzgamd {

inline ff::Fp host_fp_from_be48(const uint8_t* in, bool* lt_p) {
    ff::Fp r;
    for (int i = 0; i < 12; ++i) {
        const uint8_t* q = in + (11 - i) * 4;
        r.v[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    uint64_t borrow = 0;
    for (int i = 0; i < 12; ++i) {
        uint64_t d = (uint64_t)r.v[i] - ff::FpParams::p(i) - borrow;
        borrow = (d >> 32) & 1;
    }
    *lt_p = borrow != 0;
    return r;
}

inline void host_fp_to_be48(uint8_t* out, const ff::Fp& plain) {
    for (int i = 0; i < 12; ++i) {
        uint8_t* q = out + (11 - i) * 4;
        q[0] = (uint8_t)(plain.v[i] >> 24);
        q[1] = (uint8_t)(plain.v[i] >> 16);
        q[2] = (uint8_t)(plain.v[i] >> 8);
        q[3] = (uint8_t)plain.v[i];
    }
}

inline bool host_fp_lex_largest(const ff::Fp& plain) {  // plain > (p-1)/2
    static const uint32_t half[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                      0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
    uint64_t borrow = 0;
    for (int i = 0; i < 12; ++i) {
        uint64_t d = (uint64_t)half[i] - plain.v[i] - borrow;
        borrow = (d >> 32) & 1;
    }
    return borrow != 0;
}

// blst_p1_compress
inline void host_p1_compress(uint8_t out[48], const blst_p1* p) {
    const ff::Fp* P = reinterpret_cast<const ff::Fp*>(p);
    if (P[2].is_zero()) {
        memset(out, 0, 48);
        out[0] = 0xc0;
        return;
    }
    ff::Fp zi = ff::inverse_bgcd(P[2]), zi2 = hfp::sqr(zi);
    ff::Fp x = hfp::from_mont(hfp::mul(P[0], zi2)), y = hfp::from_mont(hfp::mul(P[1], hfp::mul(zi2, zi)));
    host_fp_to_be48(out, x);
    out[0] |= 0x80;
    if (host_fp_lex_largest(y)) out[0] |= 0x20;
}

// blst_p1_compress of n Jacobian points with ONE field inversion (Montgomery's trick over the non-zero Z): the small
// batches of the concurrent-caller lanes compress here — the device's one-lane binary-Euclid inversion is ~250 us of
// latency per batch, this is ~10 us
inline void host_p1_compress_batch(uint8_t* out48, const blst_p1* jac, size_t n) {
    constexpr size_t CH = 64;
    for (size_t lo = 0; lo < n; lo += CH) {
        const size_t m = n - lo < CH ? n - lo : CH;
        ff::Fp pre[CH];
        ff::Fp acc = ff::Fp::one();
        for (size_t i = 0; i < m; ++i) {
            const ff::Fp* P = reinterpret_cast<const ff::Fp*>(&jac[lo + i]);
            pre[i] = acc;
            if (!P[2].is_zero()) acc = hfp::mul(acc, P[2]);
        }
        ff::Fp inv = ff::inverse_bgcd(acc);
        for (size_t i = m; i-- > 0;) {
            const ff::Fp* P = reinterpret_cast<const ff::Fp*>(&jac[lo + i]);
            uint8_t* out = out48 + 48 * (lo + i);
            if (P[2].is_zero()) {
                memset(out, 0, 48);
                out[0] = 0xc0;
                continue;
            }
            const ff::Fp zi = hfp::mul(inv, pre[i]), zi2 = hfp::sqr(zi);
            inv = hfp::mul(inv, P[2]);
            const ff::Fp x = hfp::from_mont(hfp::mul(P[0], zi2)), y = hfp::from_mont(hfp::mul(P[1], hfp::mul(zi2, zi)));
            host_fp_to_be48(out, x);
            out[0] |= 0x80;
            if (host_fp_lex_largest(y)) out[0] |= 0x20;
        }
    }
}

// blst_p1_uncompress + blst_p1_from_affine (FsG1::from_bytes, blst/src/types/g1.rs:65-87)
inline bool host_p1_uncompress(blst_p1* out, const uint8_t in[48]) {
    ff::Fp* O = reinterpret_cast<ff::Fp*>(out);
    const bool compressed = (in[0] >> 7) & 1, infinity = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    if (!compressed) return false;
    uint8_t tmp[48];
    memcpy(tmp, in, 48);
    tmp[0] &= 0x1f;
    bool lt = false;
    ff::Fp xs = host_fp_from_be48(tmp, &lt);
    if (infinity) {
        if (sort || !xs.is_zero()) return false;
        memset(out, 0, sizeof *out);
        return true;
    }
    if (!lt) return false;
    ff::Fp x = hfp::to_mont(xs);
    ff::Fp four = ff::Fp::zero();
    four.v[0] = 4;
    ff::Fp y2 = hfp::add(hfp::mul(hfp::sqr(x), x), hfp::to_mont(four));
    static const uint32_t e[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                   0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
    ff::Fp y = hfp::pow_u32(y2, e, 12);
    if (hfp::sqr(y) != y2) return false;
    if (host_fp_lex_largest(hfp::from_mont(y)) != sort) y = hfp::neg(y);
    O[0] = x;
    O[1] = y;
    O[2] = ff::Fp::one();
    return true;
}

// ---- host-side Jacobian arithmetic on blst-layout values (a handful of points per call at most) ----
struct HostJac {
    ff::Fp x, y, z;  // Montgomery; infinity <=> z == 0
};
inline HostJac host_jac_dbl(const HostJac& p) {  // dbl-2009-l (a = 0)
    if (p.z.is_zero()) return p;
    using ff::Fp;
    Fp A = hfp::sqr(p.x), B = hfp::sqr(p.y), C = hfp::sqr(B);
    Fp t = hfp::sub(hfp::sub(hfp::sqr(hfp::add(p.x, B)), A), C);
    Fp D = hfp::add(t, t), E = hfp::add(hfp::add(A, A), A), F = hfp::sqr(E);
    HostJac r;
    r.x = hfp::sub(hfp::sub(F, D), D);
    Fp C8 = hfp::dbl(hfp::dbl(hfp::dbl(C)));
    r.y = hfp::sub(hfp::mul(E, hfp::sub(D, r.x)), C8);
    r.z = hfp::dbl(hfp::mul(p.y, p.z));
    return r;
}
inline HostJac host_jac_add(const HostJac& a, const HostJac& b) {  // add-2007-bl with the exceptional cases
    if (a.z.is_zero()) return b;
    if (b.z.is_zero()) return a;
    using ff::Fp;
    Fp z1z1 = hfp::sqr(a.z), z2z2 = hfp::sqr(b.z);
    Fp u1 = hfp::mul(a.x, z2z2), u2 = hfp::mul(b.x, z1z1);
    Fp s1 = hfp::mul(hfp::mul(a.y, b.z), z2z2), s2 = hfp::mul(hfp::mul(b.y, a.z), z1z1);
    Fp h = hfp::sub(u2, u1), rr = hfp::sub(s2, s1);
    if (h.is_zero()) {
        if (rr.is_zero()) return host_jac_dbl(a);
        HostJac inf;
        inf.x = inf.y = inf.z = Fp::zero();
        return inf;
    }
    rr = hfp::dbl(rr);
    Fp i = hfp::sqr(hfp::dbl(h)), j = hfp::mul(h, i), v = hfp::mul(u1, i);
    HostJac r;
    r.x = hfp::sub(hfp::sub(hfp::sub(hfp::sqr(rr), j), v), v);
    r.y = hfp::sub(hfp::mul(rr, hfp::sub(v, r.x)), hfp::dbl(hfp::mul(s1, j)));
    r.z = hfp::mul(hfp::sub(hfp::sub(hfp::sqr(hfp::add(a.z, b.z)), z1z1), z2z2), h);
    return r;
}
inline HostJac host_jac_mul_u64(const HostJac& p, uint64_t k) {
    HostJac acc;
    acc.x = acc.y = acc.z = ff::Fp::zero();
    for (int bit = 63; bit >= 0; --bit) {
        acc = host_jac_dbl(acc);
        if ((k >> bit) & 1) acc = host_jac_add(acc, p);
    }
    return acc;
}
// blst_p1_in_g1 by the endomorphism test phi(P) == -[x^2]P (see k_check_commitments in ckzg.hip)
inline bool host_p1_in_g1(const blst_p1* pt) {
    const ff::Fp* P = reinterpret_cast<const ff::Fp*>(pt);
    if (P[2].is_zero()) return true;
    HostJac p{P[0], P[1], P[2]};
    const uint64_t BLS_X = 0xd201000000010000ull;
    HostJac q = host_jac_mul_u64(host_jac_mul_u64(p, BLS_X), BLS_X);
    if (q.z.is_zero()) return false;
    // beta (cube root of unity), plain: 0x5f19672f...fffefffe
    static const uint32_t beta_plain[12] = {0xfffefffeu, 0x2e01ffffu, 0x620a0002u, 0xde17d813u, 0xe6f89688u, 0xddb3a93bu,
                                            0x6a0f77eau, 0xba69c607u, 0xdf76ce51u, 0x5f19672fu, 0x00000000u, 0x00000000u};
    ff::Fp beta;
    for (int i = 0; i < 12; ++i) beta.v[i] = beta_plain[i];
    beta = hfp::to_mont(beta);
    HostJac e{hfp::mul(p.x, beta), p.y, p.z};  // phi(P) in Jacobian form (x scales by beta, Z unchanged)
    q.y = hfp::neg(q.y);
    // projective equality
    ff::Fp z1z1 = hfp::sqr(e.z), z2z2 = hfp::sqr(q.z);
    if (hfp::mul(e.x, z2z2) != hfp::mul(q.x, z1z1)) return false;
    return hfp::mul(e.y, hfp::mul(z2z2, q.z)) == hfp::mul(q.y, hfp::mul(z1z1, e.z));
}

}  // namespace kzgamd
