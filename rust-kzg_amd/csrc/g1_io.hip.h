// G1 wire format on the device: ZCash-compressed 48-byte points <-> fp28 coordinates.
// Replaces what the reference gets from blst_p1_uncompress / blst_p1_compress through
// FsG1::from_bytes / to_bytes (blst/src/types/g1.rs:65-100); flag semantics as stated
// in-tree at zkcrypto/bls12_381/src/g1.rs:337-392.
#pragma once
#include "g1_28.hip.h"

namespace g1io {
using ff::u32;
using ff::u64;
using fp28::Fe;

FF_HD constexpr u32 r2_392_l(int i) {  // 2^784 mod p
    constexpr u32 t[14] = {0x10370edu, 0x6d1c345u, 0xe243d62u, 0xec45c53u, 0x3b1d65au, 0x93317du, 0xb4f36a0u,
                           0x5d74088u, 0xc10ea72u, 0x865d118u, 0x7320a75u, 0xfd5cd50u, 0xcc8a759u, 0xc8d4u};
    return t[i];
}
FF_HD constexpr u32 b4_392_l(int i) {  // 4 * 2^392 mod p  (curve constant b = 4)
    constexpr u32 t[14] = {0xd1ff2e0u, 0x6000000u, 0xac467u, 0x3379b48u, 0x1c84b80u, 0xe88243u, 0xdd9a7eu,
                           0x683dcf8u, 0x6c26d0bu, 0x4a5eec2u, 0x457663cu, 0x4b29f1u, 0x967f3e8u, 0x15de9u};
    return t[i];
}
FF_HD constexpr u32 p_half_l(int i) {  // (p-1)/2, saturated limbs
    constexpr u32 t[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                           0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
    return t[i];
}
FF_HD constexpr u32 p_sqrt_exp_l(int i) {  // (p+1)/4, saturated limbs
    constexpr u32 t[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                           0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
    return t[i];
}

// plain (non-Montgomery) canonical integer of a field element
FF_HD ff::Fp to_plain(const Fe& a) {
    Fe one_raw = fp28::zero();
    one_raw.v[0] = 1;
    ff::Fp s = fp28::pack(fp28::mul(a, one_raw));
    ff::reduce_once(s);
    return s;
}
// plain canonical integer (< p) -> Montgomery-392
FF_HD Fe from_plain(const ff::Fp& a) {
    Fe c;
#pragma unroll
    for (int i = 0; i < 14; ++i) c.v[i] = r2_392_l(i);
    return fp28::mul(fp28::unpack(a), c);
}

FF_HD bool sat_geq_p(const ff::Fp& a) {
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        u64 d = (u64)a.v[i] - ff::FpParams::p(i) - borrow;
        borrow = (d >> 32) & 1;
    }
    return borrow == 0;
}
FF_HD bool is_lex_largest(const ff::Fp& plain) {  // plain > (p-1)/2
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        u64 d = (u64)p_half_l(i) - plain.v[i] - borrow;
        borrow = (d >> 32) & 1;
    }
    return borrow != 0;
}

template <class ExpFn>
FF_HD Fe pow_sat(const Fe& a, ExpFn e) {
    Fe r = fp28::one();
    bool started = false;
    for (int i = 11; i >= 0; --i) {
        const u32 w = e(i);
        for (int b = 31; b >= 0; --b) {
            if (started) r = fp28::sqr(r);
            if ((w >> b) & 1) {
                r = started ? fp28::mul(r, a) : a;
                started = true;
            }
        }
    }
    return r;
}
FF_HD Fe inverse_fermat(const Fe& a) {
    return pow_sat(a, [](int i) -> u32 { return i == 0 ? ff::FpParams::p(0) - 2 : ff::FpParams::p(i); });
}
// a^-1 in the 2^392 Montgomery domain via the (batched) binary-Euclid inverse of the plain residue:
// (x*R')^-1 = x^-1 * R'^-1, then two multiplications by R'^2 give x^-1 * R'
FF_HD Fe inverse(const Fe& a) {
    ff::Fp s = fp28::pack(a);  // normalized, value < 2^384 (callers pass mul outputs, < 2p)
    ff::reduce_once(s);
    ff::reduce_once(s);
    const ff::Fp inv = ff::inverse_plain_fast(s);
    Fe c;
#pragma unroll
    for (int i = 0; i < 14; ++i) c.v[i] = r2_392_l(i);
    return fp28::mul(fp28::mul(fp28::unpack(inv), c), c);
}

// 48 big-endian bytes -> saturated plain integer
FF_HD ff::Fp be48_to_sat(const unsigned char* in) {
    ff::Fp r;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const unsigned char* q = in + (11 - i) * 4;
        r.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    return r;
}
FF_HD void sat_to_be48(unsigned char* out, const ff::Fp& a) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        unsigned char* q = out + (11 - i) * 4;
        q[0] = (unsigned char)(a.v[i] >> 24);
        q[1] = (unsigned char)(a.v[i] >> 16);
        q[2] = (unsigned char)(a.v[i] >> 8);
        q[3] = (unsigned char)a.v[i];
    }
}

// blst_p1_uncompress: returns false on an invalid encoding.  On-curve by construction,
// no subgroup check (same as FsG1::from_bytes).  Infinity -> flags bit0.
FF_HD bool uncompress(g1::AffPt& out, const unsigned char in[48]) {
    out.flags = 0;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
    out.x = fp28::zero();
    out.y = fp28::zero();
    const bool compressed = (in[0] >> 7) & 1, infinity = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    if (!compressed) return false;
    unsigned char tmp[48];
    for (int i = 0; i < 48; ++i) tmp[i] = in[i];
    tmp[0] &= 0x1f;
    ff::Fp xs = be48_to_sat(tmp);
    if (infinity) {
        if (sort || !xs.is_zero()) return false;
        out.flags = 1;
        return true;
    }
    if (sat_geq_p(xs)) return false;
    Fe x = from_plain(xs);
    Fe b4;
#pragma unroll
    for (int i = 0; i < 14; ++i) b4.v[i] = b4_392_l(i);
    Fe y2 = fp28::addn(fp28::mul(fp28::sqr(x), x), b4);  // x^3 + 4, < 4p
    Fe y = pow_sat(y2, [](int i) -> u32 { return p_sqrt_exp_l(i); });
    // is it a root?
    Fe chk = fp28::sub<8>(fp28::sqr(y), y2);
    if (!fp28::is_zero_mod_p(chk)) return false;
    y = fp28::canon(y);
    if (is_lex_largest(to_plain(y)) != sort) y = fp28::canon(fp28::neg<2>(y));
    out.x = fp28::canon(x);
    out.y = y;
    return true;
}

// blst_p1_compress of an XYZZ point, given zi = 1 / (ZZ * ZZZ)
FF_HD void compress_with_inverse(unsigned char out[48], const g1::Xyzz& p, const Fe& zi) {
    if (g1::is_inf(p)) {
        for (int i = 0; i < 48; ++i) out[i] = 0;
        out[0] = 0xc0;
        return;
    }
    Fe x = fp28::mul(p.x, fp28::mul(zi, p.zzz));
    Fe y = fp28::mul(p.y, fp28::mul(zi, p.zz));
    sat_to_be48(out, to_plain(x));
    out[0] |= 0x80;
    if (is_lex_largest(to_plain(y))) out[0] |= 0x20;
}
FF_HD void compress(unsigned char out[48], const g1::Xyzz& p) {
    Fe zi = fp28::one();
    if (!g1::is_inf(p)) zi = inverse(fp28::mul(p.zz, p.zzz));
    compress_with_inverse(out, p, zi);
}

}  // namespace g1io
