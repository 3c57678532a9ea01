// Fr for the NTT butterflies: 9 limbs of 29 bits, Montgomery radix R' = 2^261 — the same idea as
// fp28.hip.h (carry-free columns of v_mad_u64_u32: 18 partial products < 2^58 per 64-bit column).
//
// Data elements keep blst's Montgomery form d*2^256 (only re-sliced 32 -> 29 bits); twiddles are
// stored as w*2^261, so  mul(D, W) = d*w*2^256  stays in blst form without any domain conversion.
// Butterfly outputs are lazy (value < 64r, limbs renormalised each stage); a pass ends with one
// multiplication by 2^261 mod r (or by the inverse-transform scale) that brings the value below 2r.
#pragma once
#include "ff.hip.h"

namespace fr29 {
using ff::u32;
using ff::u64;

constexpr int L = 9;
constexpr u32 MASK = (1u << 29) - 1;
constexpr u32 R0INV = 0x1fffffffu;  // -r^-1 mod 2^29

struct Fe {
    u32 v[L];
};

FF_HD constexpr u32 rl(int i) {
    constexpr u32 t[L] = {0x1u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0xc0404d0u, 0x1520cce7u, 0xa6533afu, 0x73eda7u};
    return t[i];
}
FF_HD constexpr u32 one_l(int i) {  // 2^261 mod r
    constexpr u32 t[L] = {0x1fffffbau, 0x22fu, 0x1cb61180u, 0xa4e5c00u, 0xee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x853ddf7u, 0x4d043fu};
    return t[i];
}
FF_HD constexpr u32 pad4_l(int i) {  // 4r with limbs 0..7 >= 2^29 - 1
    constexpr u32 t[L] = {0x20000004u, 0x3fffffdfu, 0x3e5bfefeu, 0x2d2017feu, 0x360154eeu, 0x30101342u, 0x3483339cu, 0x2994cebdu, 0x1cfb69cu};
    return t[i];
}

FF_HD constexpr u32 pad8_l(int i) {  // 8r, same shape
    constexpr u32 t[L] = {0x20000008u, 0x3fffffbfu, 0x3cb7fdfeu, 0x3a402ffeu, 0x2c02a9ddu, 0x20202686u, 0x2906673au, 0x33299d7cu, 0x39f6d39u};
    return t[i];
}

FF_HD Fe one() {
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = one_l(i);
    return r;
}

FF_HD void norm(Fe& a) {
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        a.v[i + 1] += a.v[i] >> 29;
        a.v[i] &= MASK;
    }
}

// x + t and x + 4r - t  (t normalized, value < 3r), both renormalised
FF_HD void butterfly(Fe& x, Fe& y_out, const Fe& t) {
    Fe s, d;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        s.v[i] = x.v[i] + t.v[i];
        d.v[i] = x.v[i] + pad4_l(i) - t.v[i];
    }
    norm(s);
    norm(d);
    x = s;
    y_out = d;
}

// the same without the carry passes: limbs grow by < 2^29 (x + t) resp. < 2^30 (x + 4r - t) per call; a round of
// butterflies normalises after every second stage (mul wants multiplicand limbs < 2^31)
FF_HD void butterfly_lazy(Fe& x, Fe& y_out, const Fe& t) {
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const u32 xi = x.v[i];
        x.v[i] = xi + t.v[i];
        y_out.v[i] = xi + pad4_l(i) - t.v[i];
    }
}

// x + t and x + 8r - t for a normalized t with value < 7r (a data element used as its own product by w^0 = 1)
FF_HD void butterfly_lazy8(Fe& x, Fe& y_out, const Fe& t) {
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const u32 xi = x.v[i];
        x.v[i] = xi + t.v[i];
        y_out.v[i] = xi + pad8_l(i) - t.v[i];
    }
}

// a*b*2^-261 mod r; limbs of a < 2^31, of b < 2^29; a*b < 2^261 * r; output normalized, < 2r
FF_HD Fe mul(const Fe& a, const Fe& b) {
    u32 m[L];
    Fe r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        u64 acc2 = 0;
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc2 += (u64)m[i] * rl(k - i);
        acc += acc2;
        m[k] = ((u32)acc * R0INV) & MASK;
        acc += (u64)m[k] * rl(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
        u64 acc2 = 0;
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) acc2 += (u64)m[i] * rl(k - i);
        acc += acc2;
        r.v[k - L] = (u32)acc & MASK;
        acc >>= 29;
    }
    r.v[L - 1] = (u32)acc;
    return r;
}

// ---- the butterfly multiplier of the NTT: subtractive Montgomery steps, one accumulator chain ----
// With r = 1 mod 2^29 the quotient digit of a column is the column's low 29 bits themselves (m = acc mod 2^29) if
// m*r is SUBTRACTED instead of added: (acc - m) >> 29 is a plain arithmetic shift, no negation and no carry fix-up
// (two instructions per column instead of four).  The products m_i * (-r_j) are signed multiply-adds
// (v_mad_i64_i32) on the same 64-bit accumulator as the unsigned a_i * b_j.  CHAIN keeps every column on one
// accumulator: left to itself the compiler splits a column into several chains and pays a 64-bit addition per merge
// (~40 per multiplication inside the NTT kernel).
//   The result is congruent to a*b*2^-261 and lies in (-r, r) for a < 64r, b < r: limbs 0..7 normalised, the top limb
//   SIGNED (two's complement in the u32).  Limbs 0..7 of a <= 1.5 * 2^30, of b < 2^29: every column stays below 2^63
//   in absolute value.  a's top limb is taken as signed too (a lazy sum x + t + r can have top limb -1 with the
//   carries of the lower limbs still pending): its nine products are signed multiply-adds.
// `acc += a * b` as ONE accumulator chain: the empty asm makes the compiler treat every step's result as opaque, so
// it cannot re-associate the column into several chains (each merge costs a 64-bit addition); the multiply-add is
// still pattern-matched to a single v_mad_u64_u32 / v_mad_i64_i32.
template <bool CHAIN>
FF_HD void chain_step(u64& acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (CHAIN) asm("" : "+v"(acc));
#endif
}
template <bool CHAIN>
FF_HD void mad_uu(u64& acc, u32 a, u32 b) {  // unsigned x unsigned
    acc += (u64)a * b;
    chain_step<CHAIN>(acc);
}
template <bool CHAIN>
FF_HD void mad_ii(u64& acc, u32 a, int b) {  // signed x signed
    acc += (u64)((long long)(int)a * (long long)b);
    chain_step<CHAIN>(acc);
}
template <bool CHAIN = true>
FF_HD Fe mul_signed(const Fe& a, const Fe& b) {
    u32 m[L];
    Fe r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            if (i == L - 1) mad_ii<CHAIN>(acc, a.v[i], (int)b.v[k - i]);
            else mad_uu<CHAIN>(acc, a.v[i], b.v[k - i]);
        }
#pragma unroll
        for (int i = 0; i < k; ++i) mad_ii<CHAIN>(acc, m[i], -(int)rl(k - i));
        m[k] = (u32)acc & MASK;
        acc = (u64)((long long)acc >> 29);  // (acc - m[k] * r_0) / 2^29, r_0 = 1
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) {
            if (i == L - 1) mad_ii<CHAIN>(acc, a.v[i], (int)b.v[k - i]);
            else mad_uu<CHAIN>(acc, a.v[i], b.v[k - i]);
        }
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) mad_ii<CHAIN>(acc, m[i], -(int)rl(k - i));
        r.v[k - L] = (u32)acc & MASK;
        acc = (u64)((long long)acc >> 29);
    }
    r.v[L - 1] = (u32)acc;
    return r;
}

// Two products side by side: the multiply-adds of r0 = a0 * b0 and r1 = a1 * b1 alternate, each on its own accumulator
// chain.  Same instructions as two calls of mul_signed, but no multiply-add follows the one it depends on: back to back,
// a dependent v_mad_u64_u32 needs a wait state (the compiler's s_nop 0 after every one of them), and at four waves per
// SIMD such streams issue a multiply-add per 6.2 cycles where two interleaved chains issue one per 4.4
// (tools/lone_wave_issue.hip, profiles/r06_lone_wave_issue.log).
template <bool CHAIN = true>
FF_HD void mul_signed2(Fe& r0, Fe& r1, const Fe& a0, const Fe& b0, const Fe& a1, const Fe& b1) {
    u32 m0[L], m1[L];
    u64 acc0 = 0, acc1 = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            if (i == L - 1) {
                mad_ii<CHAIN>(acc0, a0.v[i], (int)b0.v[k - i]);
                mad_ii<CHAIN>(acc1, a1.v[i], (int)b1.v[k - i]);
            } else {
                mad_uu<CHAIN>(acc0, a0.v[i], b0.v[k - i]);
                mad_uu<CHAIN>(acc1, a1.v[i], b1.v[k - i]);
            }
        }
#pragma unroll
        for (int i = 0; i < k; ++i) {
            mad_ii<CHAIN>(acc0, m0[i], -(int)rl(k - i));
            mad_ii<CHAIN>(acc1, m1[i], -(int)rl(k - i));
        }
        m0[k] = (u32)acc0 & MASK;
        m1[k] = (u32)acc1 & MASK;
        acc0 = (u64)((long long)acc0 >> 29);
        acc1 = (u64)((long long)acc1 >> 29);
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) {
            if (i == L - 1) {
                mad_ii<CHAIN>(acc0, a0.v[i], (int)b0.v[k - i]);
                mad_ii<CHAIN>(acc1, a1.v[i], (int)b1.v[k - i]);
            } else {
                mad_uu<CHAIN>(acc0, a0.v[i], b0.v[k - i]);
                mad_uu<CHAIN>(acc1, a1.v[i], b1.v[k - i]);
            }
        }
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) {
            mad_ii<CHAIN>(acc0, m0[i], -(int)rl(k - i));
            mad_ii<CHAIN>(acc1, m1[i], -(int)rl(k - i));
        }
        r0.v[k - L] = (u32)acc0 & MASK;
        r1.v[k - L] = (u32)acc1 & MASK;
        acc0 = (u64)((long long)acc0 >> 29);
        acc1 = (u64)((long long)acc1 >> 29);
    }
    r0.v[L - 1] = (u32)acc0;
    r1.v[L - 1] = (u32)acc1;
}

// x + t + r  and  x + 4r - t  for t = mul_signed(..) in (-r, 2r): the +r that makes the first one positive rides in a
// three-operand addition, the second is |4r_i - t_i| + x_i in one instruction for the normalised limbs (4r's limbs
// 0..7 are >= 2^29 - 1 >= t_i) and an ordinary add/sub for the signed top limb.  Limbs grow by < 2^30 per call, the
// value by < 5r.
FF_HD void butterfly_signed(Fe& x, Fe& y_out, const Fe& t) {
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const u32 xi = x.v[i], ti = t.v[i];
        x.v[i] = xi + ti + rl(i);
        // |4r_i - t_i| + x_i with 4r_i >= t_i: written so that it is selected as one v_sad_u32
        if (i < L - 1) y_out.v[i] = (pad4_l(i) > ti ? pad4_l(i) - ti : ti - pad4_l(i)) + xi;
        else y_out.v[i] = xi + pad4_l(i) - ti;
    }
}

// bit re-slicing 8 x 32 <-> 9 x 29
FF_HD Fe unpack(const ff::Fr& a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        u64 two = (u64)a.v[w] | ((w + 1 < 8) ? ((u64)a.v[w + 1] << 32) : 0);
        r.v[i] = (u32)(two >> s) & MASK;
    }
    return r;
}
FF_HD ff::Fr pack(const Fe& a) {  // normalized, value < 2^256
    ff::Fr r;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int lo = (32 * w) / 29, s = 32 * w - 29 * lo;
        u64 two = (u64)a.v[lo] | ((lo + 1 < L) ? ((u64)a.v[lo + 1] << 29) : 0);
        r.v[w] = (u32)(two >> s);
    }
    return r;
}
// a * 32 as 9 x 29-bit limbs (bit re-slicing with the 5-bit shift folded in; value < 2^261)
FF_HD Fe unpack_shl5(const ff::Fr& a) {
    Fe r;
    r.v[0] = (a.v[0] << 5) & MASK;
#pragma unroll
    for (int i = 1; i < L; ++i) {
        const int bit = 29 * i - 5, w = bit >> 5, s = bit & 31;
        u64 two = (u64)a.v[w] | ((w + 1 < 8) ? ((u64)a.v[w + 1] << 32) : 0);
        r.v[i] = (u32)(two >> s) & MASK;
    }
    return r;
}
// ff::mul for blst_fr operands (a * b * 2^-256 mod r, result in [0, r)) through the 29-bit multiplier:
// (32 a) * b * 2^-261.  Same value as ff::mul bit for bit, in about half the instructions (the 8 x 32-bit CIOS form
// pays a 64-bit add and a pile of moves per multiply-add).  Operands below 2^256 with a * b < 2^256 * r.
FF_HD ff::Fr mul_blst(const ff::Fr& a, const ff::Fr& b) {
    ff::Fr r = pack(mul(unpack_shl5(a), unpack(b)));
    ff::reduce_once(r);
    return r;
}

// lazy value (< 64r) times a normalized canonical multiplier (W = w*2^261), fully reduced to [0, r)
FF_HD ff::Fr finish(const Fe& a, const Fe& mult) {
    ff::Fr r = pack(mul(a, mult));
    ff::reduce_once(r);
    return r;
}

// lazy value (normalized limbs, value < 64r) -> canonical [0, r), without a multiplication: the quotient
// q = floor(value / r) <= 63 is estimated from the top limb (value >> 232 against (r >> 232) + 1 by a reciprocal
// multiplication; the estimate is q or q - 1), q*r is subtracted limb-wise with signed carries, and one conditional
// subtraction finishes.  ~85 instructions against ~330 for finish().
FF_HD ff::Fr reduce_lazy(const Fe& a) {
    // 2^40 / ((r >> 232) + 1) = 2^40 / 7597480 = 144719.03
    const u32 q = (u32)(((u64)a.v[L - 1] * 144719ull) >> 40);
    Fe d;
    long long c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        c += (long long)a.v[i] - (long long)((u64)q * rl(i));
        d.v[i] = (u32)c & MASK;
        c >>= 29;
    }
    c += (long long)a.v[L - 1] - (long long)((u64)q * rl(L - 1));
    d.v[L - 1] = (u32)c;  // value < 2r: fits
    ff::Fr r = pack(d);
    ff::reduce_once(r);
    return r;
}

}  // namespace fr29
