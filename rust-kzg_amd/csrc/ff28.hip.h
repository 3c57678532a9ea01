// Experimental unsaturated Fp: 14 limbs of 28 bits, Montgomery R = 2^392.
// Column sums of 28x28-bit products fit a 64-bit accumulator with no carry
// handling at all, so the multiplier is a pure v_mad_u64_u32 stream.
// Candidate measured by tools/ffbench.hip against the saturated 12x32 form.
#pragma once
#include "ff.hip.h"

namespace ff28 {
using ff::u32;
using ff::u64;

constexpr int L = 14;
constexpr u32 MASK = (1u << 28) - 1;
constexpr u32 P0INV = 0xffcfffdu;  // -p^-1 mod 2^28

FF_HD constexpr u32 p28(int i) {
    constexpr u32 t[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                           0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
    return t[i];
}

struct Fp28 {
    u32 v[L];
};

FF_HD Fp28 from_sat(const ff::Fp& a) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int bit = 28 * i, w = bit >> 5, s = bit & 31;
        u64 two = (u64)a.v[w] | ((w + 1 < 12) ? ((u64)a.v[w + 1] << 32) : 0);
        r.v[i] = (u32)(two >> s) & MASK;
    }
    return r;
}

FF_HD ff::Fp to_sat(const Fp28& a) {
    ff::Fp r;
#pragma unroll
    for (int w = 0; w < 12; ++w) {
        // bits [32w, 32w+32)
        const int lo = (32 * w) / 28, s = 32 * w - 28 * lo;
        u64 two = (u64)a.v[lo] | ((lo + 1 < L) ? ((u64)a.v[lo + 1] << 28) : 0);
        r.v[w] = (u32)(two >> s);
    }
    ff::reduce_once(r);
    return r;
}

// a*b*2^-392 mod p, output limbs < 2^28, value < 2p for inputs < 2^5 p.
FF_HD Fp28 mul(const Fp28& a, const Fp28& b) {
    u32 m[L];
    Fp28 r;
    u64 acc = 0, acc2 = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc2 += (u64)m[i] * p28(k - i);
        acc += acc2;
        acc2 = 0;
        m[k] = ((u32)acc * P0INV) & MASK;
        acc += (u64)m[k] * p28(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) acc2 += (u64)m[i] * p28(k - i);
        acc += acc2;
        acc2 = 0;
        r.v[k - L] = (u32)acc & MASK;
        acc >>= 28;
    }
    r.v[L - 1] = (u32)acc;
    return r;
}

}  // namespace ff28
