// GLV scalar split on BLS12-381 G1, shared by the variable-base MSM (device) and fft_g1 (host, per root).
#pragma once
#include "ff.hip.h"

namespace kzgamd {
using ff::u32;
using ff::u64;

// GLV split: k = +-k1 +- k2 * X2 with X2 = x^2 (x the BLS parameter, X2 ~ 2^127.4), so
// k*P = +-k1*P +- k2*[x^2]P with [x^2]P = (beta*x, -y) one field multiplication away.  Half as many windows, and
// the Horner chain over the window sums is ~112 doublings instead of 255.
//   1. k > (r-1)/2 -> use r - k and flip both signs            (k <= (r-1)/2)
//   2. q = floor(k / X2), rem = k - q*X2: division by the constant through its reciprocal
//      M = floor(2^256 / X2), at most two corrections
//   3. rem > X2/2 -> rem = X2 - rem (negative), q += 1
// Both halves end below 2^126.5, so ceil(128/c) signed windows never carry out of the top one.
// Branch-free (selects only): lanes of a wave take different paths through the corrections, and the compare loops
// with early exits of a first version cost ~2500 instructions per call against ~300 for this form.
FF_HD void glv_split(const u32 kin[8], u32 k1[8], u32 k2[8], u32& neg1, u32& neg2) {
    constexpr u32 X2[5] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u, 0u};
    constexpr u32 X2H[4] = {0x80000000u, 0x00000000u, 0x8000d201u, 0x5622d200u};  // X2 / 2
    constexpr u32 M[5] = {0xf6cfee2eu, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x1u};
    constexpr u32 RH[8] = {0x80000000u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u,
                           0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};  // (r - 1) / 2
    // 1. flip = k > (r-1)/2  (borrow of (r-1)/2 - k);  k <- flip ? r - k : k
    u32 k[8];
    u32 flip;
    {
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u64 v = (u64)RH[i] - kin[i] - bw;
            bw = (u32)(v >> 63);
        }
        flip = bw;
        bw = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u64 v = (u64)ff::FrParams::p(i) - kin[i] - bw;
            k[i] = flip ? (u32)v : kin[i];
            bw = (u32)(v >> 63);
        }
    }
    // 2. q = floor(k * M / 2^256): operand scanning, 8 x 5 words
    u32 t[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 carry = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const u64 v = (u64)k[i] * M[j] + t[i + j] + carry;
            t[i + j] = (u32)v;
            carry = (u32)(v >> 32);
        }
        t[i + 5] = carry;
    }
    u32 q[4] = {t[8], t[9], t[10], t[11]};
    // 3. rem = k - q * X2 on 160 bits (rem < 3 * X2)
    u32 pr[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32 carry = 0;
#pragma unroll
        for (int j = 1; j < 4; ++j) {  // X2[0] == 0
            if (i + j < 5) {
                const u64 v = (u64)q[i] * X2[j] + pr[i + j] + carry;
                pr[i + j] = (u32)v;
                carry = (u32)(v >> 32);
            }
        }
        if (i + 4 < 5) pr[i + 4] = carry;
    }
    u32 rem[5];
    {
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const u64 v = (u64)k[i] - pr[i] - bw;
            rem[i] = (u32)v;
            bw = (u32)(v >> 63);
        }
    }
    // 4. at most two corrections: rem >= X2 -> rem -= X2, q += 1
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        u32 d[5];
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const u64 v = (u64)rem[i] - X2[i] - bw;
            d[i] = (u32)v;
            bw = (u32)(v >> 63);
        }
        const u32 ge = bw ^ 1u;
#pragma unroll
        for (int i = 0; i < 5; ++i) rem[i] = ge ? d[i] : rem[i];
        u32 cy = ge;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u64 v = (u64)q[i] + cy;
            q[i] = (u32)v;
            cy = (u32)(v >> 32);
        }
    }
    // 5. balance the remainder: rem > X2/2  ->  X2 - rem (negative), q += 1
    u32 big;
    {
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u64 v = (u64)X2H[i] - rem[i] - bw;
            bw = (u32)(v >> 63);
        }
        big = bw;
        u32 d[4];
        bw = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u64 v = (u64)X2[i] - rem[i] - bw;
            d[i] = (u32)v;
            bw = (u32)(v >> 63);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) rem[i] = big ? d[i] : rem[i];
        u32 cy = big;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u64 v = (u64)q[i] + cy;
            q[i] = (u32)v;
            cy = (u32)(v >> 32);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        k1[i] = rem[i];
        k2[i] = q[i];
        k1[i + 4] = 0;
        k2[i + 4] = 0;
    }
    neg1 = big ^ flip;
    neg2 = flip;
}

}  // namespace kzgamd
