// Device-side Fp for the MSM kernels: 14 limbs of 28 bits ("unsaturated"),
// Montgomery radix R' = 2^392.
//
// Why not blst's 12 x 32 layout on the device: gfx950 issues v_mad_u64_u32 at
// the rate of any other VALU instruction (one per SIMD per 4 cycles by the SQ
// counters; tools/ffbench.hip sees 1.3x the wall time of a plain add because the
// chip clocks lower under a multiply-heavy stream), so the cheapest multiplier
// is the one with the fewest instructions overall, not the fewest multiplies.  With 28-bit limbs a whole
// column of partial products (<= 28 terms of < 2^58) fits a 64-bit accumulator,
// so the 381-bit Montgomery product is a carry-free stream of 392 mads plus
// ~100 shifts/masks (measured 58 G mul/s on MI355X against 35 G mul/s for the
// saturated CIOS form, profiles/r01_ffbench.log).
//
// Representation rules ("normalized" = every limb but the top one < 2^28):
//   mul/sqr   inputs: limbs < 2^30, value product < 2^392 * p
//             output: normalized, value < 2p
//   add       lazy limb-wise sum (not normalized)
//   sub<K>    a + K*p - b, b normalized with value < (K-1)*p; output normalized
// Values are only ever reduced to [0,p) at the I/O boundary (to_blst) and inside
// the rare exact zero test.
#pragma once
#include "ff.hip.h"

namespace fp28 {
using ff::u32;
using ff::u64;

constexpr int L = 14;
constexpr u32 MASK = (1u << 28) - 1;
constexpr u32 P0INV = 0xffcfffdu;     // -p^-1 mod 2^28
constexpr u32 P0INV_POS = 0x0030003u;  //  p^-1 mod 2^28

struct Fe {
    u32 v[L];
};

FF_HD constexpr u32 pl(int i) {  // p
    constexpr u32 t[L] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                          0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
    return t[i];
}
FF_HD constexpr u32 one_l(int i) {  // 2^392 mod p
    constexpr u32 t[L] = {0x347fcb8u, 0xd800000u, 0x2b119u, 0xcde6d2u, 0xc7212e0u, 0x83a2090u, 0x37669fu,
                          0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x12ca7cu, 0x659fcfau, 0x577au};
    return t[i];
}
FF_HD constexpr u32 to_blst_l(int i) {  // 2^384 mod p : x*2^392 -> x*2^384
    constexpr u32 t[L] = {0x2fffdu, 0x900000u, 0xc000276u, 0xbc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u,
                          0x7052574u, 0x7ce5853u, 0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x15f65u};
    return t[i];
}
FF_HD constexpr u32 from_blst_l(int i) {  // 2^400 mod p : x*2^384 -> x*2^392
    constexpr u32 t[L] = {0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u,
                          0xa4e6fe9u, 0x3e8a053u, 0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x156dau};
    return t[i];
}
// K*p written so that limbs 0..12 are >= 2^28-1 (one unit borrowed from the limb above)
template <int K>
FF_HD constexpr u32 pad_l(int i) {
    static_assert(K == 2 || K == 4 || K == 8 || K == 16 || K == 32, "pad multiple");
    constexpr u32 t2[L] = {0x1fff5556u, 0x1fdffffeu, 0x17ffff72u, 0x1fffd629u, 0x1c483d56u, 0x141ed61du, 0x1ece61a4u,
                           0x1e70a256u, 0x18ee9708u, 0x19759aebu, 0x174f6c85u, 0x1cd34962u, 0x13d472feu, 0x34021u};
    constexpr u32 t4[L] = {0x1ffeaaacu, 0x1fbffffeu, 0x1ffffee6u, 0x1fffac53u, 0x18907aaeu, 0x183dac3cu, 0x1d9cc349u,
                           0x1ce144aeu, 0x11dd2e12u, 0x12eb35d8u, 0x1e9ed90cu, 0x19a692c5u, 0x17a8e5feu, 0x68043u};
    constexpr u32 t8[L] = {0x1ffd5558u, 0x1f7ffffeu, 0x1ffffdceu, 0x1fff58a8u, 0x1120f55eu, 0x107b587au, 0x1b398694u,
                           0x19c2895eu, 0x13ba5c26u, 0x15d66bb1u, 0x1d3db219u, 0x134d258cu, 0x1f51cbfeu, 0xd0087u};
    constexpr u32 t16[L] = {0x1ffaaab0u, 0x1efffffeu, 0x1ffffb9eu, 0x1ffeb152u, 0x1241eabeu, 0x10f6b0f5u, 0x16730d29u,
                            0x138512beu, 0x1774b84eu, 0x1bacd763u, 0x1a7b6433u, 0x169a4b1au, 0x1ea397fdu, 0x1a0110u};
    constexpr u32 t32[L] = {0x1ff55560u, 0x1dfffffeu, 0x1ffff73eu, 0x1ffd62a6u, 0x1483d57eu, 0x11ed61ebu, 0x1ce61a53u,
                            0x170a257du, 0x1ee9709du, 0x1759aec7u, 0x14f6c868u, 0x1d349636u, 0x1d472ffbu, 0x340222u};
    return K == 2 ? t2[i] : K == 4 ? t4[i] : K == 8 ? t8[i] : K == 16 ? t16[i] : t32[i];
}

FF_HD Fe zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = 0;
    return r;
}
FF_HD Fe one() {
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = one_l(i);
    return r;
}
FF_HD bool is_zero_limbs(const Fe& a) {  // exact all-zero representation (used for the infinity flag)
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) acc |= a.v[i];
    return acc == 0;
}

// carry-propagate: limbs 0..12 < 2^28 afterwards
FF_HD void norm(Fe& a) {
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        a.v[i + 1] += a.v[i] >> 28;
        a.v[i] &= MASK;
    }
}

FF_HD Fe add(const Fe& a, const Fe& b) {  // lazy
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
FF_HD Fe addn(const Fe& a, const Fe& b) {
    Fe r = add(a, b);
    norm(r);
    return r;
}

template <int K>
FF_HD Fe sub(const Fe& a, const Fe& b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = a.v[i] + pad_l<K>(i) - b.v[i];
    norm(r);
    return r;
}

// a + K*p - b without the carry pass: limbs < 2^28 + 2^29 (still a valid mul/sqr operand)
template <int K>
FF_HD Fe sub_lazy(const Fe& a, const Fe& b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = a.v[i] + pad_l<K>(i) - b.v[i];
    return r;
}

// K*p - a  (a normalized, value < (K-1)p)
template <int K>
FF_HD Fe neg(const Fe& a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = pad_l<K>(i) - a.v[i];
    norm(r);
    return r;
}

// -DFP28_ONE_CHAIN: every column on ONE accumulator chain.  The empty asm after a multiply-add makes its result
// opaque, so the compiler cannot re-associate a column into several chains (each merge is a 64-bit addition,
// 4.4 issue cycles); the multiply-add is still selected as one v_mad_u64_u32.
FF_HD void chain_step(u64& acc) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(FP28_ONE_CHAIN)
    asm("" : "+v"(acc));
#else
    (void)acc;
#endif
}

// a*b*2^-392 mod p.  Product scanning with the Montgomery quotient digits
// folded into the same column accumulators; two accumulators keep two mad
// chains in flight.
FF_HD Fe mul_inline(const Fe& a, const Fe& b) {
    u32 m[L];
    Fe r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#if defined(FP28_ONE_CHAIN)
        u64& acc2 = acc;
#else
        u64 acc2 = 0;
#endif
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            chain_step(acc);
        }
#pragma unroll
        for (int i = 0; i < k; ++i) {
            acc2 += (u64)m[i] * pl(k - i);
            chain_step(acc2);
        }
#if !defined(FP28_ONE_CHAIN)
        acc += acc2;
#endif
        m[k] = ((u32)acc * P0INV) & MASK;
        acc += (u64)m[k] * pl(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
#if defined(FP28_ONE_CHAIN)
        u64& acc2 = acc;
#else
        u64 acc2 = 0;
#endif
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            chain_step(acc);
        }
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) {
            acc2 += (u64)m[i] * pl(k - i);
            chain_step(acc2);
        }
#if !defined(FP28_ONE_CHAIN)
        acc += acc2;
#endif
        r.v[k - L] = (u32)acc & MASK;
        acc >>= 28;
    }
    r.v[L - 1] = (u32)acc;
    return r;
}

// a^2: the off-diagonal products are taken once against 2a.
FF_HD Fe sqr_inline(const Fe& a) {
    u32 m[L], a2[L];
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) a2[i] = a.v[i] << 1;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#if defined(FP28_ONE_CHAIN)
        u64& acc2 = acc;
#else
        u64 acc2 = 0;
#endif
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) {
            acc += (u64)a2[i] * a.v[k - i];
            chain_step(acc);
        }
        if ((k & 1) == 0) {
            acc += (u64)a.v[k / 2] * a.v[k / 2];
            chain_step(acc);
        }
#pragma unroll
        for (int i = 0; i < k; ++i) {
            acc2 += (u64)m[i] * pl(k - i);
            chain_step(acc2);
        }
#if !defined(FP28_ONE_CHAIN)
        acc += acc2;
#endif
        m[k] = ((u32)acc * P0INV) & MASK;
        acc += (u64)m[k] * pl(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
#if defined(FP28_ONE_CHAIN)
        u64& acc2 = acc;
#else
        u64 acc2 = 0;
#endif
#pragma unroll
        for (int i = k - L + 1; 2 * i < k; ++i) {
            acc += (u64)a2[i] * a.v[k - i];
            chain_step(acc);
        }
        if ((k & 1) == 0) {
            acc += (u64)a.v[k / 2] * a.v[k / 2];
            chain_step(acc);
        }
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) {
            acc2 += (u64)m[i] * pl(k - i);
            chain_step(acc2);
        }
#if !defined(FP28_ONE_CHAIN)
        acc += acc2;
#endif
        r.v[k - L] = (u32)acc & MASK;
        acc >>= 28;
    }
    r.v[L - 1] = (u32)acc;
    return r;
}

// mul/sqr are inlined by default.  -DFP28_OUTLINE_MUL makes them real device functions (operands in
// VGPRs as 14-wide vectors, no scratch traffic): ~8x smaller kernels, but the argument marshalling
// (~60 v_mov per call) costs 5 % in the accumulation kernel and 9 % end to end (A/B on MI355X:
// 83.3k vs 76.0k commitments/s), and the fully inlined ~75 KB loop body showed no instruction-cache
// penalty.
#if defined(__HIP_DEVICE_COMPILE__) && defined(FP28_OUTLINE_MUL)
typedef u32 fe_vec __attribute__((ext_vector_type(14)));
__device__ __forceinline__ fe_vec to_vec(const Fe& a) {
    fe_vec r;
#pragma unroll
    for (int i = 0; i < L; ++i) r[i] = a.v[i];
    return r;
}
__device__ __forceinline__ Fe from_vec(fe_vec a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = a[i];
    return r;
}
static __device__ __noinline__ fe_vec mul_call(fe_vec a, fe_vec b) { return to_vec(mul_inline(from_vec(a), from_vec(b))); }
static __device__ __noinline__ fe_vec sqr_call(fe_vec a) { return to_vec(sqr_inline(from_vec(a))); }
__device__ __forceinline__ Fe mul(const Fe& a, const Fe& b) { return from_vec(mul_call(to_vec(a), to_vec(b))); }
__device__ __forceinline__ Fe sqr(const Fe& a) { return from_vec(sqr_call(to_vec(a))); }
#else
FF_HD Fe mul(const Fe& a, const Fe& b) { return mul_inline(a, b); }
FF_HD Fe sqr(const Fe& a) { return sqr_inline(a); }
#endif

// exact test a == 0 (mod p) for normalized a with value < 64p.
// If a = k*p then k = a_0 * p_0^-1 mod 2^28 must be < 64: a 3-instruction filter that
// rejects all but 64/2^28 of the non-zero values; the exact compare runs only then.
FF_HD bool is_zero_mod_p(const Fe& a_in) {
    // the low 28 bits of limb 0 are the value mod 2^28 whether or not the limbs are normalized
    const u32 k = (a_in.v[0] * P0INV_POS) & MASK;
    // -DKZGAMD_FORCE_EXACT_TESTS (build.py: libkzg_mi355x_exact.so, test infrastructure): no filter, every call runs
    // the exact comparison below — the branch 2.4e-7 of the values take becomes the one every value takes.  For
    // k >= 64 the comparison is against k*p >= 64p > a, so the answer is the same.
#if !defined(KZGAMD_FORCE_EXACT_TESTS)
    if (k >= 64) return false;
#endif
    Fe a = a_in;
    norm(a);
    u64 c = 0;
    u32 diff = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        c += (u64)k * pl(i);
        diff |= ((u32)c & MASK) ^ a.v[i];
        c >>= 28;
    }
    c += (u64)k * pl(L - 1);
    diff |= (u32)c ^ a.v[L - 1];
    return diff == 0;
}

// (a*b + c*d) * 2^-392 mod p with ONE reduction: both products share the column accumulators.
// Limb bounds: a*b and c*d columns together must stay below 2^63 (e.g. one lazy operand per product).
FF_HD Fe mul2_inline(const Fe& a, const Fe& b, const Fe& c, const Fe& d) {
    u32 m[L];
    Fe r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        u64 acc2 = 0, acc3 = 0;
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i <= k; ++i) acc3 += (u64)c.v[i] * d.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc2 += (u64)m[i] * pl(k - i);
        acc += acc2 + acc3;
        m[k] = ((u32)acc * P0INV) & MASK;
        acc += (u64)m[k] * pl(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
        u64 acc2 = 0, acc3 = 0;
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) acc3 += (u64)c.v[i] * d.v[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) acc2 += (u64)m[i] * pl(k - i);
        acc += acc2 + acc3;
        r.v[k - L] = (u32)acc & MASK;
        acc >>= 28;
    }
    r.v[L - 1] = (u32)acc;
    return r;
}

// ---- boundary conversions (blst layout: 12 x u32 saturated, Montgomery 2^384) ----
FF_HD Fe unpack(const ff::Fp& a) {  // bit re-slicing only
    Fe r;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int bit = 28 * i, w = bit >> 5, s = bit & 31;
        u64 two = (u64)a.v[w] | ((w + 1 < 12) ? ((u64)a.v[w + 1] << 32) : 0);
        r.v[i] = (u32)(two >> s) & MASK;
    }
    return r;
}
FF_HD ff::Fp pack(const Fe& a) {  // normalized, value < 2^384
    ff::Fp r;
#pragma unroll
    for (int w = 0; w < 12; ++w) {
        const int lo = (32 * w) / 28, s = 32 * w - 28 * lo;
        u64 two = (u64)a.v[lo] | ((lo + 1 < L) ? ((u64)a.v[lo + 1] << 28) : 0);
        r.v[w] = (u32)(two >> s);
    }
    return r;
}
// canonical residue of a normalized value < 2p
FF_HD Fe canon(const Fe& a) {
    ff::Fp s = pack(a);
    ff::reduce_once(s);
    return unpack(s);
}
FF_HD Fe from_blst(const ff::Fp& a) {
    Fe c;
#pragma unroll
    for (int i = 0; i < L; ++i) c.v[i] = from_blst_l(i);
    return mul(unpack(a), c);
}
FF_HD ff::Fp to_blst(const Fe& a) {  // canonical output in [0,p)
    Fe c;
#pragma unroll
    for (int i = 0; i < L; ++i) c.v[i] = to_blst_l(i);
    ff::Fp r = pack(mul(a, c));
    ff::reduce_once(r);
    return r;
}

}  // namespace fp28
