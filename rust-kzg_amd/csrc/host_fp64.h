// Host-side Fp for the pairing and the G1 / G2 point work the c-kzg surface does on the CPU (decompression, subgroup
// checks, the verify_* pairing check): the same values and the same memory layout as ff::Fp (Montgomery radix 2^384,
// canonical residues, twelve little-endian 32-bit words = six 64-bit words on the little-endian host), multiplied with
// 64 x 64 -> 128-bit products.  ff::mul is written for the GPU's 32-bit multiplier; on a CPU core this form is ~5x
// faster (75 ns per multiplication where ff::mul takes ~350).
#pragma once
#include <stdint.h>
#include <string.h>

#include "ff.hip.h"

namespace hfp {
using ff::Fp;
typedef unsigned __int128 u128;

struct W6 {
    uint64_t v[6];
};
inline W6 load(const Fp& a) {
    W6 r;
    memcpy(r.v, a.v, 48);
    return r;
}
inline Fp store(const W6& a) {
    Fp r;
    memcpy(r.v, a.v, 48);
    return r;
}
constexpr uint64_t P64[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                             0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
constexpr uint64_t N0 = 0x89f3fffcfffcfffdull;  // -p^-1 mod 2^64

// r = a - p if a >= p (a < 2p, `carry` = bit 384 of a)
inline void reduce_once(uint64_t a[6], uint64_t carry) {
    uint64_t d[6];
    u128 b = 0;
    for (int i = 0; i < 6; ++i) {
        const u128 t = (u128)a[i] - P64[i] - (uint64_t)b;
        d[i] = (uint64_t)t;
        b = (t >> 64) & 1;
    }
    if (carry || !(uint64_t)b)
        for (int i = 0; i < 6; ++i) a[i] = d[i];
}

inline Fp add(const Fp& x, const Fp& y) {
    W6 a = load(x);
    const W6 b = load(y);
    u128 c = 0;
    for (int i = 0; i < 6; ++i) {
        c += (u128)a.v[i] + b.v[i];
        a.v[i] = (uint64_t)c;
        c >>= 64;
    }
    reduce_once(a.v, (uint64_t)c);
    return store(a);
}
inline Fp dbl(const Fp& x) { return add(x, x); }
inline Fp sub(const Fp& x, const Fp& y) {
    W6 a = load(x);
    const W6 b = load(y);
    u128 bw = 0;
    for (int i = 0; i < 6; ++i) {
        const u128 t = (u128)a.v[i] - b.v[i] - (uint64_t)bw;
        a.v[i] = (uint64_t)t;
        bw = (t >> 64) & 1;
    }
    if ((uint64_t)bw) {
        u128 c = 0;
        for (int i = 0; i < 6; ++i) {
            c += (u128)a.v[i] + P64[i];
            a.v[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return store(a);
}
inline Fp neg(const Fp& x) { return x.is_zero() ? x : sub(Fp::zero(), x); }

// a * b * 2^-384 mod p.  p has a clear top bit, so the running sum of the operand-scanning Montgomery product never
// needs a seventh word ("no-carry" form): per word of b one multiply-add row of a and one of the modulus.
#define KZGAMD_HFP_MUL_BODY                                   \
    uint64_t a[6], b[6], t[6] = {0, 0, 0, 0, 0, 0};           \
    memcpy(a, x.v, 48);                                       \
    memcpy(b, y.v, 48);                                       \
    for (int i = 0; i < 6; ++i) {                             \
        u128 A = (u128)a[0] * b[i] + t[0];                    \
        const uint64_t m = (uint64_t)A * N0;                  \
        u128 C = (u128)m * P64[0] + (uint64_t)A;              \
        A >>= 64;                                             \
        C >>= 64;                                             \
        for (int j = 1; j < 6; ++j) {                         \
            A += (u128)a[j] * b[i] + t[j];                    \
            C += (u128)m * P64[j] + (uint64_t)A;              \
            t[j - 1] = (uint64_t)C;                           \
            A >>= 64;                                         \
            C >>= 64;                                         \
        }                                                     \
        t[5] = (uint64_t)(C + A);                             \
    }                                                         \
    reduce_once(t, 0);                                        \
    Fp r;                                                     \
    memcpy(r.v, t, 48);                                       \
    return r;
inline Fp mul_generic(const Fp& x, const Fp& y) { KZGAMD_HFP_MUL_BODY }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// the same source compiled with mulx / adcx / adox where the host has them (30 % faster than the baseline x86-64
// code: 65 against 95 ns); chosen once per process
__attribute__((target("bmi2,adx"), noinline)) inline Fp mul_adx(const Fp& x, const Fp& y) { KZGAMD_HFP_MUL_BODY }
inline bool have_adx() {
    static const bool v = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
    return v;
}
inline Fp mul(const Fp& x, const Fp& y) { return have_adx() ? mul_adx(x, y) : mul_generic(x, y); }
#else
inline Fp mul(const Fp& x, const Fp& y) { return mul_generic(x, y); }
#endif
#undef KZGAMD_HFP_MUL_BODY
inline Fp sqr(const Fp& x) { return mul(x, x); }
inline Fp to_mont(const Fp& plain) { return mul(plain, Fp::r2()); }
inline Fp from_mont(const Fp& m) {
    Fp one_plain = Fp::zero();
    one_plain.v[0] = 1;
    return mul(m, one_plain);
}
// a^e, e = nlimbs little-endian 32-bit words (square and multiply from the top bit)
inline Fp pow_u32(const Fp& a, const uint32_t* e, int nlimbs) {
    Fp r = Fp::one();
    bool started = false;
    for (int i = nlimbs * 32 - 1; i >= 0; --i) {
        if (started) r = sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            r = started ? mul(r, a) : a;
            started = true;
        }
    }
    return r;
}

}  // namespace hfp
