// BLS12-381 prime fields for gfx950 (and for the host side of this library).
//
// Fp  : 381-bit base field, 12 x u32 little-endian limbs, Montgomery R = 2^384
// Fr  : 255-bit scalar field, 8 x u32 little-endian limbs, Montgomery R = 2^256
//
// The in-memory layout is bit-identical to blst's `blst_fp {u64 l[6]}` /
// `blst_fr {u64 l[4]}` (reference: kzg/src/eth/c_bindings.rs:429-450), so device
// buffers can be filled straight from the caller's arrays.
//
// The multiplier is written around v_mad_u64_u32 (32x32+64 -> 64), the widest
// integer multiply-add CDNA4 has: every `(u64)a * b + c` below becomes exactly
// one of them.  No MFMA: this is wide-integer modular arithmetic.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FF_HD __host__ __device__ __forceinline__
#else
#define FF_HD inline
#endif

namespace ff {

typedef uint32_t u32;
typedef uint64_t u64;

struct FpParams {
    static constexpr int N = 12;
    static constexpr u32 M0 = 0xfffcfffdu;  // -p^-1 mod 2^32
    FF_HD static constexpr u32 p(int i) {
        constexpr u32 t[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                               0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return t[i];
    }
    FF_HD static constexpr u32 one(int i) {  // 2^384 mod p
        constexpr u32 t[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                               0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return t[i];
    }
    FF_HD static constexpr u32 r2(int i) {  // 2^768 mod p
        constexpr u32 t[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                               0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return t[i];
    }
};

struct FrParams {
    static constexpr int N = 8;
    static constexpr u32 M0 = 0xffffffffu;  // -r^-1 mod 2^32
    FF_HD static constexpr u32 p(int i) {
        constexpr u32 t[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                              0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return t[i];
    }
    FF_HD static constexpr u32 one(int i) {  // 2^256 mod r
        constexpr u32 t[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                              0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return t[i];
    }
    FF_HD static constexpr u32 r2(int i) {  // 2^512 mod r
        constexpr u32 t[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                              0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return t[i];
    }
};

template <class P>
struct Field {
    static constexpr int N = P::N;
    u32 v[N];

    FF_HD static Field zero() {
        Field r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    FF_HD static Field one() {
        Field r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = P::one(i);
        return r;
    }
    FF_HD static Field r2() {
        Field r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = P::r2(i);
        return r;
    }
    FF_HD static Field modulus() {
        Field r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = P::p(i);
        return r;
    }
    FF_HD bool is_zero() const {
        u32 acc = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) acc |= v[i];
        return acc == 0;
    }
    FF_HD bool operator==(const Field& o) const {
        u32 acc = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) acc |= v[i] ^ o.v[i];
        return acc == 0;
    }
    FF_HD bool operator!=(const Field& o) const { return !(*this == o); }
};

// r = a - p if a >= p else a   (a < 2p)
template <class P>
FF_HD void reduce_once(Field<P>& a) {
    constexpr int N = P::N;
    u32 t[N];
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        u64 d = (u64)a.v[i] - P::p(i) - borrow;
        t[i] = (u32)d;
        borrow = (d >> 32) & 1;
    }
    if (!borrow) {
#pragma unroll
        for (int i = 0; i < N; ++i) a.v[i] = t[i];
    }
}

template <class P>
FF_HD Field<P> add(const Field<P>& a, const Field<P>& b) {
    constexpr int N = P::N;
    Field<P> r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        c += (u64)a.v[i] + b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    // both moduli leave the top limb's high bit(s) free, so no carry out of limb N-1
    reduce_once(r);
    return r;
}

template <class P>
FF_HD Field<P> sub(const Field<P>& a, const Field<P>& b) {
    constexpr int N = P::N;
    Field<P> r;
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        u64 d = (u64)a.v[i] - b.v[i] - borrow;
        r.v[i] = (u32)d;
        borrow = (d >> 32) & 1;
    }
    u32 mask = (u32)0 - (u32)borrow;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        c += (u64)r.v[i] + (P::p(i) & mask);
        r.v[i] = (u32)c;
        c >>= 32;
    }
    return r;
}

template <class P>
FF_HD Field<P> neg(const Field<P>& a) {
    constexpr int N = P::N;
    Field<P> r;
    u64 borrow = 0;
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        u64 d = (u64)P::p(i) - a.v[i] - borrow;
        r.v[i] = (u32)d;
        borrow = (d >> 32) & 1;
        nz |= a.v[i];
    }
    u32 mask = nz ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] &= mask;
    return r;
}

// conditional negate
template <class P>
FF_HD Field<P> cneg(const Field<P>& a, bool flag) {
    Field<P> n = neg(a);
    Field<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = flag ? n.v[i] : a.v[i];
    return r;
}

template <class P>
FF_HD Field<P> dbl(const Field<P>& a) {
    return add(a, a);
}

// Montgomery product a*b*R^-1 mod p.  CIOS, one row of the product and one
// row of the reduction per outer step; every inner statement is one
// v_mad_u64_u32.  Because p < 2^(32N-1) the running value stays below 2p and
// fits N+1 limbs.
template <class P>
FF_HD Field<P> mul(const Field<P>& a, const Field<P>& b) {
    constexpr int N = P::N;
    u32 t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const u32 bi = b.v[i];
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            c = (u64)a.v[j] * bi + t[j] + c;
            t[j] = (u32)c;
            c >>= 32;
        }
        c += t[N];
        t[N] = (u32)c;  // < 2 here, the carry-out is always zero
        const u32 m = t[0] * P::M0;
        c = (u64)m * P::p(0) + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            c = (u64)m * P::p(j) + t[j] + c;
            t[j - 1] = (u32)c;
            c >>= 32;
        }
        c += t[N];
        t[N - 1] = (u32)c;
        t[N] = (u32)(c >> 32);
    }
    Field<P> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
    reduce_once(r);
    return r;
}

template <class P>
FF_HD Field<P> sqr(const Field<P>& a) {
    return mul(a, a);
}

template <class P>
FF_HD Field<P> to_mont(const Field<P>& a) {
    return mul(a, Field<P>::r2());
}

template <class P>
FF_HD Field<P> from_mont(const Field<P>& a) {
    // multiply by 1: reduction rows only
    constexpr int N = P::N;
    u32 t[N + 1];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = a.v[i];
    t[N] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const u32 m = t[0] * P::M0;
        u64 c = (u64)m * P::p(0) + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            c = (u64)m * P::p(j) + t[j] + c;
            t[j - 1] = (u32)c;
            c >>= 32;
        }
        c += t[N];
        t[N - 1] = (u32)c;
        t[N] = (u32)(c >> 32);
    }
    Field<P> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
    reduce_once(r);
    return r;
}

// a^e for a little-endian u32 exponent array (not constant time; exponents are public)
template <class P>
FF_HD Field<P> pow_u32(const Field<P>& a, const u32* e, int nlimbs) {
    Field<P> r = Field<P>::one();
    bool started = false;
    for (int i = nlimbs - 1; i >= 0; --i) {
        for (int b = 31; b >= 0; --b) {
            if (started) r = sqr(r);
            if ((e[i] >> b) & 1) {
                r = started ? mul(r, a) : a;
                started = true;
            }
        }
    }
    return r;
}

// a^(p-2)
template <class P>
FF_HD Field<P> inverse(const Field<P>& a) {
    constexpr int N = P::N;
    u32 e[N];
    u64 borrow = 2;  // e = p - 2
#pragma unroll
    for (int i = 0; i < N; ++i) {
        u64 d = (u64)P::p(i) - borrow;
        e[i] = (u32)d;
        borrow = (d >> 32) & 1;
    }
    return pow_u32(a, e, N);
}

// Modular inverse of a PLAIN residue by the binary extended Euclid algorithm (right-shift variant):
// ~2*bits rounds of 32-bit-limb shifts / adds instead of the ~1.5*bits Montgomery multiplications of
// Fermat's a^(p-2) — about 10x fewer instructions, which matters where one lane inverts alone
// (affine conversion before compressing a commitment).  Not constant time; inputs are public.
// Returns 0 for a == 0.
template <class P>
FF_HD Field<P> inverse_plain_bgcd(const Field<P>& a) {
    constexpr int N = P::N;
    typedef Field<P> F;
    if (a.is_zero()) return F::zero();
    F u = a, v = F::modulus(), x1 = F::zero(), x2 = F::zero();
    x1.v[0] = 1;
    auto is_one = [](const F& x) {
        u32 acc = x.v[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < N; ++i) acc |= x.v[i];
        return acc == 0;
    };
    auto shr1 = [](F& x) {
#pragma unroll
        for (int i = 0; i < N - 1; ++i) x.v[i] = (x.v[i] >> 1) | (x.v[i + 1] << 31);
        x.v[N - 1] >>= 1;
    };
    auto add_p = [](F& x) {  // x += p (no overflow: x < p < 2^(32N-1))
        u64 c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            c += (u64)x.v[i] + P::p(i);
            x.v[i] = (u32)c;
            c >>= 32;
        }
    };
    auto sub_raw = [](F& x, const F& y) -> u32 {  // x -= y, returns borrow
        u64 b = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            u64 d = (u64)x.v[i] - y.v[i] - b;
            x.v[i] = (u32)d;
            b = (d >> 32) & 1;
        }
        return (u32)b;
    };
    auto halve_mod = [&](F& x) {  // x = x/2 mod p
        if (x.v[0] & 1) add_p(x);
        shr1(x);
    };
    while (!is_one(u) && !is_one(v)) {
        while (!(u.v[0] & 1)) {
            shr1(u);
            halve_mod(x1);
        }
        while (!(v.v[0] & 1)) {
            shr1(v);
            halve_mod(x2);
        }
        F t = u;
        if (!sub_raw(t, v)) {  // u >= v
            u = t;
            if (sub_raw(x1, x2)) add_p(x1);
        } else {
            sub_raw(v, u);
            if (sub_raw(x2, x1)) add_p(x2);
        }
    }
    return is_one(u) ? x1 : x2;
}

// inverse in Montgomery form: (a*R)^-1 as a residue is a^-1 * R^-1; two multiplications by R^2 give a^-1 * R
template <class P>
FF_HD Field<P> inverse_bgcd(const Field<P>& a_mont) {
    Field<P> r = inverse_plain_bgcd(a_mont);
    const Field<P> r2 = Field<P>::r2();
    return mul(mul(r, r2), r2);
}

typedef Field<FpParams> Fp;
typedef Field<FrParams> Fr;

}  // namespace ff
