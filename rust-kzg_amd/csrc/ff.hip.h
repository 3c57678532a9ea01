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

// The same inverse with the multiword work done once per 30 halving steps (the batched binary GCD of T. Pornin,
// "Optimized Binary GCD for Modular Inversion", 2020, in its variable-time form): 30 steps of the binary GCD run on
// 62-bit approximations (the low 30 and the top 32 bits of A and B) and record the linear map
// (A, B) -> ((f0 A + g0 B) / 2^30, (f1 A + g1 B) / 2^30), |f| + |g| <= 2^30, which is then applied exactly to A, B and
// — with the division by 2^30 done modulo p, Montgomery fashion — to U, V (A = y U, B = y V mod p throughout).  A wrong
// comparison on the approximations only costs a sign, fixed after the exact update.  ~17 outer steps for Fr, ~26 for
// Fp, instead of ~500 / ~760 multiword shift-and-subtract steps: the one-lane inversions of the quotient and
// compression kernels are latency chains.  Ends with B = gcd = 1; anything else (never observed) falls back to
// inverse_plain_bgcd, and tests/test_host_cpu.py compares the two on random and edge values.
template <class P>
FF_HD Field<P> inverse_plain_fast(const Field<P>& y) {
    
    constexpr int K = 30;
    typedef Field<P> F;
    typedef int64_t i64;
    if (y.is_zero()) return F::zero();
    u32 A[P::N], B[P::N], U[P::N], V[P::N];
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        A[i] = y.v[i];
        B[i] = P::p(i);
        U[i] = 0;
        V[i] = 0;
    }
    U[0] = 1;
    const u32 minv = P::M0 & ((1u << K) - 1u);  // -p^-1 mod 2^30
    auto bitlen = [](const u32* x) {
        int l = 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i)
            if (x[i]) l = 32 * i + 32 - __builtin_clz(x[i]);
        return l;
    };
    // bits [lo, lo + 32) of x, lo >= 0
    auto window32 = [](const u32* x, int lo) -> u32 {
        const int w = lo >> 5, s = lo & 31;
        u32 a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i) {
            if (i == w) a = x[i];
            if (i == w + 1) b = x[i];
        }
        return s ? (a >> s) | (b << (32 - s)) : a;
    };
    // x = (f x + g z) / 2^30 exactly; returns true if the result was negative (then it is negated)
    auto lincomb_exact = [](u32* x, const u32* xs, const u32* zs, i64 f, i64 g) -> bool {
        u32 t[P::N + 1];
        i64 acc = 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i) {
            acc += f * (i64)(u64)xs[i] + g * (i64)(u64)zs[i];
            t[i] = (u32)acc;
            acc >>= 32;
        }
        t[P::N] = (u32)acc;
        const bool neg = acc < 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i) x[i] = (t[i] >> K) | (t[i + 1] << (32 - K));
        if (neg) {
            u64 c = 1;
#pragma unroll
            for (int i = 0; i < P::N; ++i) {
                c += (u64)(u32)~x[i];
                x[i] = (u32)c;
                c >>= 32;
            }
        }
        return neg;
    };
    // x = (f xs + g zs) / 2^30 mod p, in [0, p), for xs, zs in [0, p)
    auto lincomb_mod = [minv](u32* x, const u32* xs, const u32* zs, i64 f, i64 g) {
        const u32 t0 = (u32)((u64)f * xs[0] + (u64)g * zs[0]);
        const i64 q = (i64)((t0 * minv) & ((1u << K) - 1u));
        u32 t[P::N + 1];
        i64 acc = 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i) {
            acc += f * (i64)(u64)xs[i] + g * (i64)(u64)zs[i] + q * (i64)(u64)P::p(i);
            t[i] = (u32)acc;
            acc >>= 32;
        }
        t[P::N] = (u32)acc;
        const bool neg = acc < 0;  // value in (-p, 2p) after the shift
#pragma unroll
        for (int i = 0; i < P::N; ++i) x[i] = (t[i] >> K) | (t[i + 1] << (32 - K));
        if (neg) {
            u64 c = 0;
#pragma unroll
            for (int i = 0; i < P::N; ++i) {
                c += (u64)x[i] + P::p(i);
                x[i] = (u32)c;
                c >>= 32;
            }
        } else {
            u32 d[P::N];
            u64 b = 0;
#pragma unroll
            for (int i = 0; i < P::N; ++i) {
                const u64 e = (u64)x[i] - P::p(i) - b;
                d[i] = (u32)e;
                b = (e >> 32) & 1;
            }
            // bit 2 of the word above the top limb tells 2^(32N) apart from the borrow: the shifted value's limb P::N
            const u32 top = (u32)(t[P::N] >> K) & 3u;  // 0 or 1: the value's bits above 32N (p may fill its top limb)
            if (top || !b) {
#pragma unroll
                for (int i = 0; i < P::N; ++i) x[i] = d[i];
            }
        }
    };
    int pbits = 0;
    {
        u32 pm[P::N];
#pragma unroll
        for (int i = 0; i < P::N; ++i) pm[i] = P::p(i);
        pbits = bitlen(pm);
    }
    const int max_outer = (2 * pbits - 1 + K - 1) / K + 2;
    for (int it = 0; it < max_outer; ++it) {
        u32 nz = 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i) nz |= A[i];
        if (!nz) break;
        const int la = bitlen(A), lb = bitlen(B);
        int n = la > lb ? la : lb;
        if (n < 2 * K + 2) n = 2 * K + 2;
        u64 xa = (u64)(A[0] & ((1u << K) - 1u)) | ((u64)window32(A, n - 32) << K);
        u64 xb = (u64)(B[0] & ((1u << K) - 1u)) | ((u64)window32(B, n - 32) << K);
        i64 f0 = 1, g0 = 0, f1 = 0, g1 = 1;
        for (int i = 0; i < K; ++i) {
            if (xa & 1) {
                if (xa < xb) {
                    const u64 tx = xa;
                    xa = xb;
                    xb = tx;
                    i64 tf = f0;
                    f0 = f1;
                    f1 = tf;
                    tf = g0;
                    g0 = g1;
                    g1 = tf;
                }
                xa -= xb;
                f0 -= f1;
                g0 -= g1;
            }
            xa >>= 1;
            f1 <<= 1;
            g1 <<= 1;
        }
        u32 nA[P::N], nB[P::N], nU[P::N], nV[P::N];
        if (lincomb_exact(nA, A, B, f0, g0)) {
            f0 = -f0;
            g0 = -g0;
        }
        if (lincomb_exact(nB, A, B, f1, g1)) {
            f1 = -f1;
            g1 = -g1;
        }
        lincomb_mod(nU, U, V, f0, g0);
        lincomb_mod(nV, U, V, f1, g1);
#pragma unroll
        for (int i = 0; i < P::N; ++i) {
            A[i] = nA[i];
            B[i] = nB[i];
            U[i] = nU[i];
            V[i] = nV[i];
        }
    }
    u32 bad = B[0] ^ 1u;
#pragma unroll
    for (int i = 0; i < P::N; ++i) bad |= (i ? B[i] : 0u) | A[i];
    if (bad) {
#ifdef FF_INV_COUNT_FALLBACK
        ++FF_INV_COUNT_FALLBACK;
#endif
        return inverse_plain_bgcd(y);
    }
    F r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = V[i];
    return r;
}

// inverse in Montgomery form: (a*R)^-1 as a residue is a^-1 * R^-1; two multiplications by R^2 give a^-1 * R
template <class P>
FF_HD Field<P> inverse_bgcd(const Field<P>& a_mont) {
    Field<P> r = inverse_plain_fast(a_mont);
    const Field<P> r2 = Field<P>::r2();
    return mul(mul(r, r2), r2);
}

typedef Field<FpParams> Fp;
typedef Field<FrParams> Fr;

}  // namespace ff
