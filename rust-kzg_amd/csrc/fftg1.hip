// G1-valued radix-2 transform (B2, widening row): replaces FFTG1::fft_g1 for FsFFTSettings
// (blst/src/fft_g1.rs:13-83).  Same contract: natural order in and out, roots taken with stride
// max_width/n from roots_of_unity / reverse_roots_of_unity, inverse scaled by n^-1.
//
// The reference recurses (even/odd split) and spends its time in FsG1::mul, a 255-bit scalar
// multiplication per butterfly: n/2 * log2(n) of them.  Here the same butterfly network runs iteratively,
// one kernel per stage, TWO lanes per butterfly (all transforms of a batch in one launch).  The root w^j is split
// on the host, once per settings object, into w^j = +-k1 +- k2 * x^2 with 127-bit halves (glv.hip.h), so
//     lane 0:  a = k1 * (+-y)          lane 1:  b = k2 * (+-[x^2]y),  [x^2](X, Y, ZZZ, ZZ) = (beta*X, -Y, ZZZ, ZZ)
//              fixed 4-bit windows, the 15 multiples in a per-lane table in HBM
//              (32 windows: 124 doublings + <= 32 additions + 14 for the table, XYZZ coordinates)
//     t = a + b (exchanged by lane shuffles),   lane 0 writes x + t,  lane 1 writes x - t
// — half the dependent chain of the plain 255-bit double-and-add, on twice the lanes (a stage of a 2^15-point
// transform has 16 384 butterflies: a quarter of the chip's lanes).
// Points live in HBM as XYZZ over the 14 x 28-bit field (g1_28.hip.h) between stages; blst Jacobian at the
// boundary.  The work is integer-VALU bound; HBM traffic is negligible next to it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// -DKZGAMD_FFTG1_OUTLINE_MUL: fp28::mul / sqr as real functions in this file (a window step shrinks from 60 - 130 KB of
// code to under 20 KB).  Measured, not adopted: the long-lane kernels run one or two waves per SIMD and might have been
// bound by instruction fetch (64 KB instruction cache per two CUs) — they are not: FK20 of 256 blobs 23.9 -> 27.6 ms,
// one lane per half-butterfly 31.6 -> 37.8 ms (the argument marshalling of the calls), DESIGN.md §9.
#ifdef KZGAMD_FFTG1_OUTLINE_MUL
#define FP28_OUTLINE_MUL 1
#endif
#include "../../include/kzg_mi355x.h"
#include "ff.hip.h"
#include "g1_28.hip.h"
#include "g1w.hip.h"
#include "glv.hip.h"
#include "device_guard.h"
#include "ntt_internal.h"

using ff::Fr;
using ff::u32;
using ff::u64;
using g1::Xyzz;

namespace {

constexpr int WIN = 4;              // scalar window
constexpr int NTAB = (1 << WIN) - 1;  // multiples 1..15
constexpr int NTHREADS = 64;        // one wave per workgroup: long serial lanes, spread over all CUs
constexpr int MAXTHREADS = 256;     // ... or four waves per workgroup once there is a wave for every SIMD (block_threads)
constexpr int BOUND = 512;          // launch bound of the long-lane kernels: two waves per SIMD must fit (<= 256 registers)

// one root (or n^-1) split for the two lanes of a butterfly: 127-bit magnitudes and their signs
struct RootSplit {
    u32 k[2][4];
    u32 neg[2];
    u32 pad[2];
};
static_assert(sizeof(RootSplit) == 48, "RootSplit");

__device__ __forceinline__ u32 brev(u32 v, int bits) { return bits == 0 ? 0u : __builtin_bitreverse32(v) >> (32 - bits); }

// The point routines are kept out of line here: a lane executes ~170 of them per butterfly, and inlining each
// (20-35 KB of code apiece) buys nothing against their ~6000-instruction bodies.
__device__ __noinline__ void pt_dbl(Xyzz& a) {
    if (!g1::is_inf(a)) g1::dbl(a);
}
__device__ __noinline__ void pt_add(Xyzz& a, const Xyzz& b) { g1::dadd(a, b); }

__device__ __forceinline__ fp28::Fe beta28() {  // cube root of unity, Montgomery 2^392 (see msm.hip)
    constexpr u32 t[14] = {0xa75929au, 0x681b798u, 0x22a3e9du, 0xabc02bfu, 0x4e5bb45u, 0x55e6e7eu, 0x4814117u,
                           0x6d04f1bu, 0xae3387du, 0x54acb0cu, 0xa4c74bu, 0x56138b5u, 0xb64e066u, 0x76f2u};
    fp28::Fe b;
#pragma unroll
    for (int k = 0; k < 14; ++k) b.v[k] = t[k];
    return b;
}

// acc = k * acc for a 128-bit k (little-endian words); tab = this lane's 15 table slots, slot e at tab[e * stride]
__device__ void scalar_mul128(Xyzz& acc, const u32 k[4], Xyzz* tab, size_t tab_stride) {
    if (g1::is_inf(acc)) return;
    tab[0] = acc;
    Xyzz m = acc;
    pt_dbl(m);
    tab[tab_stride] = m;
    for (int e = 2; e < NTAB; ++e) {
        pt_add(m, acc);
        tab[(size_t)e * tab_stride] = m;
    }
    g1::set_inf(acc);
    for (int w = 128 / WIN - 1; w >= 0; --w) {
        if (!g1::is_inf(acc)) {
            pt_dbl(acc);
            pt_dbl(acc);
            pt_dbl(acc);
            pt_dbl(acc);
        }
        const u32 d = (k[w >> 3] >> ((w & 7) * WIN)) & (u32)NTAB;
        if (d) {
            Xyzz q = tab[(size_t)(d - 1) * tab_stride];
            pt_add(acc, q);
        }
    }
}

// this lane's half of  rs * p :  half 0 -> +-k1 * p,  half 1 -> +-k2 * [x^2]p;  then the sum of both halves,
// exchanged with the neighbouring lane (lanes 2m and 2m+1 hold the two halves of the same product)
__device__ void glv_mul_pair(Xyzz& p, const RootSplit& rs, int half, Xyzz* tab, size_t tab_stride) {
    if (!g1::is_inf(p)) {
        bool negate = rs.neg[half] != 0;
        if (half) {
            p.x = fp28::mul(p.x, beta28());
            negate = !negate;  // [x^2]p = (beta*X, -Y)
        }
        if (negate) p.y = fp28::neg<8>(p.y);
        scalar_mul128(p, rs.k[half], tab, tab_stride);
    }
    Xyzz other;
    u32* o = (u32*)&other;
    const u32* mine = (const u32*)&p;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(Xyzz) / 4); ++k) o[k] = __shfl_xor(mine[k], 1, 64);
    pt_add(p, other);
}

// blst Jacobian (X, Y, Z) -> XYZZ (X, Y, Z^2, Z^3), written at the bit-reversed position of its transform
__global__ void __launch_bounds__(256) k_g1_load(Xyzz* __restrict__ out, const ff::Fp* __restrict__ in, u32 n, int logn,
                                                 size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t xf = t / n;
    const u32 pos = (u32)(t % n);
    const ff::Fp* src = in + (xf * n + brev(pos, logn)) * 3;
    Xyzz p;
    if (src[2].is_zero()) {
        g1::set_inf(p);
    } else {
        const fp28::Fe z = fp28::from_blst(src[2]);
        p.x = fp28::from_blst(src[0]);
        p.y = fp28::from_blst(src[1]);
        p.zz = fp28::sqr(z);
        p.zzz = fp28::mul(p.zz, z);
    }
    out[t] = p;
}

// stage s of the DIT network on bit-reversed-order data: pairs i0 and i0 + 2^s,
// twiddle w_n^(j * n / 2^(s+1)) = roots[j * (W >> (s+1))]  (fft_g1.rs:22-29 unrolled); two lanes per butterfly
// (out of place: the two lanes of a butterfly both read x and y, so the stage writes to the other half of a
// ping-pong buffer instead of relying on the pair running in lockstep)
__global__ void __launch_bounds__(BOUND) k_g1_stage(Xyzz* __restrict__ dst, const Xyzz* __restrict__ data, Xyzz* __restrict__ tab,
                                                       const RootSplit* __restrict__ kroots, u32 n, int s, u32 W, int inverse,
                                                       size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // total = 2 * butterflies, even
    if (t >= total) return;
    const int half = (int)(t & 1);
    const size_t bf = t >> 1;
    const u32 halfn = n >> 1;
    const size_t xf = bf / halfn;
    const u32 b = (u32)(bf % halfn);
    const u32 hs = 1u << s, j = b & (hs - 1);
    const u32 i0 = ((b >> s) << (s + 1)) | j, i1 = i0 + hs;
    const Xyzz* base = data + xf * n;
    Xyzz y = base[i1];
    const u32 idx = j * (W >> (s + 1));
    if (idx != 0) glv_mul_pair(y, kroots[inverse ? W - idx : idx], half, tab + t, total);
    Xyzz x = base[i0];
    if (half) y.y = fp28::neg<8>(y.y);  // lane 1: x - t   (Y < 8p is within what dadd/dbl accept)
    pt_add(x, y);
    dst[xf * n + (half ? i1 : i0)] = x;
}

// XYZZ -> blst Jacobian; an inverse transform multiplies by n^-1 first (fft_g1.rs:72-79), two lanes per point
__global__ void __launch_bounds__(BOUND) k_g1_store(ff::Fp* __restrict__ out, const Xyzz* __restrict__ data,
                                                       Xyzz* __restrict__ tab, RootSplit inv_n, int scale, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // total = 2 * points
    if (t >= total) return;
    const int half = (int)(t & 1);
    Xyzz p = data[t >> 1];
    if (scale) glv_mul_pair(p, inv_n, half, tab + t, total);
    if (half == 0) g1::to_blst_jacobian(out + (t >> 1) * 3, p);
}

// device-resident form: XYZZ in natural order -> bit-reversed order of each transform
__global__ void __launch_bounds__(256) k_g1_brp_xyzz(Xyzz* __restrict__ out, const Xyzz* __restrict__ in, u32 n, int logn,
                                                     size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t xf = t / n;
    out[t] = in[xf * n + brev((u32)(t % n), logn)];
}
// data[p] *= n^-1 (two lanes per point, both in one wave: the loads of a pair precede its store)
__global__ void __launch_bounds__(BOUND) k_g1_scale_xyzz(Xyzz* __restrict__ data, Xyzz* __restrict__ tab, RootSplit inv_n,
                                                            size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // total = 2 * points
    if (t >= total) return;
    const int half = (int)(t & 1);
    Xyzz p = data[t >> 1];
    glv_mul_pair(p, inv_n, half, tab + t, total);
    if (half == 0) data[t >> 1] = p;
}

// ---------------------------------------------------------------- lane-group stages (mid-size grids)
// Between "a wave per half-butterfly" (limb-parallel, ~5x the instructions) and "a lane per half-butterfly" (the
// shortest instruction total, a 1.6 ms chain) sits a group of G = 2 or 4 neighbouring lanes per half-butterfly: every
// lane of the group keeps the whole point state and the single-lane multiplier, and the INDEPENDENT products of a point
// formula run side by side on the lanes of the group — each lane multiplies the operand pair of its role, the G results
// go round the group with DPP quad permutes.  A doubling is 5 / 3 multiplications deep instead of 9, an addition 7 / 4
// instead of 14; with signed 4-bit windows (8 multiples) a half-butterfly is ~890 / ~530 dependent multiplications
// instead of 1 760, on 2x / 4x the lanes.  Worth it while those lanes still find an idle SIMD: a stage with up to 2^15
// half-butterflies (256 blobs of FK20, a 2^15-point transform) runs on two lanes each, up to 2^14 on four.
namespace grp {
using fp28::Fe;

template <int CTRL>
__device__ __forceinline__ Fe dpp(const Fe& a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < fp28::L; ++i) {
        r.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)a.v[i], CTRL, 0xF, 0xF, true);
        // keep the permute an instruction of its own: folded into its consumer, an expression with TWO permuted
        // operands of the same register (t0 - t1 of a level's results) came out with one permute applied to both
        // (tools/grp_check.hip: Y3 = pad on role 0)
        asm("" : "+v"(r.v[i]));
    }
    return r;
}
// the value the lane of role K of this group holds (quad_perm: groups never straddle a quad)
template <int G, int K>
__device__ __forceinline__ Fe from_role(const Fe& a) {
    return dpp<G == 4 ? K * 0x55 : (K ? 0xF5 : 0xA0)>(a);
}
__device__ __forceinline__ Fe pick(bool c, const Fe& a, const Fe& b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < fp28::L; ++i) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}
__device__ __forceinline__ Fe pick4(int r, const Fe& a0, const Fe& a1, const Fe& a2, const Fe& a3) {
    return pick(r < 2, pick(r == 0, a0, a1), pick(r == 2, a2, a3));
}

template <int G>
__device__ void dbl(Xyzz& acc, int r);
template <int G>
__device__ void dadd(Xyzz& acc, const Xyzz& b, int r);

// dbl-2008-s-1 as in g1::dbl (same bounds): [V, M] [W, S] [MM, ZZ3] [M3*(S - X3), W*Y] [ZZZ3]
template <>
__device__ __noinline__ void dbl<2>(Xyzz& acc, int r) {
    using namespace fp28;
    if (g1::is_inf(acc)) return;
    const bool r0 = r == 0;
    const Fe u = addn(acc.y, acc.y);
    Fe t = mul(pick(r0, u, acc.x), pick(r0, u, acc.x));
    const Fe v = from_role<2, 0>(t), m = from_role<2, 1>(t);
    const Fe m3 = addn(add(m, m), m);
    t = mul(pick(r0, v, acc.x), pick(r0, u, v));
    const Fe w = from_role<2, 0>(t), s = from_role<2, 1>(t);
    t = mul(pick(r0, m3, acc.zz), pick(r0, m3, v));
    const Fe mm = from_role<2, 0>(t), zz3 = from_role<2, 1>(t);
    const Fe x3 = sub<8>(mm, addn(s, s));
    t = mul(pick(r0, m3, w), pick(r0, sub<16>(s, x3), acc.y));
    const Fe y3 = sub<4>(from_role<2, 0>(t), from_role<2, 1>(t));
    acc.zzz = mul(acc.zzz, w);  // the same product on both lanes
    acc.x = x3;
    acc.y = y3;
    acc.zz = zz3;
}
// [V, M, -, -] [W, S, ZZ3, MM] [M3*(S - X3), W*Y, ZZZ3, -]
template <>
__device__ __noinline__ void dbl<4>(Xyzz& acc, int r) {
    using namespace fp28;
    if (g1::is_inf(acc)) return;
    const Fe u = addn(acc.y, acc.y);
    const bool lo = (r & 1) == 0;
    Fe t = mul(pick(lo, u, acc.x), pick(lo, u, acc.x));  // roles 2, 3 repeat 0, 1
    const Fe v = from_role<4, 0>(t), m = from_role<4, 1>(t);
    const Fe m3 = addn(add(m, m), m);
    t = mul(pick4(r, v, acc.x, acc.zz, m3), pick4(r, u, v, v, m3));
    const Fe w = from_role<4, 0>(t), s = from_role<4, 1>(t), zz3 = from_role<4, 2>(t), mm = from_role<4, 3>(t);
    const Fe x3 = sub<8>(mm, addn(s, s));
    t = mul(pick4(r, m3, w, acc.zzz, acc.zzz), pick4(r, sub<16>(s, x3), acc.y, w, w));
    acc.y = sub<4>(from_role<4, 0>(t), from_role<4, 1>(t));
    acc.zzz = from_role<4, 2>(t);
    acc.x = x3;
    acc.zz = zz3;
}

// add-2008-s as in g1::dadd, exceptional cases included (every lane of a group holds the same values, so the group
// branches as one): [U, U2] [S, S2] [PP, RR] [PPP, Q] [ZZ12, ZZZ12] [R*(Q - X3), S*PPP] [ZZ3, ZZZ3]
template <>
__device__ __noinline__ void dadd<2>(Xyzz& acc, const Xyzz& b, int r) {
    using namespace fp28;
    if (g1::is_inf(b)) return;
    if (g1::is_inf(acc)) {
        acc = b;
        return;
    }
    const bool r0 = r == 0;
    Fe t = mul(pick(r0, acc.x, b.x), pick(r0, b.zz, acc.zz));
    const Fe u = from_role<2, 0>(t);
    const Fe p = sub<4>(from_role<2, 1>(t), u);
    t = mul(pick(r0, acc.y, b.y), pick(r0, b.zzz, acc.zzz));
    const Fe s = from_role<2, 0>(t);
    const Fe rr_ = sub<4>(from_role<2, 1>(t), s);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(rr_)) dbl<2>(acc, r);
        else g1::set_inf(acc);
        return;
    }
    t = mul(pick(r0, p, rr_), pick(r0, p, rr_));
    const Fe pp = from_role<2, 0>(t), rr = from_role<2, 1>(t);
    t = mul(pick(r0, p, u), pp);
    const Fe ppp = from_role<2, 0>(t), q = from_role<2, 1>(t);
    t = mul(pick(r0, acc.zz, acc.zzz), pick(r0, b.zz, b.zzz));
    const Fe zz12 = from_role<2, 0>(t), zzz12 = from_role<2, 1>(t);
    const Fe x3 = sub<8>(rr, addn(add(q, q), ppp));
    t = mul(pick(r0, rr_, s), pick(r0, sub<16>(q, x3), ppp));
    acc.y = sub<4>(from_role<2, 0>(t), from_role<2, 1>(t));
    t = mul(pick(r0, zz12, zzz12), pick(r0, pp, ppp));
    acc.zz = from_role<2, 0>(t);
    acc.zzz = from_role<2, 1>(t);
    acc.x = x3;
}
// [U, S, U2, S2] [PP, RR, ZZ12, ZZZ12] [PPP, Q, ZZ3, -] [R*(Q - X3), S*PPP, ZZZ3, -]
template <>
__device__ __noinline__ void dadd<4>(Xyzz& acc, const Xyzz& b, int r) {
    using namespace fp28;
    if (g1::is_inf(b)) return;
    if (g1::is_inf(acc)) {
        acc = b;
        return;
    }
    Fe t = mul(pick4(r, acc.x, acc.y, b.x, b.y), pick4(r, b.zz, b.zzz, acc.zz, acc.zzz));
    const Fe u = from_role<4, 0>(t), s = from_role<4, 1>(t);
    const Fe p = sub<4>(from_role<4, 2>(t), u), rr_ = sub<4>(from_role<4, 3>(t), s);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(rr_)) dbl<4>(acc, r);
        else g1::set_inf(acc);
        return;
    }
    t = mul(pick4(r, p, rr_, acc.zz, acc.zzz), pick4(r, p, rr_, b.zz, b.zzz));
    const Fe pp = from_role<4, 0>(t), rr = from_role<4, 1>(t), zz12 = from_role<4, 2>(t), zzz12 = from_role<4, 3>(t);
    t = mul(pick4(r, p, u, zz12, zz12), pp);
    const Fe ppp = from_role<4, 0>(t), q = from_role<4, 1>(t), zz3 = from_role<4, 2>(t);
    const Fe x3 = sub<8>(rr, addn(add(q, q), ppp));
    t = mul(pick4(r, rr_, s, zzz12, zzz12), pick4(r, sub<16>(q, x3), ppp, ppp, ppp));
    acc.y = sub<4>(from_role<4, 0>(t), from_role<4, 1>(t));
    acc.zzz = from_role<4, 2>(t);
    acc.x = x3;
    acc.zz = zz3;
}

// acc = k * acc for a 127-bit k, Booth digits in [-8, 8] (d_w = k[4w-1] + k[4w] + 2 k[4w+1] + 4 k[4w+2] - 8 k[4w+3]);
// tab = this group's 8 table slots (slot e at tab[e * stride]), written by role 0 and read by the whole group
template <int G>
__device__ void scalar_mul128(Xyzz& acc, const u32 k[4], Xyzz* tab, size_t tab_stride, int r) {
    if (g1::is_inf(acc)) return;
    if (r == 0) tab[0] = acc;
    Xyzz m = acc;
    dbl<G>(m, r);
    if (r == 0) tab[tab_stride] = m;
    for (int e = 2; e < 8; ++e) {
        dadd<G>(m, acc, r);
        if (r == 0) tab[(size_t)e * tab_stride] = m;
    }
    g1::set_inf(acc);
    for (int w = 31; w >= 0; --w) {
        if (!g1::is_inf(acc)) {
            dbl<G>(acc, r);
            dbl<G>(acc, r);
            dbl<G>(acc, r);
            dbl<G>(acc, r);
        }
        // the five bits k[4w+3 .. 4w-1]
        const int bit = 4 * w - 1;
        u32 v;
        if (bit < 0) v = (k[0] << 1) & 31u;
        else {
            const u64 two = ((u64)(bit / 32 + 1 < 4 ? k[bit / 32 + 1] : 0u) << 32) | k[bit / 32];
            v = (u32)(two >> (bit & 31)) & 31u;
        }
        const int d = (int)((v + 1) >> 1) - (int)((v >> 4) << 4);
        if (d != 0) {
            Xyzz q = tab[(size_t)((d < 0 ? -d : d) - 1) * tab_stride];
            if (d < 0) q.y = fp28::neg<8>(q.y);
            dadd<G>(acc, q, r);
        }
    }
}

// Booth digit w (4 bits, in [-8, 8]) of a 127-bit k
__device__ __forceinline__ int booth4(const u32 k[4], int w) {
    const int bit = 4 * w - 1;
    u32 v;
    if (bit < 0) v = (k[0] << 1) & 31u;
    else {
        const u64 two = ((u64)(bit / 32 + 1 < 4 ? k[bit / 32 + 1] : 0u) << 32) | k[bit / 32];
        v = (u32)(two >> (bit & 31)) & 31u;
    }
    return (int)((v + 1) >> 1) - (int)((v >> 4) << 4);
}

// acc = (+-k1) * acc + (+-k2) * [x^2]acc in ONE chain (Straus / Shamir): the doublings are shared by the two halves,
//   k * P = s1 k1 * P - s2 k2 * Q,   Q = (beta X, Y) = -[x^2]P,   d * Q = (beta X_d, Y_d) for the table entry d * P,
// so a window is four doublings and two additions (P-table, Q-table) — against the two-chains form (a lane group per
// half, the halves added at the end) the same number of additions and HALF the doublings, on half the lanes.
// tab: 16 slots of this butterfly (d * P at slot d - 1, d * Q at slot 8 + d - 1), written by role 0.
template <int G>
__device__ void scalar_mul_glv(Xyzz& acc, const RootSplit& rs, Xyzz* tab, size_t tab_stride, int r) {
    if (g1::is_inf(acc)) return;
    const fp28::Fe beta = beta28();
    Xyzz m = acc;
    for (int e = 0; e < 8; ++e) {
        if (e == 1) dbl<G>(m, r);
        else if (e > 1) dadd<G>(m, acc, r);
        Xyzz q = m;
        q.x = fp28::mul(m.x, beta);  // the same product on every lane of the group
        if (r == 0) {
            tab[(size_t)e * tab_stride] = m;
            tab[(size_t)(8 + e) * tab_stride] = q;
        }
    }
    const bool neg1 = rs.neg[0] != 0, neg2 = rs.neg[1] == 0;  // the Q half carries the minus of [x^2]P = -Q
    g1::set_inf(acc);
    for (int w = 31; w >= 0; --w) {
        if (!g1::is_inf(acc)) {
            dbl<G>(acc, r);
            dbl<G>(acc, r);
            dbl<G>(acc, r);
            dbl<G>(acc, r);
        }
        const int d1 = booth4(rs.k[0], w), d2 = booth4(rs.k[1], w);
        if (d1 != 0) {
            Xyzz q = tab[(size_t)((d1 < 0 ? -d1 : d1) - 1) * tab_stride];
            if ((d1 < 0) != neg1) q.y = fp28::neg<8>(q.y);
            dadd<G>(acc, q, r);
        }
        if (d2 != 0) {
            Xyzz q = tab[(size_t)(8 + (d2 < 0 ? -d2 : d2) - 1) * tab_stride];
            if ((d2 < 0) != neg2) q.y = fp28::neg<8>(q.y);
            dadd<G>(acc, q, r);
        }
    }
}

// the partner group's point (the other half of the same butterfly): lanes G apart
template <int G>
__device__ __forceinline__ Xyzz from_partner(const Xyzz& p) {
    Xyzz o;
    if (G == 2) {
        o.x = dpp<0x4E>(p.x);  // quad_perm [2, 3, 0, 1]
        o.y = dpp<0x4E>(p.y);
        o.zzz = dpp<0x4E>(p.zzz);
        o.zz = dpp<0x4E>(p.zz);
    } else {
        u32* d = (u32*)&o;
        const u32* sck = (const u32*)&p;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(Xyzz) / 4); ++i) d[i] = __shfl_xor(sck[i], G, 64);
    }
    return o;
}
}  // namespace grp

// stage s with G lanes per half-butterfly (see above); total = 2 * G * butterflies lanes
template <int G>
__global__ void __launch_bounds__(BOUND) k_g1_stage_grp(Xyzz* __restrict__ dst, const Xyzz* __restrict__ data, Xyzz* __restrict__ tab,
                                                           const RootSplit* __restrict__ kroots, u32 n, int s, u32 W, int inverse,
                                                           size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;  // total is a multiple of 2 * G: whole butterflies drop out together
    const int r = (int)(t % G);
    const size_t unit = t / G;
    const int half = (int)(unit & 1);
    const size_t bf = unit >> 1;
    const u32 halfn = n >> 1;
    const size_t xf = bf / halfn;
    const u32 b = (u32)(bf % halfn);
    const u32 hs = 1u << s, j = b & (hs - 1);
    const u32 i0 = ((b >> s) << (s + 1)) | j, i1 = i0 + hs;
    const Xyzz* base = data + xf * n;
    Xyzz y = base[i1];
    const u32 idx = j * (W >> (s + 1));
    if (idx != 0) {
        const RootSplit& rs = kroots[inverse ? W - idx : idx];
        if (!g1::is_inf(y)) {
            bool negate = rs.neg[half] != 0;
            if (half) {
                y.x = fp28::mul(y.x, beta28());
                negate = !negate;
            }
            if (negate) y.y = fp28::neg<8>(y.y);
            grp::scalar_mul128<G>(y, rs.k[half], tab + unit, total / G, r);
        }
        const Xyzz other = grp::from_partner<G>(y);
        grp::dadd<G>(y, other, r);
    }
    Xyzz x = base[i0];
    if (half) y.y = fp28::neg<8>(y.y);
    grp::dadd<G>(x, y, r);
    if (r == 0) dst[xf * n + (half ? i1 : i0)] = x;
}

// stage s with G lanes per BUTTERFLY and the two GLV halves in one chain (grp::scalar_mul_glv); total = G * butterflies
template <int G>
__global__ void __launch_bounds__(BOUND) k_g1_stage_bf(Xyzz* __restrict__ dst, const Xyzz* __restrict__ data, Xyzz* __restrict__ tab,
                                                        const RootSplit* __restrict__ kroots, u32 n, int s, u32 W, int inverse,
                                                        size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int r = (int)(t % G);
    const size_t bf = t / G;
    const u32 halfn = n >> 1;
    const size_t xf = bf / halfn;
    const u32 b = (u32)(bf % halfn);
    const u32 hs = 1u << s, j = b & (hs - 1);
    const u32 i0 = ((b >> s) << (s + 1)) | j, i1 = i0 + hs;
    const Xyzz* base = data + xf * n;
    Xyzz y = base[i1];
    const u32 idx = j * (W >> (s + 1));
    if (idx != 0) grp::scalar_mul_glv<G>(y, kroots[inverse ? W - idx : idx], tab + bf, total / G, r);
    const Xyzz x = base[i0];
    Xyzz a = x;
    grp::dadd<G>(a, y, r);
    if (r == 0) dst[xf * n + i0] = a;
    if (!g1::is_inf(y)) y.y = fp28::neg<8>(y.y);
    a = x;
    grp::dadd<G>(a, y, r);
    if (r == 0) dst[xf * n + i1] = a;
}

// data[i] *= inv_n, G lanes per point, one chain
template <int G>
__global__ void __launch_bounds__(BOUND) k_g1_scale_bf(Xyzz* __restrict__ data, Xyzz* __restrict__ tab, RootSplit inv_n, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // total = G * points
    if (t >= total) return;
    const int r = (int)(t % G);
    const size_t i = t / G;
    Xyzz p = data[i];
    grp::scalar_mul_glv<G>(p, inv_n, tab + i, total / G, r);
    if (r == 0) data[i] = p;
}

// data[i] *= inv_n, G lanes per half
template <int G>
__global__ void __launch_bounds__(BOUND) k_g1_scale_grp(Xyzz* __restrict__ data, Xyzz* __restrict__ tab, RootSplit inv_n, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // total = 2 * G * points
    if (t >= total) return;
    const int r = (int)(t % G);
    const size_t unit = t / G;
    const int half = (int)(unit & 1);
    Xyzz p = data[unit >> 1];
    if (!g1::is_inf(p)) {
        bool negate = inv_n.neg[half] != 0;
        if (half) {
            p.x = fp28::mul(p.x, beta28());
            negate = !negate;
        }
        if (negate) p.y = fp28::neg<8>(p.y);
        grp::scalar_mul128<G>(p, inv_n.k[half], tab + unit, total / G, r);
    }
    const Xyzz other = grp::from_partner<G>(p);
    grp::dadd<G>(p, other, r);
    if (half == 0 && r == 0) data[unit >> 1] = p;
}

// ---------------------------------------------------------------- limb-parallel stages (small grids)
// A stage is one 128-bit scalar multiplication deep however few butterflies it has: a single lane runs ~170 dependent
// point operations of ~9.5 us each (1.6 ms per stage; FK20 is 12 such stages — a 19 ms floor for any batch that does
// not fill the chip).  Below that size the chain is what counts, not the instruction total: here ONE WAVE runs a
// half-butterfly with the limb-parallel arithmetic of g1w.hip.h (a doubling is 3 multiplication steps deep instead of
// 9, an addition 4 instead of 14; ~0.5 us per step), signed 4-bit windows (multiples 1..8 and their negatives in LDS),
// and a second small kernel adds the two halves and forms x + t, x - t.  About 5x the instructions per butterfly of the
// single-lane kernel, so it is used while the waves of a stage fit the chip a few times over (KZGAMD_G1_WIDE_MAX).
constexpr int WTAB = 8;
struct WideScratch {
    u32 sh[16];              // g1w's exact zero test
    u32 tab[WTAB][5][16];    // k*P, k = 1..8: x, y, -y, zzz, zz (one limb per lane of a row)
    int dig[32];             // signed digits of the half-scalar, least significant first
};

__device__ __forceinline__ u32 wide_const_one(int lane) {
    constexpr u32 t[16] = {fp28::one_l(0), fp28::one_l(1), fp28::one_l(2),  fp28::one_l(3),  fp28::one_l(4),  fp28::one_l(5),
                           fp28::one_l(6), fp28::one_l(7), fp28::one_l(8),  fp28::one_l(9),  fp28::one_l(10), fp28::one_l(11),
                           fp28::one_l(12), fp28::one_l(13), 0u, 0u};
    return t[lane & 15];
}
__device__ __forceinline__ u32 wide_const_beta(int lane) {  // beta28(), one limb per lane
    constexpr u32 t[16] = {0xa75929au, 0x681b798u, 0x22a3e9du, 0xabc02bfu, 0x4e5bb45u, 0x55e6e7eu, 0x4814117u, 0x6d04f1bu,
                           0xae3387du, 0x54acb0cu, 0xa4c74bu,  0x56138b5u, 0xb64e066u, 0x76f2u,    0u,         0u};
    return t[lane & 15];
}
// -a for a normalized value below 31p: (32p - a) * 1, below 2p again (the point routines take Y < 18p)
__device__ __forceinline__ u32 wide_neg(u32 a, const fpw::Lane& lc, int lane) {
    return fpw::wmul(fpw::wsub32(0u, a, lc), wide_const_one(lane), lc);
}

// p <- (+-k) * (half ? [x^2]p : p) for a 127-bit k; the whole wave works on the one point
__device__ void wide_half_mul(g1w::WPt& p, const u32 k[4], bool negate, int half, const fpw::Lane& lc, WideScratch& ws, int lane) {
    using namespace fpw;
    if (g1w::is_inf(p)) return;
    const int row = lane >> 4, li = lane & 15;
    if (half) negate = !negate;  // [x^2](X, Y) = (beta * X, -Y)
    {
        // one multiplication step: row 0 the (conditional) negation of Y, row 1 beta * X
        const u32 t = wmul4(rows4(row, wsub32(0u, p.y, lc), p.x, 0u, 0u), rows4(row, wide_const_one(lane), wide_const_beta(lane), 0u, 0u), lc);
        if (negate) p.y = row_all(t, 0, lane);
        if (half) p.x = row_all(t, 1, lane);
    }
    if (lane == 0) {
        u32 carry = 0;
        for (int w = 0; w < 32; ++w) {
            const u32 v = ((k[w >> 3] >> ((w & 7) * 4)) & 15u) + carry;
            carry = v > 8u ? 1u : 0u;
            ws.dig[w] = (int)v - (int)(carry << 4);
        }
    }
    // the table: P, 2P, 3P .. 8P
    auto put = [&](int e, const g1w::WPt& q) {
        if (row == 0) {
            ws.tab[e][0][li] = q.x;
            ws.tab[e][1][li] = q.y;
            ws.tab[e][3][li] = q.zzz;
            ws.tab[e][4][li] = q.zz;
        }
    };
    g1w::WPt m = p;
    put(0, m);
    g1w::dbl(m, lc, lane);
    put(1, m);
    for (int e = 2; e < WTAB; ++e) {
        g1w::dadd(m, p, lc, ws.sh, lane);
        put(e, m);
    }
    __syncthreads();
    // the negated Y of the eight multiples: two multiplication steps, four entries each
    for (int g = 0; g < WTAB; g += 4) {
        const u32 y = ws.tab[g + row][1][li];
        const u32 t = wmul4(wsub32(0u, y, lc), wide_const_one(lane), lc);
        ws.tab[g + row][2][li] = t;
    }
    __syncthreads();
    g1w::WPt acc;
    g1w::set_inf(acc);
    for (int w = 31; w >= 0; --w) {
        if (!g1w::is_inf(acc)) {
            g1w::dbl(acc, lc, lane);
            g1w::dbl(acc, lc, lane);
            g1w::dbl(acc, lc, lane);
            g1w::dbl(acc, lc, lane);
        }
        const int d = ws.dig[w];
        if (d != 0) {
            const int e = (d < 0 ? -d : d) - 1;
            g1w::WPt q;
            q.x = ws.tab[e][0][li];
            q.y = ws.tab[e][d < 0 ? 2 : 1][li];
            q.zzz = ws.tab[e][3][li];
            q.zz = ws.tab[e][4][li];
            g1w::dadd(acc, q, lc, ws.sh, lane);
        }
    }
    p = acc;
}

// one wave per half-butterfly of stage s: h[unit] = this half of w^j * y  (nothing for the unit twiddle)
__global__ void __launch_bounds__(64) k_g1_stage_mul_wide(Xyzz* __restrict__ h, const Xyzz* __restrict__ data,
                                                          const RootSplit* __restrict__ kroots, u32 n, int s, u32 W, int inverse) {
    __shared__ WideScratch ws;
    const int lane = threadIdx.x;
    const size_t unit = blockIdx.x;
    const int half = (int)(unit & 1);
    const size_t bf = unit >> 1;
    const u32 halfn = n >> 1;
    const size_t xf = bf / halfn;
    const u32 b = (u32)(bf % halfn);
    const u32 hs = 1u << s, j = b & (hs - 1);
    const u32 idx = j * (W >> (s + 1));
    if (idx == 0) return;
    const u32 i1 = (((b >> s) << (s + 1)) | j) + hs;
    const RootSplit* rs = kroots + (inverse ? W - idx : idx);
    u32 k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = rs->k[half][i];
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt p = g1w::load(data + xf * n + i1, lane);
    wide_half_mul(p, k, rs->neg[half] != 0, half, lc, ws, lane);
    g1w::store(h + unit, p, lc, lane);
}

// one wave per butterfly: t = the two halves' sum (or y itself for the unit twiddle), dst = x + t, x - t
__global__ void __launch_bounds__(64) k_g1_stage_bfly_wide(Xyzz* __restrict__ dst, const Xyzz* __restrict__ data,
                                                           const Xyzz* __restrict__ h, u32 n, int s, u32 W) {
    __shared__ u32 sh[16];
    const int lane = threadIdx.x;
    const size_t bf = blockIdx.x;
    const u32 halfn = n >> 1;
    const size_t xf = bf / halfn;
    const u32 b = (u32)(bf % halfn);
    const u32 hs = 1u << s, j = b & (hs - 1);
    const u32 i0 = ((b >> s) << (s + 1)) | j, i1 = i0 + hs;
    const u32 idx = j * (W >> (s + 1));
    const fpw::Lane lc = fpw::lane_consts(lane);
    const Xyzz* base = data + xf * n;
    g1w::WPt t;
    if (idx != 0) {
        t = g1w::load(h + 2 * bf, lane);
        g1w::dadd(t, g1w::load(h + 2 * bf + 1, lane), lc, sh, lane);
    } else {
        t = g1w::load(base + i1, lane);
    }
    const g1w::WPt x = g1w::load(base + i0, lane);
    g1w::WPt a = x, d = x;
    g1w::dadd(a, t, lc, sh, lane);
    if (!g1w::is_inf(t)) t.y = wide_neg(t.y, lc, lane);
    g1w::dadd(d, t, lc, sh, lane);
    // one more multiplication step (by one) brings the four X, Y back below 2p: what is stored here is also read by
    // the single-lane kernels (g1_28.hip.h carries X < 10p, Y < 6p; the wide routines let them grow to 18p)
    const int row = lane >> 4;
    const u32 r = fpw::wmul4(fpw::rows4(row, a.x, a.y, d.x, d.y), wide_const_one(lane), lc);
    a.x = fpw::row_all(r, 0, lane);
    a.y = fpw::row_all(r, 1, lane);
    d.x = fpw::row_all(r, 2, lane);
    d.y = fpw::row_all(r, 3, lane);
    g1w::store(dst + xf * n + i0, a, lc, lane);
    g1w::store(dst + xf * n + i1, d, lc, lane);
}

// the n^-1 of an inverse transform, limb-parallel: h[2i], h[2i + 1] = the halves of inv_n * data[i]; then their sum
__global__ void __launch_bounds__(64) k_g1_scale_mul_wide(Xyzz* __restrict__ h, const Xyzz* __restrict__ data, RootSplit inv_n) {
    __shared__ WideScratch ws;
    const int lane = threadIdx.x;
    const size_t unit = blockIdx.x;
    const int half = (int)(unit & 1);
    u32 k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = inv_n.k[half][i];
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt p = g1w::load(data + (unit >> 1), lane);
    wide_half_mul(p, k, inv_n.neg[half] != 0, half, lc, ws, lane);
    g1w::store(h + unit, p, lc, lane);
}
__global__ void __launch_bounds__(64) k_g1_scale_sum_wide(Xyzz* __restrict__ data, const Xyzz* __restrict__ h) {
    __shared__ u32 sh[16];
    const int lane = threadIdx.x;
    const size_t i = blockIdx.x;
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt t = g1w::load(h + 2 * i, lane);
    g1w::dadd(t, g1w::load(h + 2 * i + 1, lane), lc, sh, lane);
    const int row = lane >> 4;
    const u32 r = fpw::wmul4(fpw::rows4(row, t.x, t.y, 0u, 0u), wide_const_one(lane), lc);  // X, Y below 2p (see the butterfly)
    t.x = fpw::row_all(r, 0, lane);
    t.y = fpw::row_all(r, 1, lane);
    g1w::store(data + i, t, lc, lane);
}

int ilog2(size_t n) {
    int l = 0;
    while (((size_t)1 << l) < n) ++l;
    return l;
}

RootSplit split_scalar(const Fr& plain) {  // canonical (non-Montgomery) scalar -> GLV halves
    RootSplit rs;
    memset(&rs, 0, sizeof rs);
    u32 k1[8], k2[8];
    kzgamd::glv_split(plain.v, k1, k2, rs.neg[0], rs.neg[1]);
    for (int i = 0; i < 4; ++i) {
        rs.k[0][i] = k1[i];
        rs.k[1][i] = k2[i];
    }
    return rs;
}

// Threads per workgroup of the long-lane kernels for a launch of `lanes` lanes: one wave per workgroup, or four once
// there is a wave for every SIMD (measured: no difference either way — the hardware spreads single-wave workgroups
// over the SIMDs of a CU as evenly as the waves of one workgroup).
inline unsigned block_threads(size_t lanes) { return lanes >= (size_t)1024 * 64 ? MAXTHREADS : NTHREADS; }
inline dim3 grid_for(size_t lanes) {
    const unsigned bt = block_threads(lanes);
    return dim3((unsigned)((lanes + bt - 1) / bt));
}

// the stages of one batch of transforms on bit-reversed-order data in bufs[0]; returns the index of the buffer that
// holds the result.  Stages whose half-butterflies (units) are few enough run limb-parallel (see above).
int enqueue_stages(NttCtx* ctx, Xyzz* bufs[2], Xyzz* tab, size_t n, size_t nbatch, int inverse, hipStream_t st) {
    const size_t total = n * nbatch, bf = total / 2;
    const int logn = ilog2(n);
    const bool wide = 2 * bf <= ctx->g1_wide_max;
    // Between g1_quad_max and g1_pair_max half-butterflies the better form depends on the transform: the twiddles of
    // short transforms (FK20: 128 points) are low-order roots whose GLV halves are short — two lanes per half win
    // (256 blobs: 23.8 ms against 28.6 with four) — while the full-length chains of a long transform want four
    // (2^15 points: 24.5 ms against 38 with two, 42.7 with one).
    const size_t quad_max = n > 256 && ctx->g1_quad_max ? (ctx->g1_pair_max > ctx->g1_quad_max ? ctx->g1_pair_max : ctx->g1_quad_max)
                                                        : ctx->g1_quad_max;
    for (int s = 0; s < logn; ++s) {
        Xyzz* dst = bufs[(s + 1) & 1];
        const Xyzz* src = bufs[s & 1];
        if (wide) {
            if (s > 0)  // stage 0 has unit twiddles only
                hipLaunchKernelGGL(k_g1_stage_mul_wide, dim3((unsigned)(2 * bf)), dim3(64), 0, st, tab, src,
                                   (const RootSplit*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0);
            hipLaunchKernelGGL(k_g1_stage_bfly_wide, dim3((unsigned)bf), dim3(64), 0, st, dst, src, (const Xyzz*)tab, (u32)n, s,
                               (u32)ctx->W);
        } else if (s > 0 && 2 * bf <= ctx->g1_bf_max) {
            hipLaunchKernelGGL(k_g1_stage_bf<4>, grid_for(4 * bf), dim3(block_threads(4 * bf)), 0, st, dst, src, tab,
                               (const RootSplit*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0, 4 * bf);
        } else if (s > 0 && 2 * bf <= quad_max) {
            hipLaunchKernelGGL(k_g1_stage_grp<4>, grid_for(8 * bf), dim3(block_threads(8 * bf)), 0, st, dst, src, tab,
                               (const RootSplit*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0, 8 * bf);
        } else if (s > 0 && 2 * bf <= ctx->g1_pair_max) {
            hipLaunchKernelGGL(k_g1_stage_grp<2>, grid_for(4 * bf), dim3(block_threads(4 * bf)), 0, st, dst, src, tab,
                               (const RootSplit*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0, 4 * bf);
        } else {  // stage 0 (unit twiddles: one addition per lane) and the grids that fill the chip on their own
            hipLaunchKernelGGL(k_g1_stage, grid_for(2 * bf), dim3(block_threads(2 * bf)), 0, st, dst, src, tab,
                               (const RootSplit*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0, 2 * bf);
        }
    }
    return logn & 1;
}

// data[i] *= inv_n for `total` points (the n^-1 of an inverse transform)
void enqueue_scale(NttCtx* ctx, Xyzz* data, Xyzz* tab, const RootSplit& inv_n, size_t total, hipStream_t st) {
    if (2 * total <= ctx->g1_wide_max) {
        hipLaunchKernelGGL(k_g1_scale_mul_wide, dim3((unsigned)(2 * total)), dim3(64), 0, st, tab, (const Xyzz*)data, inv_n);
        hipLaunchKernelGGL(k_g1_scale_sum_wide, dim3((unsigned)total), dim3(64), 0, st, data, (const Xyzz*)tab);
    } else if (2 * total <= ctx->g1_bf_max) {
        hipLaunchKernelGGL(k_g1_scale_bf<4>, grid_for(4 * total), dim3(block_threads(4 * total)), 0, st, data, tab, inv_n, 4 * total);
    } else if (2 * total <= ctx->g1_quad_max) {
        hipLaunchKernelGGL(k_g1_scale_grp<4>, grid_for(8 * total), dim3(block_threads(8 * total)), 0, st, data, tab,
                           inv_n, 8 * total);
    } else if (2 * total <= ctx->g1_pair_max) {
        hipLaunchKernelGGL(k_g1_scale_grp<2>, grid_for(4 * total), dim3(block_threads(4 * total)), 0, st, data, tab,
                           inv_n, 4 * total);
    } else {
        hipLaunchKernelGGL(k_g1_scale_xyzz, grid_for(2 * total), dim3(block_threads(2 * total)), 0, st, data, tab,
                           inv_n, 2 * total);
    }
}

void ensure_g1(NttCtx* ctx, size_t total, size_t tab_lanes) {
    if (!ctx->d_kroots) {
        std::vector<RootSplit> split(ctx->W + 1);
        for (size_t i = 0; i <= ctx->W; ++i) split[i] = split_scalar(ff::from_mont(ctx->roots[i]));
        NTT_TRY(hipMalloc(&ctx->d_kroots, (ctx->W + 1) * sizeof(RootSplit)));
        NTT_TRY(hipMemcpy(ctx->d_kroots, split.data(), (ctx->W + 1) * sizeof(RootSplit), hipMemcpyHostToDevice));
    }
    if (total > ctx->cap_g1) {
        if (ctx->d_p1) (void)hipFree(ctx->d_p1);
        if (ctx->d_pts) (void)hipFree(ctx->d_pts);
        ctx->d_p1 = ctx->d_pts = nullptr;
        ctx->cap_g1 = 0;
        NTT_TRY(hipMalloc(&ctx->d_p1, total * sizeof(blst_p1)));
        NTT_TRY(hipMalloc(&ctx->d_pts, 2 * total * sizeof(Xyzz)));  // ping-pong halves
        ctx->cap_g1 = total;
    }
    if (tab_lanes > ctx->cap_tab) {
        if (ctx->d_tab) (void)hipFree(ctx->d_tab);
        ctx->d_tab = nullptr;
        ctx->cap_tab = 0;
        NTT_TRY(hipMalloc(&ctx->d_tab, tab_lanes * NTAB * sizeof(Xyzz)));
        ctx->cap_tab = tab_lanes;
    }
}

void ensure_g1_tab(NttCtx* ctx, size_t tab_lanes) {
    if (!ctx->d_kroots) {
        std::vector<RootSplit> split(ctx->W + 1);
        for (size_t i = 0; i <= ctx->W; ++i) split[i] = split_scalar(ff::from_mont(ctx->roots[i]));
        NTT_TRY(hipMalloc(&ctx->d_kroots, (ctx->W + 1) * sizeof(RootSplit)));
        NTT_TRY(hipMemcpy(ctx->d_kroots, split.data(), (ctx->W + 1) * sizeof(RootSplit), hipMemcpyHostToDevice));
    }
    if (tab_lanes > ctx->cap_tab) {
        if (ctx->d_tab) (void)hipFree(ctx->d_tab);
        ctx->d_tab = nullptr;
        ctx->cap_tab = 0;
        NTT_TRY(hipMalloc(&ctx->d_tab, tab_lanes * NTAB * sizeof(Xyzz)));
        ctx->cap_tab = tab_lanes;
    }
}

}  // namespace

// G1 transforms of device-resident XYZZ data (natural order in `data`, nbatch transforms of n points each); `scratch`
// has the same size.  Enqueued on `st`, nothing synchronised; returns the buffer (data or scratch) that holds the
// result in natural order, or nullptr on an allocation failure.  scale_inverse = false leaves out the n^-1 of an inverse
// transform (a caller that can fold it into its scalars saves one scalar multiplication per point).  The per-lane tables are shared by the handle: one
// stream at a time (the c-kzg layer calls this under its settings lock and synchronises before it returns).
void* kzgamd::fftg1_device(NttCtx* ctx, void* data_v, void* scratch_v, size_t n, size_t nbatch, int inverse, hipStream_t st,
                           bool scale_inverse) {
    if (!ctx || n == 0 || (n & (n - 1)) || n > ctx->W) return nullptr;
    try {
        const size_t total = n * nbatch, bf = total / 2;
        const int logn = ilog2(n);
        // ctx->mu: ensure_g1_tab may reallocate the handle's tables, which kzgamd_fft_g1_batch (host buffers, its own
        // stream, under the same mutex) uses too.  The tables stay in use by the kernels enqueued here after this
        // returns: one stream at a time per handle, as stated above.
        std::lock_guard<std::mutex> lk(ctx->mu);
        ensure_g1_tab(ctx, 2 * total);
        Xyzz* bufs[2] = {(Xyzz*)scratch_v, (Xyzz*)data_v};
        Xyzz* tab = (Xyzz*)ctx->d_tab;
        hipLaunchKernelGGL(k_g1_brp_xyzz, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, bufs[0],
                           (const Xyzz*)data_v, (u32)n, logn, total);
        Xyzz* res = bufs[enqueue_stages(ctx, bufs, tab, n, nbatch, inverse, st)];
        if (inverse && n > 1 && scale_inverse) {
            Fr v = Fr::zero();
            v.v[0] = (u32)n;
            v.v[1] = (u32)((u64)n >> 32);
            const RootSplit inv_n = split_scalar(ff::from_mont(ff::inverse_bgcd(ff::to_mont(v))));
            enqueue_scale(ctx, res, tab, inv_n, total, st);
        }
        NTT_TRY(hipGetLastError());
        return res;
    } catch (const NttErr&) {
        return nullptr;
    }
}

extern "C" int kzgamd_fft_g1_batch(void* vctx, blst_p1* out, const blst_p1* in, size_t n, size_t nbatch, int inverse) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx || !out || !in) return -1;
    if (n > ctx->W) return 1;               // "Supplied list is longer than the available max width"
    if (n == 0 || (n & (n - 1))) return 2;  // "A list with power-of-two length expected"
    if (nbatch == 0) return 0;
    std::lock_guard<std::mutex> lk(ctx->mu);
    try {
        kzgamd::DeviceGuard on_device(ctx->device);
        NTT_TRY(on_device.err);
        const size_t total = n * nbatch, bf = total / 2;
        const int logn = ilog2(n);
        ensure_g1(ctx, total, inverse ? 2 * total : (bf ? 2 * bf : 2));
        Xyzz* pts = (Xyzz*)ctx->d_pts;
        Xyzz* tab = (Xyzz*)ctx->d_tab;
        hipStream_t st = ctx->stream;
        NTT_TRY(hipMemcpyAsync(ctx->d_p1, in, total * sizeof(blst_p1), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_g1_load, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pts, (const ff::Fp*)ctx->d_p1,
                           (u32)n, logn, total);
        Xyzz* bufs[2] = {pts, pts + total};
        Xyzz* res = bufs[enqueue_stages(ctx, bufs, tab, n, nbatch, inverse, st)];
        RootSplit inv_n;
        memset(&inv_n, 0, sizeof inv_n);
        int scale = inverse && n > 1 ? 1 : 0;
        if (scale) {
            Fr v = Fr::zero();
            v.v[0] = (u32)n;
            v.v[1] = (u32)((u64)n >> 32);
            inv_n = split_scalar(ff::from_mont(ff::inverse_bgcd(ff::to_mont(v))));
            if (2 * total <= ctx->g1_wide_max || 2 * total <= ctx->g1_bf_max || 2 * total <= ctx->g1_quad_max || 2 * total <= ctx->g1_pair_max) {
                // not enough points to fill the chip one lane each: the shorter-chain scaling, then a plain store
                enqueue_scale(ctx, res, tab, inv_n, total, st);
                scale = 0;
            }
        }
        hipLaunchKernelGGL(k_g1_store, grid_for(2 * total), dim3(block_threads(2 * total)), 0, st,
                           (ff::Fp*)ctx->d_p1, (const Xyzz*)res, tab, inv_n, scale, 2 * total);
        NTT_TRY(hipGetLastError());
        NTT_TRY(hipMemcpyAsync(out, ctx->d_p1, total * sizeof(blst_p1), hipMemcpyDeviceToHost, st));
        NTT_TRY(hipStreamSynchronize(st));
    } catch (const NttErr& e) {
        return -(int)e.e - 100;
    }
    return 0;
}

extern "C" int fft_g1(void* vctx, blst_p1* out, const blst_p1* in, size_t n, int inverse) {
    return kzgamd_fft_g1_batch(vctx, out, in, n, 1, inverse);
}
