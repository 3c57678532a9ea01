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
// one lane per half-butterfly 31.6 -> 37.8 ms (the argument marshalling of the calls), profiles/NOTES.md §9.
#ifdef KZGAMD_FFTG1_OUTLINE_MUL
#define FP28_OUTLINE_MUL 1
#endif
#include "../../include/kzg_mi355x.h"
#include "ff.hip.h"
#include "g1_28.hip.h"
#include "g1w.hip.h"
#include "g1grp.hip.h"
#include "glv.hip.h"
#include "device_guard.h"
#include "ntt_internal.h"

using ff::Fr;
using ff::u32;
using ff::u64;
using g1::Xyzz;

namespace {

constexpr int WIN = 4;              // scalar window
constexpr int NTAB = (1 << WIN) - 1;  // table slots reserved per lane (the chain kernels use 9: multiples 1 .. 8 and the half's product)
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

__device__ __forceinline__ fp28::Fe beta28() {  // cube root of unity, Montgomery 2^392 (see msm.hip)
    constexpr u32 t[14] = {0xa75929au, 0x681b798u, 0x22a3e9du, 0xabc02bfu, 0x4e5bb45u, 0x55e6e7eu, 0x4814117u,
                           0x6d04f1bu, 0xae3387du, 0x54acb0cu, 0xa4c74bu, 0x56138b5u, 0xb64e066u, 0x76f2u};
    fp28::Fe b;
#pragma unroll
    for (int k = 0; k < 14; ++k) b.v[k] = t[k];
    return b;
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

// device-resident form: XYZZ in natural order -> bit-reversed order of each transform
__global__ void __launch_bounds__(256) k_g1_brp_xyzz(Xyzz* __restrict__ out, const Xyzz* __restrict__ in, u32 n, int logn,
                                                     size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t xf = t / n;
    out[t] = in[xf * n + brev((u32)(t % n), logn)];
}

// ---------------------------------------------------------------- the chain kernels: 1, 2 or 4 lanes per half-butterfly
// One half of a butterfly's scalar multiplication is a chain of ~170 point operations (signed 4-bit windows: 7 for
// the table of 1 .. 8 P, then 31 x (4 doublings + 1 addition)).  A group of G = 1, 2 or 4 neighbouring lanes runs it:
// every lane of the group keeps the whole point state and the single-lane multiplier, and the INDEPENDENT products of
// a point formula run side by side on the lanes of the group — each lane multiplies the operand pair of its role, the
// G results go round the group with DPP quad permutes.  A doubling is 9 / 5 / 3 multiplications deep for G = 1 / 2 / 4,
// an addition 14 / 7 / 4: the chain is 1 / 0.57 / 0.39 as long, on 1 / 2 / 4 times the lanes (1 / 1.14 / 1.55 times the
// instructions).  Which G a stage gets depends on how many half-butterflies it has (enqueue_stages).
//
// The kernels are ONE loop over the steps of the chain with ONE inlined doubling and ONE inlined addition in its body
// (the step number says which; the addition's second operand always comes from memory: a table slot, the partner
// half's product, the butterfly's other input).  No calls, so nothing lives in scratch: round 3's form called the point
// routines out of line with their 224-byte operands by reference — 688 B of scratch per lane and, at 1024 waves, more
// HBM traffic for call frames (2.6 GB per stage) than for the tables.
namespace grp {
using namespace ::grp;  // dpp / from_role / pick / dbl_body / dadd_body: g1grp.hip.h
using fp28::Fe;

// Booth digit w (4 bits, in [-8, 8]) of a 127-bit k:  d_w = k[4w-1] + k[4w] + 2 k[4w+1] + 4 k[4w+2] - 8 k[4w+3].
// w is the same for the whole wave; the words are picked with selects (a dynamic index would put k in scratch).
__device__ __forceinline__ int booth4(u32 k0, u32 k1, u32 k2, u32 k3, int w) {
    const int bit = 4 * w - 1;
    u32 v;
    if (bit < 0) {
        v = (k0 << 1) & 31u;
    } else {
        const int q = bit >> 5;
        const u32 lo = q == 0 ? k0 : q == 1 ? k1 : q == 2 ? k2 : k3;
        const u32 hi = q == 0 ? k1 : q == 1 ? k2 : q == 2 ? k3 : 0u;
        v = (u32)((((u64)hi << 32) | lo) >> (bit & 31)) & 31u;
    }
    return (int)((v + 1) >> 1) - (int)((v >> 4) << 4);
}

constexpr int TAB_STEPS = 7;            // 2P, then 3P .. 8P
constexpr int MAIN_STEPS = 32 * 5;      // per window: four doublings and an addition
constexpr int SLOT_HALF = 8;            // table slot 8: this half's finished product, for the partner group

// The chain of one half-butterfly.  On entry acc = the point (endomorphism and sign already applied) and `mul` says
// whether it is multiplied at all (not for a unit twiddle or a point at infinity); slots: this group's column of the
// table (slot e at slots[e * stride]), written by role 0 and read by the whole group.  Steps 0 .. 6 build the table,
// 7 .. 166 are the windows; then `tail_steps` additions whose operand tail(i) names:
//   tail(i).src == nullptr: nothing;  tail(i).negate_acc: acc <- -acc first (the x - t output).
// After the last main step the product is published in slot SLOT_HALF (publish == true) for the partner's first tail step.
struct TailOp {
    const Xyzz* src;
    bool negate_acc;
};
template <int G, class Tail>
__device__ __forceinline__ void run_chain(Xyzz& acc, bool mul, u32 k0, u32 k1, u32 k2, u32 k3, Xyzz* slots, size_t stride,
                                          bool publish, int tail_steps, Tail&& tail, int r) {
    if (mul && r == 0) slots[0] = acc;
    const int last = TAB_STEPS + MAIN_STEPS + tail_steps;
    // A wave none of whose lanes multiplies (stage 0: unit twiddles only; the j = 0 butterflies; points at infinity) has
    // nothing to do in the 167 table and window steps: it publishes its (unchanged) operand and goes to the tail.
    int first = 0;
    if (__ballot(mul) == 0) {
        if (publish) {
            if (r == 0) slots[(size_t)SLOT_HALF * stride] = acc;
            __threadfence_block();
        }
        first = TAB_STEPS + MAIN_STEPS;
    }
#pragma unroll 1
    for (int step = first; step < last; ++step) {
        const bool in_tab = step < TAB_STEPS, in_tail = step >= TAB_STEPS + MAIN_STEPS;
        if (step == TAB_STEPS && mul) g1::set_inf(acc);  // the windows start from infinity; P is in slot 0
        bool want_dbl = false, need_dbl = false;
        const Xyzz* src = nullptr;
        bool neg_src = false;
        if (in_tab) {
            want_dbl = step == 0;
            if (!want_dbl && mul) src = slots;
        } else if (!in_tail) {
            const int q = step - TAB_STEPS, w = 31 - q / 5;
            want_dbl = q % 5 < 4;
            if (!want_dbl && mul) {
                const int d = booth4(k0, k1, k2, k3, w);
                if (d != 0) {
                    src = slots + (size_t)((d < 0 ? -d : d) - 1) * stride;
                    neg_src = d < 0;
                }
            }
        } else {
            const TailOp op = tail(step - TAB_STEPS - MAIN_STEPS);
            src = op.src;
            if (op.negate_acc && !g1::is_inf(acc)) acc.y = fp28::neg<8>(acc.y);
        }
        if (src) {
            Xyzz qv = *src;
            if (neg_src) qv.y = fp28::neg<8>(qv.y);
            need_dbl = dadd_body<G>(acc, qv, r);
        }
        if (want_dbl ? (mul && !g1::is_inf(acc)) : need_dbl) dbl_body<G>(acc, r);
        if (in_tab && mul && r == 0) slots[(size_t)(step + 1) * stride] = acc;
        if (step == TAB_STEPS + MAIN_STEPS - 1 && publish) {
            if (r == 0) slots[(size_t)SLOT_HALF * stride] = acc;
            __threadfence_block();  // the partner group (same wave) reads it in its next step
        }
    }
}

// (+-k) * (half ? [x^2]p : p): the endomorphism and the sign go into the point before the chain
__device__ __forceinline__ void apply_half(Xyzz& p, bool negate, int half) {
    if (half) {
        p.x = fp28::mul(p.x, beta28());
        negate = !negate;  // [x^2](X, Y) = (beta * X, -Y)
    }
    if (negate) p.y = fp28::neg<8>(p.y);
}
}  // namespace grp

// stage s of the DIT network on bit-reversed-order data: pairs i0 and i0 + 2^s,
// twiddle w_n^(j * n / 2^(s+1)) = roots[j * (W >> (s+1))]  (fft_g1.rs:22-29 unrolled); 2 * G lanes per butterfly:
// the two halves of w^j * y on neighbouring groups, t = their sum, half 0 writes x + t, half 1 writes x - t
// (out of place: both halves read x and y).  total = 2 * G * butterflies lanes.
template <int G>
__global__ void __launch_bounds__(BOUND) k_g1_stage_chain(Xyzz* __restrict__ dst, const Xyzz* __restrict__ data, Xyzz* __restrict__ tab,
                                                           const RootSplit* __restrict__ kroots, u32 n, int s, u32 W, int inverse,
                                                           size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;  // total is a multiple of 2 * G: whole butterflies drop out together
    const int r = (int)(t % G);
    const size_t unit = t / G, nunits = total / G;
    const int half = (int)(unit & 1);
    const size_t bf = unit >> 1;
    const u32 halfn = n >> 1;
    const size_t xf = bf / halfn;
    const u32 b = (u32)(bf % halfn);
    const u32 hs = 1u << s, j = b & (hs - 1);
    const u32 i0 = ((b >> s) << (s + 1)) | j, i1 = i0 + hs;
    const Xyzz* base = data + xf * n;
    Xyzz acc = base[i1];
    const u32 idx = j * (W >> (s + 1));
    const bool twiddled = idx != 0;
    const bool mul = twiddled && !g1::is_inf(acc);
    u32 k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    if (twiddled) {
        const RootSplit* rs = kroots + (inverse ? W - idx : idx);
        k0 = rs->k[half][0];
        k1 = rs->k[half][1];
        k2 = rs->k[half][2];
        k3 = rs->k[half][3];
        if (mul) grp::apply_half(acc, rs->neg[half] != 0, half);
    }
    Xyzz* slots = tab + unit;
    const Xyzz* partner = tab + (unit ^ 1) + (size_t)grp::SLOT_HALF * nunits;
    const Xyzz* xin = base + i0;
    grp::run_chain<G>(acc, mul, k0, k1, k2, k3, slots, nunits, twiddled, 2,
                      [&](int i) {
                          if (i == 0) return grp::TailOp{twiddled ? partner : nullptr, false};  // t = both halves
                          return grp::TailOp{xin, half != 0};                                   // x + t  /  x - t
                      },
                      r);
    if (r == 0) dst[xf * n + (half ? i1 : i0)] = acc;
}

// data[i] *= inv_n (the n^-1 of an inverse transform): 2 * G lanes per point, in place (both groups of a point are in
// one wave: the loads of a pair precede its store)
template <int G>
__global__ void __launch_bounds__(BOUND) k_g1_scale_chain(Xyzz* __restrict__ data, Xyzz* __restrict__ tab, RootSplit inv_n, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // total = 2 * G * points
    if (t >= total) return;
    const int r = (int)(t % G);
    const size_t unit = t / G, nunits = total / G;
    const int half = (int)(unit & 1);
    Xyzz acc = data[unit >> 1];
    const bool mul = !g1::is_inf(acc);
    const u32 k0 = half ? inv_n.k[1][0] : inv_n.k[0][0], k1 = half ? inv_n.k[1][1] : inv_n.k[0][1];
    const u32 k2 = half ? inv_n.k[1][2] : inv_n.k[0][2], k3 = half ? inv_n.k[1][3] : inv_n.k[0][3];
    if (mul) grp::apply_half(acc, (half ? inv_n.neg[1] : inv_n.neg[0]) != 0, half);
    Xyzz* slots = tab + unit;
    const Xyzz* partner = tab + (unit ^ 1) + (size_t)grp::SLOT_HALF * nunits;
    grp::run_chain<G>(acc, mul, k0, k1, k2, k3, slots, nunits, true, 1, [&](int) { return grp::TailOp{partner, false}; }, r);
    if (half == 0 && r == 0) data[unit >> 1] = acc;
}

// XYZZ -> blst Jacobian, one lane per point
__global__ void __launch_bounds__(256) k_g1_store(ff::Fp* __restrict__ out, const Xyzz* __restrict__ data, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const Xyzz p = data[t];
    g1::to_blst_jacobian(out + t * 3, p);
}


// ---------------------------------------------------------------- limb-parallel stages (small grids)
// A stage is one 128-bit scalar multiplication deep however few butterflies it has: a single lane runs ~170 dependent
// point operations of ~9.5 us each (1.6 ms per stage; FK20 is 12 such stages — a 19 ms floor for any batch that does
// not fill the chip).  Below that size the chain is what counts, not the instruction total: here ONE WAVE runs a
// half-butterfly with the limb-parallel arithmetic of g1w.hip.h (a doubling is 3 multiplication steps deep instead of
// 9, an addition 4 instead of 14; ~0.5 us per step), signed 4-bit windows (multiples 1..8 and their negatives in LDS),
// and a second small kernel adds the two halves and forms x + t, x - t.  About 5x the instructions per butterfly of the
// single-lane kernel, so it is used while the waves of a stage fit the chip a few times over (tuning key g1_wide_max).
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
    const size_t quad_max = ctx->g1_quad_max;
    for (int s = 0; s < logn; ++s) {
        Xyzz* dst = bufs[(s + 1) & 1];
        const Xyzz* src = bufs[s & 1];
        if (wide) {
            if (s > 0)  // stage 0 has unit twiddles only
                hipLaunchKernelGGL(k_g1_stage_mul_wide, dim3((unsigned)(2 * bf)), dim3(64), 0, st, tab, src,
                                   (const RootSplit*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0);
            hipLaunchKernelGGL(k_g1_stage_bfly_wide, dim3((unsigned)bf), dim3(64), 0, st, dst, src, (const Xyzz*)tab, (u32)n, s,
                               (u32)ctx->W);
        } else if (s > 0 && 2 * bf <= quad_max) {
            hipLaunchKernelGGL(k_g1_stage_chain<4>, grid_for(8 * bf), dim3(block_threads(8 * bf)), 0, st, dst, src, tab,
                               (const RootSplit*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0, 8 * bf);
        } else if (s > 0 && 2 * bf <= ctx->g1_pair_max) {
            hipLaunchKernelGGL(k_g1_stage_chain<2>, grid_for(4 * bf), dim3(block_threads(4 * bf)), 0, st, dst, src, tab,
                               (const RootSplit*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0, 4 * bf);
        } else {  // stage 0 (unit twiddles: one addition per lane) and the grids that fill the chip on their own
            hipLaunchKernelGGL(k_g1_stage_chain<1>, grid_for(2 * bf), dim3(block_threads(2 * bf)), 0, st, dst, src, tab,
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
    } else if (2 * total <= ctx->g1_quad_max) {
        hipLaunchKernelGGL(k_g1_scale_chain<4>, grid_for(8 * total), dim3(block_threads(8 * total)), 0, st, data, tab, inv_n, 8 * total);
    } else if (2 * total <= ctx->g1_pair_max) {
        hipLaunchKernelGGL(k_g1_scale_chain<2>, grid_for(4 * total), dim3(block_threads(4 * total)), 0, st, data, tab, inv_n, 4 * total);
    } else {
        hipLaunchKernelGGL(k_g1_scale_chain<1>, grid_for(2 * total), dim3(block_threads(2 * total)), 0, st, data, tab, inv_n, 2 * total);
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
        if (inverse && n > 1) {
            Fr v = Fr::zero();
            v.v[0] = (u32)n;
            v.v[1] = (u32)((u64)n >> 32);
            enqueue_scale(ctx, res, tab, split_scalar(ff::from_mont(ff::inverse_bgcd(ff::to_mont(v)))), total, st);
        }
        hipLaunchKernelGGL(k_g1_store, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (ff::Fp*)ctx->d_p1, (const Xyzz*)res,
                           total);
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
