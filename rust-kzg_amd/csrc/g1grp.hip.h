// Lane-group point arithmetic: a group of G = 1, 2 or 4 NEIGHBOURING lanes runs one chain of point operations.  Every
// lane of the group keeps the whole point state and the single-lane multiplier (fp28.hip.h); the INDEPENDENT products
// of a point formula run side by side on the lanes of the group — each lane multiplies the operand pair of its role, the
// G results go round the group with DPP quad permutes.  A doubling is 9 / 5 / 3 multiplications deep for G = 1 / 2 / 4,
// an addition 14 / 7 / 4, a mixed addition 10 / 5 / 4: the chain is 1 / 0.57 / 0.39 as long on 1 / 2 / 4 times the lanes.
// Users: the chain kernels of the G1 transforms (fftg1.hip, profiles/NOTES.md §16) and, since round 6, the accumulation
// of a few commitments (msm.hip: k_fbw_accum_quad).
#pragma once
#include "g1_28.hip.h"

namespace grp {
using ff::u32;
using fp28::Fe;
using g1::Xyzz;

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

// acc = 2 * acc, acc != infinity (dbl-2008-s-1, the bounds of g1::dbl)
template <int G>
__device__ __forceinline__ void dbl_body(Xyzz& acc, int r);
// acc += b (add-2008-s, the bounds and exceptional cases of g1::dadd).  Returns true when b == acc: the caller doubles
// acc (the one doubling of the loop body serves that case too).  Every lane of a group holds the same values, so the
// group branches as one.
template <int G>
__device__ __forceinline__ bool dadd_body(Xyzz& acc, const Xyzz& b, int r);

template <>
__device__ __forceinline__ void dbl_body<1>(Xyzz& acc, int) {
    g1::dbl(acc);
}
// [V, M] [W, S] [MM, ZZ3] [M3*(S - X3), W*Y] [ZZZ3]
template <>
__device__ __forceinline__ void dbl_body<2>(Xyzz& acc, int r) {
    using namespace fp28;
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
__device__ __forceinline__ void dbl_body<4>(Xyzz& acc, int r) {
    using namespace fp28;
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

template <>
__device__ __forceinline__ bool dadd_body<1>(Xyzz& acc, const Xyzz& b, int) {
    using namespace fp28;
    if (g1::is_inf(b)) return false;
    if (g1::is_inf(acc)) {
        acc = b;
        return false;
    }
    const Fe u = mul(acc.x, b.zz);
    const Fe s = mul(acc.y, b.zzz);
    const Fe p = sub<4>(mul(b.x, acc.zz), u);
    const Fe rr_ = sub<4>(mul(b.y, acc.zzz), s);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(rr_)) return true;
        g1::set_inf(acc);
        return false;
    }
    const Fe pp = sqr(p);
    const Fe ppp = mul(p, pp);
    const Fe q = mul(u, pp);
    const Fe x3 = sub<8>(sqr(rr_), addn(add(q, q), ppp));
    acc.y = mul2_inline(rr_, sub<16>(q, x3), sub_lazy<8>(zero(), s), ppp);  // R*(Q - X3) - S*PPP, one reduction
    acc.x = x3;
    acc.zz = mul(mul(acc.zz, b.zz), pp);
    acc.zzz = mul(mul(acc.zzz, b.zzz), ppp);
    return false;
}
// [U, U2] [S, S2] [PP, RR] [PPP, Q] [ZZ12, ZZZ12] [R*(Q - X3), S*PPP] [ZZ3, ZZZ3]
template <>
__device__ __forceinline__ bool dadd_body<2>(Xyzz& acc, const Xyzz& b, int r) {
    using namespace fp28;
    if (g1::is_inf(b)) return false;
    if (g1::is_inf(acc)) {
        acc = b;
        return false;
    }
    const bool r0 = r == 0;
    Fe t = mul(pick(r0, acc.x, b.x), pick(r0, b.zz, acc.zz));
    const Fe u = from_role<2, 0>(t);
    const Fe p = sub<4>(from_role<2, 1>(t), u);
    t = mul(pick(r0, acc.y, b.y), pick(r0, b.zzz, acc.zzz));
    const Fe s = from_role<2, 0>(t);
    const Fe rr_ = sub<4>(from_role<2, 1>(t), s);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(rr_)) return true;
        g1::set_inf(acc);
        return false;
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
    return false;
}
// [U, S, U2, S2] [PP, RR, ZZ12, ZZZ12] [PPP, Q, ZZ3, -] [R*(Q - X3), S*PPP, ZZZ3, -]
template <>
__device__ __forceinline__ bool dadd_body<4>(Xyzz& acc, const Xyzz& b, int r) {
    using namespace fp28;
    if (g1::is_inf(b)) return false;
    if (g1::is_inf(acc)) {
        acc = b;
        return false;
    }
    Fe t = mul(pick4(r, acc.x, acc.y, b.x, b.y), pick4(r, b.zz, b.zzz, acc.zz, acc.zzz));
    const Fe u = from_role<4, 0>(t), s = from_role<4, 1>(t);
    const Fe p = sub<4>(from_role<4, 2>(t), u), rr_ = sub<4>(from_role<4, 3>(t), s);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(rr_)) return true;
        g1::set_inf(acc);
        return false;
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
    return false;
}

// acc += (x2, y2), an affine point that is not at infinity, y2 already carrying its sign (madd-2008-s, the bounds and
// exceptional cases of g1::madd), on a group of four lanes:
// [U2, S2, -, -] [PP, RR, -, -] [PPP, Q, ZZ3, -] [R*(Q - X3), Y1*PPP, ZZZ3, -]
__device__ __forceinline__ void madd_body4(g1::Xyzz& acc, const Fe& x2, const Fe& y2, int r) {
    using namespace fp28;
    if (g1::is_inf(acc)) {
        g1::set_affine(acc, x2, y2);
        return;
    }
    const bool lo = (r & 1) == 0;
    Fe t = mul(pick(lo, x2, y2), pick(lo, acc.zz, acc.zzz));  // roles 2, 3 repeat 0, 1
    // P and R only ever feed multiplications: no carry pass (limbs < 2^28 + 2^29), pads for X1 < 9p and Y1 < 6p as in g1::madd
    const Fe p = sub_lazy<16>(from_role<4, 0>(t), acc.x), rr_ = sub_lazy<16>(from_role<4, 1>(t), acc.y);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(rr_)) g1::dbl_affine(acc, x2, y2);
        else g1::set_inf(acc);
        return;
    }
    t = mul(pick(lo, p, rr_), pick(lo, p, rr_));
    const Fe pp = from_role<4, 0>(t), rr = from_role<4, 1>(t);
    t = mul(pick4(r, p, acc.x, acc.zz, acc.zz), pp);
    const Fe ppp = from_role<4, 0>(t), q = from_role<4, 1>(t), zz3 = from_role<4, 2>(t);
    const Fe x3 = sub<8>(rr, addn(add(q, q), ppp));
    t = mul(pick4(r, rr_, acc.y, acc.zzz, acc.zzz), pick4(r, sub<16>(q, x3), ppp, ppp, ppp));
    acc.y = sub<4>(from_role<4, 0>(t), from_role<4, 1>(t));
    acc.zzz = from_role<4, 2>(t);
    acc.x = x3;
    acc.zz = zz3;
}

}  // namespace grp
