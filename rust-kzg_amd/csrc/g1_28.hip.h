// G1 point arithmetic for the MSM kernels, over fp28::Fe.
// XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) exactly as the reference's
// bucket type P1XYZZ (kzg/src/msm/pippenger_utils.rs:5-12); the formulas are EFD
// madd-2008-s / add-2008-s / dbl-2008-s-1, the same ones p1_dadd_affine / p1_dadd
// restate (pippenger_utils.rs:90-210), with the same exceptional cases
// (infinity, P == Q -> double, P == -Q -> infinity).
//
// Value bounds carried between calls (see fp28.hip.h): X < 10p, Y < 6p, ZZ, ZZZ < 2p (dbl and dadd also take Y <= 8p:
// a negated input; madd does not).
#pragma once
#include "fp28.hip.h"

namespace g1 {
using fp28::Fe;

struct Xyzz {
    Fe x, y, zzz, zz;
};

// table / input point, 128-byte slot: x, y in fp28 form (Montgomery 2^392, canonical value),
// flags bit0 = point at infinity
struct alignas(16) AffPt {
    Fe x, y;
    ff::u32 flags, pad[3];
};
static_assert(sizeof(AffPt) == 128, "AffPt slot");

// wide-table slot: x and y in fp28 form (canonical residues) in a 128-byte, cache-line-aligned slot.
// A valid curve point never has x = y = 0 (y^2 = x^3 + 4), so all-zero encodes infinity.
// Measured alternatives on MI355X (accumulation kernel, c = 14): 96-byte bit-packed slots +6 % (unpacking),
// 112-byte unpadded slots +2 % (gathers straddle cache lines); the padded slot is the fastest.
struct alignas(128) WidePt {
    Fe x, y;
    ff::u32 pad[4];
};
static_assert(sizeof(WidePt) == 128, "WidePt slot");

FF_HD void set_inf(Xyzz& p) {
    p.x = fp28::zero();
    p.y = fp28::zero();
    p.zzz = fp28::zero();
    p.zz = fp28::zero();
}
FF_HD bool is_inf(const Xyzz& p) { return fp28::is_zero_limbs(p.zz); }

FF_HD void set_affine(Xyzz& p, const Fe& x, const Fe& y) {
    p.x = x;
    p.y = y;
    p.zzz = fp28::one();
    p.zz = fp28::one();
}

// 2 * (x2, y2)  (mdbl-2008-s-1)
FF_HD void dbl_affine(Xyzz& out, const Fe& x2, const Fe& y2) {
    using namespace fp28;
    Fe u = addn(y2, y2);
    Fe zz = sqr(u);
    Fe zzz = mul(zz, u);
    Fe s = mul(x2, zz);
    Fe m = sqr(x2);
    Fe m3 = addn(add(m, m), m);
    Fe x3 = sub<8>(sqr(m3), addn(s, s));
    Fe y3 = sub<4>(mul(m3, sub<16>(s, x3)), mul(zzz, y2));
    out.x = x3;
    out.y = y3;
    out.zz = zz;
    out.zzz = zzz;
}

// acc = 2 * acc (dbl-2008-s-1); acc != infinity
FF_HD void dbl(Xyzz& acc) {
    using namespace fp28;
    Fe u = addn(acc.y, acc.y);
    Fe v = sqr(u);
    Fe w = mul(v, u);
    Fe s = mul(acc.x, v);
    Fe m = sqr(acc.x);
    Fe m3 = addn(add(m, m), m);
    Fe x3 = sub<8>(sqr(m3), addn(s, s));
    // Y3 = M*(S - X3) - W*Y as one two-product reduction, as in madd.  The pad is 16p, not 8p: the one-lane G1 stage
    // doubles points whose Y is fp28::neg<8>(y) = 8p - y in (6p, 8p] (fftg1.hip: apply_half, negated table entries),
    // and 8p - Y underflows its top limb once Y > 8p - 2^364.  16p - Y has limbs < 2^29: columns < 2^62, value
    // product < 140 p^2 (tests/test_host_cpu.py::test_dbl_of_a_negated_point_with_a_tiny_y).
    Fe y3 = mul2_inline(m3, sub<16>(s, x3), w, sub_lazy<16>(zero(), acc.y));
    acc.x = x3;
    acc.y = y3;
    acc.zz = mul(acc.zz, v);
    acc.zzz = mul(acc.zzz, w);
}

// acc += (x2, y2), affine point not at infinity; y2 already carries the sign
// (the caller passes 2p - y for a subtraction).  madd-2008-s.
FF_HD void madd(Xyzz& acc, const Fe& x2, const Fe& y2) {
    using namespace fp28;
    if (is_inf(acc)) {
        set_affine(acc, x2, y2);
        return;
    }
    // P and R only ever feed multiplications, so they skip the carry pass (limbs < 2^28 + 2^29)
    Fe p = sub_lazy<16>(mul(x2, acc.zz), acc.x);
    Fe r = sub_lazy<16>(mul(y2, acc.zzz), acc.y);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(r)) dbl_affine(acc, x2, y2);
        else set_inf(acc);
        return;
    }
    Fe pp = sqr(p);
    Fe ppp = mul(p, pp);
    Fe q = mul(acc.x, pp);
    Fe x3 = sub<8>(sqr(r), addn(add(q, q), ppp));
    // Y3 = R*(Q - X3) - Y1*PPP as one two-product Montgomery reduction: R*V + (8p - Y1)*PPP
    Fe y3 = mul2_inline(r, sub<16>(q, x3), sub_lazy<8>(zero(), acc.y), ppp);
    acc.x = x3;
    acc.y = y3;
    acc.zz = mul(acc.zz, pp);
    acc.zzz = mul(acc.zzz, ppp);
}

// acc += b   (add-2008-s) for b != acc; returns true and leaves acc alone when the two are the same point — the caller
// doubles.  Kernels that run their additions as ONE inlined site inside a loop (k_tile_sums) keep the doubling, which
// never runs on random data, as one site of its own instead of one per addition.
FF_HD bool dadd_unequal(Xyzz& acc, const Xyzz& b) {
    using namespace fp28;
    if (is_inf(b)) return false;
    if (is_inf(acc)) {
        acc = b;
        return false;
    }
    Fe u = mul(acc.x, b.zz);
    Fe s = mul(acc.y, b.zzz);
    Fe p = sub<4>(mul(b.x, acc.zz), u);
    Fe r = sub<4>(mul(b.y, acc.zzz), s);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(r)) return true;
        set_inf(acc);
        return false;
    }
    Fe pp = sqr(p);
    Fe ppp = mul(p, pp);
    Fe q = mul(u, pp);
    Fe x3 = sub<8>(sqr(r), addn(add(q, q), ppp));
    Fe y3 = mul2_inline(r, sub<16>(q, x3), sub_lazy<8>(zero(), s), ppp);  // R*(Q - X3) - S*PPP, one reduction
    acc.x = x3;
    acc.y = y3;
    acc.zz = mul(mul(acc.zz, b.zz), pp);
    acc.zzz = mul(mul(acc.zzz, b.zzz), ppp);
    return false;
}

// acc += b   (add-2008-s)
FF_HD void dadd(Xyzz& acc, const Xyzz& b) {
    using namespace fp28;
    if (is_inf(b)) return;
    if (is_inf(acc)) {
        acc = b;
        return;
    }
    Fe u = mul(acc.x, b.zz);
    Fe s = mul(acc.y, b.zzz);
    Fe p = sub<4>(mul(b.x, acc.zz), u);
    Fe r = sub<4>(mul(b.y, acc.zzz), s);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(r)) dbl(acc);
        else set_inf(acc);
        return;
    }
    Fe pp = sqr(p);
    Fe ppp = mul(p, pp);
    Fe q = mul(u, pp);
    Fe x3 = sub<8>(sqr(r), addn(add(q, q), ppp));
    Fe y3 = mul2_inline(r, sub<16>(q, x3), sub_lazy<8>(zero(), s), ppp);  // R*(Q - X3) - S*PPP, one reduction
    acc.x = x3;
    acc.y = y3;
    acc.zz = mul(mul(acc.zz, b.zz), pp);
    acc.zzz = mul(mul(acc.zzz, b.zzz), ppp);
}

// acc = 2^k * acc through Jacobian doublings (dbl-2009-l with S = 4*X*YY taken as one product:
// 3M + 4S against 6M + 3S for the XYZZ doubling) — the Horner steps of the windowed MSM are a serial
// chain of ~255 of these.  XYZZ -> Jacobian is (X*ZZ, Y*ZZZ, ZZ) (pippenger_utils.rs:84-88);
// Jacobian -> XYZZ is (X, Y, Z^2, Z^3).
// Bounds: a mul/sqr whose input bounds multiply to <= 64 p^2 returns < 1.04p.  Loop invariant:
// X < 18p, Y < 17.1p, Z < 2.1p, all normalized.
FF_HD void dbl_k(Xyzz& acc, int k) {
    using namespace fp28;
    if (is_inf(acc) || k <= 0) return;
    Fe X = mul(acc.x, acc.zz), Y = mul(acc.y, acc.zzz), Z = acc.zz;
    for (int i = 0; i < k; ++i) {
        Fe A = sqr(X), B = sqr(Y), C = sqr(B);         // XX, YY, YYYY  (< 1.04p each... A,B < 1.2p)
        Fe S = mul(X, B);                              // X*YY
        S = addn(S, S);
        S = addn(S, S);                                // 4*X*YY < 4.2p
        Fe E = addn(add(A, A), A);                     // 3*XX < 3.6p
        Fe X3 = sub<16>(sqr(E), addn(S, S));           // M^2 - 2S  < 18p
        Fe C8 = addn(C, C);
        C8 = addn(C8, C8);
        C8 = addn(C8, C8);                             // 8*YYYY < 8.4p
        Fe Y3 = sub<16>(mul(E, sub<32>(S, X3)), C8);   // M*(S - X3) - 8*YYYY  < 17.1p
        Fe Z3 = mul(Y, Z);
        Z = addn(Z3, Z3);                              // 2*Y*Z < 2.1p
        X = X3;
        Y = Y3;
    }
    const Fe o = one();
    acc.x = mul(X, o);  // back under the XYZZ bounds (value unchanged: multiplication by 1)
    acc.y = mul(Y, o);
    acc.zz = sqr(Z);
    acc.zzz = mul(acc.zz, Z);
}

// acc = k * acc for a small public integer k (double-and-add, MSB first)
FF_HD void mul_small(Xyzz& acc, ff::u32 k) {
    if (k == 0 || is_inf(acc)) {
        set_inf(acc);
        return;
    }
    Xyzz base = acc;
    int top = 31;
    while (!((k >> top) & 1)) --top;
    for (int b = top - 1; b >= 0; --b) {
        if (!is_inf(acc)) dbl(acc);
        if ((k >> b) & 1) dadd(acc, base);
    }
}

// Jacobian (X*ZZ, Y*ZZZ, ZZ) in blst layout (pippenger_utils.rs:84-88); infinity -> all-zero
FF_HD void to_blst_jacobian(ff::Fp out[3], const Xyzz& p) {
    if (is_inf(p)) {
        out[0] = ff::Fp::zero();
        out[1] = ff::Fp::zero();
        out[2] = ff::Fp::zero();
        return;
    }
    out[0] = fp28::to_blst(fp28::mul(p.x, p.zz));
    out[1] = fp28::to_blst(fp28::mul(p.y, p.zzz));
    out[2] = fp28::to_blst(p.zz);
}

}  // namespace g1
