// G1 points with every coordinate spread over the lanes of a 16-lane row (fpw.hip.h), replicated in the four rows of
// the wave: XYZZ coordinates as in g1_28.hip.h, one point operation per wave with its independent products side by
// side in the rows, for the serial tails of the MSM where no other parallelism is left.
// Coordinate bounds between calls: X, Y < 18p, ZZ, ZZZ < 2p, limbs <= 2^28.
#pragma once
#include "fpw.hip.h"
#include "g1_28.hip.h"

namespace g1w {
using ff::u32;
using ff::u64;
using fpw::Lane;

struct WPt {
    u32 x, y, zzz, zz;  // this lane's limb of each coordinate
};

// infinity = ZZ all-zero, as in g1::Xyzz
__device__ __forceinline__ bool is_inf(const WPt& p) { return __ballot(p.zz != 0) == 0; }
__device__ __forceinline__ void set_inf(WPt& p) { p.x = p.y = p.zzz = p.zz = 0; }

// lane i (< 14) of the row loads / stores limb i of each coordinate of a g1::Xyzz in memory
__device__ __forceinline__ WPt load(const g1::Xyzz* src, int lane) {
    WPt p;
    const u32* w = (const u32*)src;
    const int i = lane & 15;
    const bool live = i < fp28::L;
    p.x = live ? w[i] : 0u;
    p.y = live ? w[fp28::L + i] : 0u;
    p.zzz = live ? w[2 * fp28::L + i] : 0u;
    p.zz = live ? w[3 * fp28::L + i] : 0u;
    return p;
}
// stores exactly normalized limbs (what the single-lane code expects)
__device__ __forceinline__ void store(g1::Xyzz* dst, const WPt& p, const Lane& c, int lane) {
    u32* w = (u32*)dst;
    const int i = lane & 15;
    const u32 x = fpw::wnorm_full(p.x, c), y = fpw::wnorm_full(p.y, c), zzz = fpw::wnorm_full(p.zzz, c),
              zz = fpw::wnorm_full(p.zz, c);
    if (lane < fp28::L) {  // the rows hold the same value: the first one writes
        w[i] = x;
        w[fp28::L + i] = y;
        w[2 * fp28::L + i] = zzz;
        w[3 * fp28::L + i] = zz;
    }
}

// exact test a == 0 (mod p) for a normalized value < 64p (the wide twin of fp28::is_zero_mod_p): a multiple
// k*p has k = a_0 * p_0^-1 mod 2^28 < 64, which almost no other value passes; the exact comparison runs on
// the single-lane code behind that filter
__device__ __forceinline__ bool is_zero_mod_p(u32 a, u32* sh, int lane) {
    const u32 a0 = (u32)__builtin_amdgcn_readlane((int)a, 0);  // the rows hold the same value
    const u32 k = (a0 * fp28::P0INV_POS) & fp28::MASK;
#if !defined(KZGAMD_FORCE_EXACT_TESTS)  // the forced flavour takes the LDS exchange and the exact test on every call
    if (k >= 64) return false;
#else
    (void)k;
#endif
    fp28::Fe f = fpw::from_wide(a, sh, lane);
    fp28::norm(f);
    return fp28::is_zero_mod_p(f);
}

// acc = 2 * acc (dbl-2008-s-1); acc != infinity.  Three multiplication steps: [V, M], [W, S, ZZ*V, M3^2],
// [M3*(S - X3), W*Y, ZZZ*W].
__device__ __forceinline__ void dbl(WPt& acc, const Lane& c, int lane) {
    using namespace fpw;
    const int row = lane >> 4;
    const u32 U = waddn(acc.y, acc.y, c);
    u32 t = wmul4(rows4(row, U, acc.x, U, acc.x), rows4(row, U, acc.x, U, acc.x), c);
    const u32 V = row_all(t, 0, lane), M = row_all(t, 1, lane);
    const u32 M3 = wnorm(M + M + M, c);
    t = wmul4(rows4(row, U, acc.x, acc.zz, M3), rows4(row, V, V, V, M3), c);
    const u32 W = row_all(t, 0, lane), S = row_all(t, 1, lane), ZZ3 = row_all(t, 2, lane), MM = row_all(t, 3, lane);
    const u32 X3 = wsub16(MM, waddn(S, S, c), c);
    t = wmul4(rows4(row, M3, W, acc.zzz, W), rows4(row, wsub32(S, X3, c), acc.y, W, acc.y), c);
    acc.x = X3;
    acc.y = wsub16(row_all(t, 0, lane), row_all(t, 1, lane), c);
    acc.zz = ZZ3;
    acc.zzz = row_all(t, 2, lane);
}

// acc += b (add-2008-s) with the exceptional cases of g1::dadd; sh = 16 words of LDS scratch.  Four multiplication
// steps: [U, S, U2, S2], [PP, RR, ZZ1*ZZ2, ZZZ1*ZZZ2], [PPP, Q, ZZ3], [R*(Q - X3), S*PPP, ZZZ3].
__device__ __forceinline__ void dadd(WPt& acc, const WPt& b, const Lane& c, u32* sh, int lane) {
    using namespace fpw;
    if (is_inf(b)) return;
    if (is_inf(acc)) {
        acc = b;
        return;
    }
    const int row = lane >> 4;
    u32 t = wmul4(rows4(row, acc.x, acc.y, b.x, b.y), rows4(row, b.zz, b.zzz, acc.zz, acc.zzz), c);
    const u32 U = row_all(t, 0, lane), S = row_all(t, 1, lane);
    const u32 P = wsub32(row_all(t, 2, lane), U, c), R = wsub32(row_all(t, 3, lane), S, c);
    if (is_zero_mod_p(P, sh, lane)) {
        if (is_zero_mod_p(R, sh, lane)) dbl(acc, c, lane);
        else set_inf(acc);
        return;
    }
    t = wmul4(rows4(row, P, R, acc.zz, acc.zzz), rows4(row, P, R, b.zz, b.zzz), c);
    const u32 PP = row_all(t, 0, lane), RR = row_all(t, 1, lane), ZZ12 = row_all(t, 2, lane), ZZZ12 = row_all(t, 3, lane);
    t = wmul4(rows4(row, P, U, ZZ12, P), rows4(row, PP, PP, PP, PP), c);
    const u32 PPP = row_all(t, 0, lane), Q = row_all(t, 1, lane), ZZ3 = row_all(t, 2, lane);
    const u32 X3 = wsub16(RR, wnorm(Q + Q + PPP, c), c);
    t = wmul4(rows4(row, R, S, ZZZ12, R), rows4(row, wsub32(Q, X3, c), PPP, PPP, PPP), c);
    acc.x = X3;
    acc.y = wsub16(row_all(t, 0, lane), row_all(t, 1, lane), c);
    acc.zz = ZZ3;
    acc.zzz = row_all(t, 2, lane);
}

// acc += src[0] + src[stride] + ... (n points): the operand of the NEXT addition is loaded before the current one starts
// (four registers per lane), so that a chain fed from memory other CUs have just written waits for one round trip, not n.
// A rolled loop on purpose: unrolled over an array of operands the body (an inlined addition per element) is too large
// for the unroller and the array lands in scratch.
__device__ __forceinline__ void add_n(WPt& acc, const g1::Xyzz* src, size_t stride, int n, const Lane& c, u32* sh, int lane) {
    WPt nx = load(src, lane);
#pragma unroll 1
    for (int k = 0; k < n; ++k) {
        const WPt cur = nx;
        if (k + 1 < n) nx = load(src + (size_t)(k + 1) * stride, lane);
        dadd(acc, cur, c, sh, lane);
    }
}

// acc = 2^k * acc through Jacobian doublings (3 multiplication steps each)
__device__ __forceinline__ void dbl_k(WPt& acc, int k, const Lane& c, int lane) {
    using namespace fpw;
    if (k <= 0 || is_inf(acc)) return;
    const int row = lane >> 4;
    const u32 t = wmul4(rows4(row, acc.x, acc.y, acc.x, acc.y), rows4(row, acc.zz, acc.zzz, acc.zz, acc.zzz), c);
    u32 X = row_all(t, 0, lane), Y = row_all(t, 1, lane), Z = acc.zz;
    for (int i = 0; i < k; ++i) wdbl(X, Y, Z, c, lane);
    acc.x = X;
    acc.y = Y;
    acc.zz = wsqr(Z, c);
    acc.zzz = wmul(acc.zz, Z, c);
}

// wide -> single-lane (the same value in every lane), exactly normalized
__device__ __forceinline__ g1::Xyzz to_single(const WPt& p, const Lane& c, u32* sh, int lane) {
    g1::Xyzz r;
    r.x = fpw::from_wide(fpw::wnorm_full(p.x, c), sh, lane);
    r.y = fpw::from_wide(fpw::wnorm_full(p.y, c), sh, lane);
    r.zzz = fpw::from_wide(fpw::wnorm_full(p.zzz, c), sh, lane);
    r.zz = fpw::from_wide(fpw::wnorm_full(p.zz, c), sh, lane);
    return r;
}

}  // namespace g1w
