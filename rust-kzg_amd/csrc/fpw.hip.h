// Fp spread over the lanes of a 16-lane DPP row: limb i of the 14 x 28-bit form (fp28.hip.h) lives in lane i of the
// row, lanes 14 and 15 hold zero.  A wave has four rows: values are kept replicated in all four, and a multiplication
// step (wmul4) takes a different operand pair in each row — the independent products of a point formula run side by
// side, so a doubling is 3 multiplication steps deep instead of 7 and an addition 4 instead of 14.
//
// Why: the tails of the MSM (Horner over the window sums, the short per-window chains) are single dependency
// chains of point doublings.  A lane that owns a whole field element issues ~490 VALU instructions per
// multiplication and a wave cannot issue faster than one instruction per ~5 cycles however few lanes are live,
// so the chain runs at ~8 us per doubling.  With the limbs across lanes the same multiplication is 14 steps of
//     acc += a_i * b_j (b_j by v_readlane -> SGPR) ; m = acc_0 * p' (scalar ALU) ; acc += p_i * m ;
//     acc_i <- low28(acc_{i+1}) + (acc_i >> 28)            (one DPP row shift folded into the add)
// ~150 VALU instructions, three times shorter.  Only worth it where there is no other parallelism left.
//
// Representation: limbs <= 2^28 after wnorm (top limb, lane 13, unbounded); wmul/wsqr accept limbs < 2^29
// (one lazy add of normalized values) and return normalized limbs, value < 2p under the same product bound as
// fp28::mul.  Subtractions add a multiple of p whose low limbs are >= 2^29 - 2, so they never go negative.
#pragma once
#include "fp28.hip.h"

namespace fpw {
using ff::u32;
using ff::u64;
constexpr u32 MASK = fp28::MASK;

// lane i <- lane i-1 (lane 0 <- 0) / lane i <- lane i+1 (lane 15 <- 0), within the 16-lane row
__device__ __forceinline__ u32 from_prev(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true); }
__device__ __forceinline__ u32 from_next(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xF, 0xF, true); }

// K*p with two units borrowed from each limb above: limbs 0..12 >= 2^29 - 2, same value as fp28::pad_l<K>
template <int K>
__device__ __forceinline__ constexpr u32 wpad_l(int i) {
    return i >= 14 ? 0u : fp28::pad_l<K>(i) + (i <= 12 ? (1u << 28) : 0u) - (i >= 1 ? 1u : 0u);
}

// per-lane constants
struct Lane {
    u32 p;       // limb of p
    u32 nmask;   // 2^28 - 1 for limbs 0..12, all ones above (the top limb keeps its excess)
    u32 pad16, pad32;
};

__device__ __forceinline__ Lane lane_consts(int lane) {
    constexpr u32 P[16] = {fp28::pl(0), fp28::pl(1), fp28::pl(2),  fp28::pl(3),  fp28::pl(4),  fp28::pl(5),  fp28::pl(6), fp28::pl(7),
                           fp28::pl(8), fp28::pl(9), fp28::pl(10), fp28::pl(11), fp28::pl(12), fp28::pl(13), 0u,          0u};
    constexpr u32 D16[16] = {wpad_l<16>(0), wpad_l<16>(1), wpad_l<16>(2),  wpad_l<16>(3),  wpad_l<16>(4),  wpad_l<16>(5),
                             wpad_l<16>(6), wpad_l<16>(7), wpad_l<16>(8),  wpad_l<16>(9),  wpad_l<16>(10), wpad_l<16>(11),
                             wpad_l<16>(12), wpad_l<16>(13), 0u, 0u};
    constexpr u32 D32[16] = {wpad_l<32>(0), wpad_l<32>(1), wpad_l<32>(2),  wpad_l<32>(3),  wpad_l<32>(4),  wpad_l<32>(5),
                             wpad_l<32>(6), wpad_l<32>(7), wpad_l<32>(8),  wpad_l<32>(9),  wpad_l<32>(10), wpad_l<32>(11),
                             wpad_l<32>(12), wpad_l<32>(13), 0u, 0u};
    Lane c;
    c.p = P[lane & 15];
    c.nmask = (lane & 15) < 13 ? MASK : 0xffffffffu;
    c.pad16 = D16[lane & 15];
    c.pad32 = D32[lane & 15];
    return c;
}

// two carry rounds: limbs < 2^31 in, limbs <= 2^28 out (top limb absorbs)
__device__ __forceinline__ u32 wnorm(u32 x, const Lane& c) {
    x = (x & c.nmask) + from_prev((x & ~c.nmask) >> 28);
    x = (x & c.nmask) + from_prev((x & ~c.nmask) >> 28);
    return x;
}

// exact normalization (limbs 0..12 < 2^28): a carry travels at most 13 lanes
__device__ __forceinline__ u32 wnorm_full(u32 x, const Lane& c) {
#pragma unroll
    for (int r = 0; r < fp28::L; ++r) x = (x & c.nmask) + from_prev((x & ~c.nmask) >> 28);
    return x;
}

__device__ __forceinline__ u32 wadd(u32 a, u32 b) { return a + b; }  // lazy
__device__ __forceinline__ u32 waddn(u32 a, u32 b, const Lane& c) { return wnorm(a + b, c); }
// a + 16p - b, a + 32p - b (b normalized, value below 15p / 31p)
__device__ __forceinline__ u32 wsub16(u32 a, u32 b, const Lane& c) { return wnorm(a + c.pad16 - b, c); }
__device__ __forceinline__ u32 wsub32(u32 a, u32 b, const Lane& c) { return wnorm(a + c.pad32 - b, c); }

// lane J of each row to every lane of that row
template <int J>
__device__ __forceinline__ u32 row_lane(u32 x) {
    // row_newbcast:J; every lane has a source, bound_ctrl only spares the compiler the "old value" move
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + J, 0xF, 0xF, true);
}
// row k of the wave to all four rows (ds_bpermute: a lane permutation through the LDS crossbar, no memory)
__device__ __forceinline__ u32 row_all(u32 x, int k, int lane) {
    return (u32)__builtin_amdgcn_ds_bpermute(((k << 4) | (lane & 15)) << 2, (int)x);
}
// per-row choice: row r of the result is row r of a_r
__device__ __forceinline__ u32 rows4(int row, u32 a0, u32 a1, u32 a2, u32 a3) {
    return row == 0 ? a0 : row == 1 ? a1 : row == 2 ? a2 : a3;
}

// a * b * 2^-392 mod p, independently in each of the four rows.  14 steps of
//     acc += a_i * b_j (b_j by a DPP row broadcast) ; m = acc_0 * p' ; acc += p_i * m ;
//     acc_i <- low28(acc_{i+1}) + (acc_i >> 28)            (one DPP row shift folded into the add)
// A step is one dependency chain of eight instructions, ~80 cycles for a lone wave (tools/lone_wave_issue.hip: a
// dependent v_mad_u64_u32 follows its producer after 12.7 cycles, a DPP read of a fresh result after 16.4, a plain
// instruction after 8.3; independent ones issue 5 apart).  KZGAMD_WMUL_DIGIT_AHEAD (round 6, measured, not adopted) takes
// the quotient digit off the first multiply-add — m = acc_0 * p' + (a_0 * p') * b_j mod 2^28, the second product known
// before the chain starts: six instructions on the chain, but nine issued, and in-order issue puts the three others on
// the path: 1.48 us per doubling against 1.40 (tools/wmul_bench.hip; same checksums: the digits are the same digits).
__device__ __forceinline__ u32 wmul4(u32 a, u32 b, const Lane& c) {
    u32 acc = 0;
    // the 14 broadcasts of b do not depend on the accumulator chain: taken up front, they leave the chain
    // mad -> broadcast of lane 0 -> quotient digit -> mad -> shift down
    const u32 bb[14] = {row_lane<0>(b), row_lane<1>(b), row_lane<2>(b),  row_lane<3>(b),  row_lane<4>(b),  row_lane<5>(b),  row_lane<6>(b),
                        row_lane<7>(b), row_lane<8>(b), row_lane<9>(b), row_lane<10>(b), row_lane<11>(b), row_lane<12>(b), row_lane<13>(b)};
#if defined(KZGAMD_WMUL_DIGIT_AHEAD)
    const u32 a0p = row_lane<0>(a) * fp28::P0INV;  // only its low 28 bits matter below
#define KZG_WSTEP(J)                                                                            \
    {                                                                                           \
        const u32 m = (u32)((u64)row_lane<0>(acc) * fp28::P0INV + (u64)(a0p * bb[J])) & MASK;   \
        u64 t = (u64)a * bb[J] + acc;                                                           \
        asm("" : "+v"(t)); /* keeps acc inside this multiply-add: re-associated, it comes back as a 64-bit add on the chain */ \
        t += (u64)c.p * m;                                                                      \
        acc = from_next((u32)t & MASK) + (u32)(t >> 28);                                        \
    }
#else
#define KZG_WSTEP(J)                                               \
    {                                                              \
        const u32 bj = bb[J];                                      \
        u64 t = (u64)a * bj + acc;                                 \
        const u32 m = (row_lane<0>((u32)t) * fp28::P0INV) & MASK;  \
        t += (u64)c.p * m;                                         \
        acc = from_next((u32)t & MASK) + (u32)(t >> 28);           \
    }
#endif
    KZG_WSTEP(0) KZG_WSTEP(1) KZG_WSTEP(2) KZG_WSTEP(3) KZG_WSTEP(4) KZG_WSTEP(5) KZG_WSTEP(6)
    KZG_WSTEP(7) KZG_WSTEP(8) KZG_WSTEP(9) KZG_WSTEP(10) KZG_WSTEP(11) KZG_WSTEP(12) KZG_WSTEP(13)
#undef KZG_WSTEP
    return wnorm(acc, c);
}
// the same product in every row (operands replicated)
__device__ __forceinline__ u32 wmul(u32 a, u32 b, const Lane& c) { return wmul4(a, b, c); }
__device__ __forceinline__ u32 wsqr(u32 a, const Lane& c) { return wmul4(a, a, c); }

// Jacobian doubling (dbl-2009-l with S = 4*X*YY as one product), the wide twin of the loop body of g1::dbl_k, in three
// multiplication steps: [XX, YY, YZ], [YYYY, X*YY, E^2], [E*(S - X3)].
// Bounds as there: X < 18p, Y < 17.1p, Z < 2.1p, all normalized.
__device__ __forceinline__ void wdbl(u32& X, u32& Y, u32& Z, const Lane& c, int lane) {
    const int row = lane >> 4;
    u32 t = wmul4(rows4(row, X, Y, Y, X), rows4(row, X, Y, Z, X), c);
    const u32 A = row_all(t, 0, lane), B = row_all(t, 1, lane), YZ = row_all(t, 2, lane);
    const u32 E = wnorm(A + A + A, c);        // 3*XX
    t = wmul4(rows4(row, B, X, E, E), rows4(row, B, B, E, E), c);
    const u32 C = row_all(t, 0, lane), EE = row_all(t, 2, lane);
    u32 S = row_all(t, 1, lane);
    S = waddn(S, S, c);
    S = waddn(S, S, c);                       // 4*X*YY
    const u32 X3 = wsub16(EE, waddn(S, S, c), c);
    u32 C8 = waddn(C, C, c);
    C8 = waddn(C8, C8, c);
    C8 = waddn(C8, C8, c);                    // 8*YYYY
    const u32 Y3 = wsub16(wmul4(E, wsub32(S, X3, c), c), C8, c);
    Z = waddn(YZ, YZ, c);
    X = X3;
    Y = Y3;
}

// The single-lane code that runs next to the wide code works on values that are the same in every lane.  The
// compiler sees that and moves it to the scalar ALU, where a 64-bit multiply-add is four instructions instead
// of one: keep such values in VGPRs.
__device__ __forceinline__ void keep_in_vgprs(fp28::Fe& a) {
#pragma unroll
    for (int k = 0; k < fp28::L; ++k) asm volatile("" : "+v"(a.v[k]));
}

// Ordering of LDS accesses WITHIN a wave: the LDS executes one wave's operations in issue order, so lanes of a wave
// can exchange through it without a workgroup barrier; the compiler only has to keep the accesses in program order.
// (Round 3 used __syncthreads() here.  In a workgroup of several waves that each run their own chain — the hybrid block
// sum — one wave's rare exact zero test then executed barriers the others did not, let it run ahead of a later real
// barrier and read partial sums that were not written yet: a wrong result about once per 10^5 small batches, found by
// tools/fuzz_ckzg.py in round 4; tests/golden/sparse_blob_hybrid_fold.json.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// single-lane element (the same value in every lane) <-> wide, through 16 words of LDS THAT BELONG TO THE CALLING WAVE
__device__ __forceinline__ u32 to_wide(const fp28::Fe& a, u32* sh, int lane) {
    wave_sync();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < fp28::L; ++k) sh[k] = a.v[k];
        sh[14] = 0;
        sh[15] = 0;
    }
    wave_sync();
    return sh[lane & 15];
}
__device__ __forceinline__ fp28::Fe from_wide(u32 w, u32* sh, int lane) {
    wave_sync();
    sh[lane & 15] = w;
    wave_sync();
    fp28::Fe r;
#pragma unroll
    for (int k = 0; k < fp28::L; ++k) r.v[k] = sh[k];
    keep_in_vgprs(r);
    return r;
}

}  // namespace fpw
