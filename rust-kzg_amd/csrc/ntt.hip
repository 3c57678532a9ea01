// MI355X-native radix-2 NTT over Fr (B2): replaces FsFFTSettings::fft_fr
// (blst/src/fft_fr.rs:112-165) and das_fft_extension (blst/src/data_availability_sampling.rs:78-100).
//
// Contract identical to the reference: natural order in, natural order out, Montgomery blst_fr,
// roots taken with stride max_width/n from roots_of_unity (forward) or reverse_roots_of_unity
// (inverse), inverse scaled by n^-1.  The reference recurses (out-of-place DIT, even/odd split);
// here the same butterfly network is run iteratively on tiles of 4096 elements (ntt_plan.h):
//   n <= 4096 : one pass, a workgroup takes 4096 / n transforms, bit reversal folded into the load;
//   n  > 4096 : log2(n) stages split evenly over ceil(log2(n) / 10) passes (2^20 = 10 + 10): the first takes, per tile,
//               4096 >> T blocks of the bit-reversed sequence whose pieces are neighbours in memory, the later ones
//               2^T rows x (4096 >> T) >= 4 consecutive columns in place — every global access is a run of >= 128
//               bytes and every element crosses HBM once per pass (64*n algorithmic bytes each).
// Inside a tile: 1024 threads x 4 elements, radix-4 rounds at <= 128 VGPRs (4 waves per SIMD); the first round
// loads from global memory into registers and the last stores from them; a wave keeps the same 256 elements for up
// to 6 stages, so its exchanges need no workgroup barrier (one barrier per 4096-point transform).
// Field arithmetic: 9 x 29-bit Montgomery with lazy butterflies (fr29.hip.h); integer VALU only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "ckzg_internal.h"
#include "ff.hip.h"
#include "fr29.hip.h"
#include "config.h"
#include "device_guard.h"
#include "ntt_internal.h"
#include "ntt_plan.h"

using ff::Fr;
using fr29::Fe;
using ff::u32;
using ff::u64;

namespace {

using nttplan::KIND_A1;
using nttplan::KIND_A2;
using nttplan::KIND_B;
constexpr int LOGT = nttplan::LOGT;
constexpr int TILE = nttplan::TILE;  // elements per workgroup
constexpr int NT = nttplan::NT;      // threads per workgroup: 4 elements each, 16 waves = 4 per SIMD at <= 128 VGPRs
constexpr int PLAN_DAS = 3;          // plan-cache key of the fused DAS plans (next to the three pass kinds)
#ifndef KZGAMD_NTT_NT_STORES
#define KZGAMD_NTT_NT_STORES 1
#endif
constexpr bool nt_stores = KZGAMD_NTT_NT_STORES != 0;
#ifndef KZGAMD_NTT_DUAL_CHAINS
#define KZGAMD_NTT_DUAL_CHAINS 1
#endif
constexpr bool dual_chains = KZGAMD_NTT_DUAL_CHAINS != 0;  // the two products of a butterfly pair interleaved (fr29::mul_signed2)

// LDS holds the tile in the 9 x 29-bit form, limb-major: sh[limb * TILE + swz(idx)] (144 KiB of the 160 KiB).
__device__ __forceinline__ Fe lds_get(const u32* sh, u32 sidx) {  // sidx already swizzled
    Fe r;
#pragma unroll
    for (int k = 0; k < fr29::L; ++k) r.v[k] = sh[k * TILE + sidx];
    return r;
}
__device__ __forceinline__ void lds_put(u32* sh, u32 sidx, const Fe& a) {
#pragma unroll
    for (int k = 0; k < fr29::L; ++k) sh[k * TILE + sidx] = a.v[k];
}

// A pass's results are streamed (non-temporal stores): they drain towards HBM while other waves still compute
// instead of sitting dirty in L2 until the kernel's end.
__device__ __forceinline__ void store_fr(Fr* p, const Fr& v) {
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const v4u lo = {v.v[0], v.v[1], v.v[2], v.v[3]}, hi = {v.v[4], v.v[5], v.v[6], v.v[7]};
    v4u* q = reinterpret_cast<v4u*>(p);
    if (nt_stores) {
        __builtin_nontemporal_store(lo, q);
        __builtin_nontemporal_store(hi, q + 1);
    } else {
        q[0] = lo;
        q[1] = hi;
    }
}

struct RoundDev {
    u32 bit;      // a thread's elements are idxA, idxA | bit, idxB, idxB | bit
    u32 sbit;     // swz(bit): the swizzle is XOR-linear
    u32 pos;      // first stage of the round (tile-local)
    u32 M;        // stages in the round: 2, 1, or 0 (n = 1)
    u32 barrier;  // the next round belongs to another phase: workgroup barrier instead of the wave-local exchange
    u32 flags;    // fused DAS plans: bit 0 = forward half (twiddles from tw2), bit 1 = unit twiddles at position 0,
                  // bit 2 = multiply the results by the twist before they go to LDS
};

struct PassParams {
    RoundDev rd[nttplan::MAXR_DAS];
    const uint2* tab;  // [round][thread] = {idxA | idxB << 16, swz(idxA) | swz(idxB) << 16}   (ntt_plan.h)
    const Fe* tw;      // stage-major twiddles of this direction: entry (2^s - 1) + j = w_{2^(s+1)}^(+-j), times 2^261
    const Fe* tw2;     // fused DAS plans: the forward table (tw is the inverse one)
    size_t total;      // elements in the batch
    u32 n;             // transform length
    int kind, T, nrounds;
    int s0;            // global stage of the tile's stage 0 (KIND_B; 0 otherwise)
    int Lh;            // KIND_A2: log2(n) - T
    u32 tiles_per_xform;
    int last;          // last pass of the transform
    int epilogue;      // what the last pass does to a result x (< 64r, lazy): 0 = bring it to [0, r),
                       // 1 = x * scale, 2 = x * post[j * post_stride] for output position j of its transform
    Fr scale;          // epilogue 1: multiplier * 2^261 (an inverse transform's n^-1)
    const Fe* post;    // epilogue 2: multipliers * 2^261 as 9 x 29-bit limbs (the DAS extension's twist by the 2n-th roots)
    u32 post_stride;
};

// Uniform per-workgroup state of a pass: where the tile sits in the batch.
template <int KIND>
struct TileGeo {
    size_t xbase = 0, tile0 = 0, origin = 0;
    u32 o_base = 0, lo0 = 0, mT = 0;
    int T = 0, s0 = 0, Lh = 0;
    __device__ __forceinline__ TileGeo(const PassParams& P) {
        T = P.T;
        s0 = P.s0;
        Lh = P.Lh;
        mT = (1u << T) - 1u;
        if (KIND == KIND_A1) {
            tile0 = (size_t)blockIdx.x * TILE;
        } else {
            const u32 xf = blockIdx.x / P.tiles_per_xform, g = blockIdx.x % P.tiles_per_xform;
            xbase = (size_t)xf * P.n;
            if (KIND == KIND_A2) {
                o_base = g << (LOGT - T);
            } else {
                const int logC = LOGT - T;
                const u32 lo_tiles = (1u << s0) >> logC;  // tiles per block of 2^(s0 + T) positions
                const u32 hi = g / lo_tiles;
                lo0 = (g % lo_tiles) << logC;
                origin = ((size_t)hi << (s0 + T)) + lo0;
            }
        }
    }
    __device__ __forceinline__ u32 brev_t(u32 p) const { return T ? (__builtin_bitreverse32(p) >> (32 - T)) : 0u; }
    // where tile element idx comes from / goes to
    __device__ __forceinline__ size_t src_index(u32 idx) const {
        const u32 p = idx & mT, c = idx >> T;
        if (KIND == KIND_A1) return tile0 + ((size_t)c << T) + brev_t(p);
        if (KIND == KIND_A2) return xbase + o_base + c + ((size_t)brev_t(p) << Lh);
        return xbase + origin + ((size_t)p << s0) + c;
    }
    __device__ __forceinline__ size_t dst_index(u32 idx) const {
        const u32 p = idx & mT, c = idx >> T;
        if (KIND == KIND_A1) return tile0 + idx;
        if (KIND == KIND_A2) return xbase + ((size_t)(__builtin_bitreverse32(o_base + c) >> (32 - Lh)) << T) + p;
        return xbase + origin + ((size_t)p << s0) + c;
    }
    // twiddle of tile stage s for the pair whose lower element is i: global stage s0 + s, position
    // (i mod 2^s) * 2^s0 + column  ->  entry (2^(s0+s) - 1) + that
    __device__ __forceinline__ u32 tw_ent(u32 s, u32 i) const {
        if (KIND != KIND_B) return ((1u << s) - 1u) + (i & ((1u << s) - 1u));
        return ((1u << (s0 + s)) - 1u) + ((i & ((1u << s) - 1u)) << s0) + lo0 + (i >> T);
    }
};

// One round of a pass on the four elements of a thread.  FIRST: the elements come from global memory (and, for a
// transform's first pass, stages 0 and 1 multiply by w^0 = 1 at position 0: no multiplication); LAST: they go back
// to it.  V: how a butterfly multiplies — 1 = subtractive Montgomery steps on one accumulator chain
// (fr29::mul_signed), the only form launched; 2 = the same left to the compiler's re-association and 0 = round 2's
// additive multiplier were measured and lost (profiles/NOTES.md), and are instantiated by tools/ only.
template <int KIND, int V, bool FIRST, bool LAST, bool DAS = false>
__device__ __forceinline__ void ntt_round(u32* sh, Fr* __restrict__ out, const Fr* __restrict__ in, const PassParams& P,
                                          const TileGeo<KIND>& G, int r, const uint2 te) {
#define KZG_BF(K0, K1, W)                                      \
    {                                                          \
        if constexpr (V == 0) {                                \
            const Fe tt_ = fr29::mul(e[K1], W);                \
            fr29::butterfly_lazy(e[K0], e[K1], tt_);           \
        } else {                                               \
            const Fe tt_ = fr29::mul_signed<V == 1>(e[K1], W); \
            fr29::butterfly_signed(e[K0], e[K1], tt_);         \
        }                                                      \
    }
/* two butterflies whose products run side by side (fr29::mul_signed2) */
#define KZG_BF2(A0, A1, WA, B0, B1, WB)                                       \
    {                                                                        \
        if constexpr (V == 1 && dual_chains) {                               \
            Fe ta_, tb_;                                                     \
            fr29::mul_signed2<true>(ta_, tb_, e[A1], WA, e[B1], WB);         \
            fr29::butterfly_signed(e[A0], e[A1], ta_);                       \
            fr29::butterfly_signed(e[B0], e[B1], tb_);                       \
        } else {                                                             \
            KZG_BF(A0, A1, WA)                                               \
            KZG_BF(B0, B1, WB)                                               \
        }                                                                    \
    }
/* twiddle w^0 = 1 on a normalised operand */
#define KZG_BF1(K0, K1)                                  \
    {                                                    \
        const Fe y_ = e[K1];                             \
        fr29::butterfly_lazy(e[K0], e[K1], y_);          \
    }
/* twiddle w^0 = 1 on a lazy sum (< 4r): normalise it first, pad with 8r */
#define KZG_BF1N(K0, K1)                                 \
    {                                                    \
        Fe y_ = e[K1];                                   \
        fr29::norm(y_);                                  \
        fr29::butterfly_lazy8(e[K0], e[K1], y_);         \
    }
    const RoundDev rd = P.rd[r];
    const u32 iA = te.x & 0xffffu, iB = te.x >> 16, sA = te.y & 0xffffu, sB = te.y >> 16;
    Fe e[4];
    if constexpr (FIRST) {
        const u32 idx[4] = {iA, iA | rd.bit, iB, iB | rd.bit};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t gi = G.src_index(idx[k]);
            e[k] = fr29::unpack((KIND != KIND_A1 || gi < P.total) ? in[gi] : Fr::zero());
        }
    } else {
        e[0] = lds_get(sh, sA);
        e[1] = lds_get(sh, sA ^ rd.sbit);
        e[2] = lds_get(sh, sB);
        e[3] = lds_get(sh, sB ^ rd.sbit);
    }
    if (rd.M) {
        // a fused DAS plan starts a transform twice: its unit rounds are a run-time property of the round
        const bool unit = DAS ? (rd.flags & 2u) != 0 : (FIRST && KIND != KIND_B);
        const Fe* tw = DAS && (rd.flags & 1u) ? P.tw2 : P.tw;
        if (unit) {
            KZG_BF1(0, 1)
            KZG_BF1(2, 3)
        } else {
            // M = 2: both pairs share the twiddle (idxB = idxA | 2 << pos); M = 1: two unrelated pairs
            const Fe w = tw[G.tw_ent(rd.pos, iA)];
            const Fe w2 = tw[G.tw_ent(rd.pos, iB)];
            KZG_BF2(0, 1, w, 2, 3, w2)
        }
        if (rd.M == 2) {
            const Fe w1 = tw[G.tw_ent(rd.pos + 1, iA | rd.bit)];
            if (unit) {
                KZG_BF1N(0, 2)
                KZG_BF(1, 3, w1)
            } else {
                const Fe w0 = tw[G.tw_ent(rd.pos + 1, iA)];
                KZG_BF2(0, 2, w0, 1, 3, w1)
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) fr29::norm(e[k]);
    }
    if constexpr (DAS && !LAST) {
        if (rd.flags & 4u) {
            // the twist of the DAS extension: result j of the inverse transform times w^j (the 2n-th root), made
            // positive (+ r) and renormalised: an input of the forward half like any other (value < 2r)
            const u32 idx[4] = {iA, iA | rd.bit, iB, iB | rd.bit};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                e[k] = fr29::mul_signed<V != 2>(e[k], P.post[(size_t)(idx[k] & G.mT) * P.post_stride]);
#pragma unroll
                for (int l = 0; l < fr29::L; ++l) e[k].v[l] += fr29::rl(l);
                fr29::norm(e[k]);
            }
        }
    }
    if constexpr (LAST) {
        const u32 idx[4] = {iA, iA | rd.bit, iB, iB | rd.bit};
        // an inverse transform multiplies by n^-1 at the end of its last pass (the DAS extension by a per-position
        // factor); everything else only needs the lazy value (< 64r) brought back to [0, r)
        if (P.last && P.epilogue == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t gi = G.dst_index(idx[k]);
                if (KIND != KIND_A1 || gi < P.total)
                    store_fr(out + gi, fr29::finish(e[k], P.post[(size_t)((u32)gi & (P.n - 1u)) * P.post_stride]));
            }
        } else if (P.last && P.epilogue == 1) {
            const Fe fin = fr29::unpack(P.scale);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t gi = G.dst_index(idx[k]);
                if (KIND != KIND_A1 || gi < P.total) store_fr(out + gi, fr29::finish(e[k], fin));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t gi = G.dst_index(idx[k]);
                if (KIND != KIND_A1 || gi < P.total) store_fr(out + gi, fr29::reduce_lazy(e[k]));
            }
        }
    } else {
        lds_put(sh, sA, e[0]);
        lds_put(sh, sA ^ rd.sbit, e[1]);
        lds_put(sh, sB, e[2]);
        lds_put(sh, sB ^ rd.sbit, e[3]);
        if (rd.barrier) {
            __syncthreads();
        } else {
            // the next round reads what other lanes of this wave have just written: LDS operations of one wave
            // execute in issue order; the fences only keep the compiler from reordering across this point
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
#undef KZG_BF2
#undef KZG_BF
#undef KZG_BF1
#undef KZG_BF1N
}

// One pass: T stages (0..12) of the DIT network on a tile of 4096 elements (ntt_plan.h has the geometry).
//   Round 0 loads its elements straight from global memory, the last round stores straight to it; in between the
//   tile lives in LDS.  Rounds of one phase exchange data between the lanes of a wave only — LDS operations of a wave
//   execute in order, no barrier — so a wave that has its data starts computing while others still wait for theirs,
//   and a 4096-point transform synchronises the workgroup once (between stages 5 and 6).
//   Butterflies are lazy (fr29.hip.h): a round's inputs have normalised limbs, its outputs are renormalised once;
//   values grow by < 5r per stage (8r for the multiplication-free stage 1 of a transform): < 61r after 12 stages.
template <int KIND, int V>
__global__ void __launch_bounds__(NT) k_ntt_pass(Fr* __restrict__ out, const Fr* __restrict__ in, const PassParams P) {
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 tid = threadIdx.x;
    const TileGeo<KIND> G(P);
    const int n = P.nrounds;  // >= 2 (ntt_plan.h)
    // a round's table entry is fetched one round ahead: the round starts with its LDS reads and twiddle loads, not
    // with a dependent global load
    uint2 te = P.tab[tid], nx = P.tab[NT + tid];
    ntt_round<KIND, V, true, false>(sh, out, in, P, G, 0, te);
    for (int r = 1; r < n - 1; ++r) {
        te = nx;
        nx = P.tab[(r + 1) * NT + tid];
        ntt_round<KIND, V, false, false>(sh, out, in, P, G, r, te);
    }
    ntt_round<KIND, V, false, true>(sh, out, in, P, G, n - 1, nx);
}

}  // namespace

namespace {

int ilog2(size_t n) {
    int l = 0;
    while (((size_t)1 << l) < n) ++l;
    return l;
}

Fr times32(Fr a) {  // a * 2^5 mod r
    for (int k = 0; k < 5; ++k) a = ff::add(a, a);
    return a;
}
Fr one261() {  // 2^261 mod r = (2^256 mod r) * 2^5
    return times32(Fr::one());
}

// n^-1 in Montgomery form
Fr inv_len(size_t n) {
    Fr v = Fr::zero();
    v.v[0] = (u32)n;
    v.v[1] = (u32)((u64)n >> 32);
    return ff::inverse_bgcd(ff::to_mont(v));
}

// The DAS extension of lists of <= 4096 elements in one pass (nttplan::make_das_plan): inverse rounds, twist, forward
// rounds, the tile never leaves the CU in between.  Same round code as k_ntt_pass; the middle rounds carry their
// unit / twist / direction flags at run time.
template <int V>
__global__ void __launch_bounds__(NT) k_das_fused(Fr* __restrict__ out, const Fr* __restrict__ in, const PassParams P) {
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 tid = threadIdx.x;
    const TileGeo<KIND_A1> G(P);
    const int n = P.nrounds;  // >= 4
    uint2 te = P.tab[tid], nx = P.tab[NT + tid];
    ntt_round<KIND_A1, V, true, false, true>(sh, out, in, P, G, 0, te);
    for (int r = 1; r < n - 1; ++r) {
        te = nx;
        nx = P.tab[(r + 1) * NT + tid];
        ntt_round<KIND_A1, V, false, false, true>(sh, out, in, P, G, r, te);
    }
    ntt_round<KIND_A1, V, false, true, true>(sh, out, in, P, G, n - 1, nx);
}

// the device copy of one plan: the round table and the per-round constants
void upload_plan(NttCtx* ctx, int kind, int T) {
    const bool das = kind == PLAN_DAS;
    const nttplan::Plan pl = das ? nttplan::make_das_plan(T) : nttplan::make_plan(kind, T);
    NttPlanDev pd;
    pd.nrounds = pl.nrounds;
    for (int r = 0; r < pl.nrounds; ++r) {
        pd.rd[r][0] = 1u << pl.elem_bit(r);
        pd.rd[r][1] = pl.sbit[r];
        pd.rd[r][2] = (u32)pl.rounds[r].pos;
        pd.rd[r][3] = (u32)pl.rounds[r].M;
        pd.rd[r][4] = (u32)pl.rounds[r].barrier_after;
        pd.rd[r][5] = (u32)(pl.rounds[r].part | pl.rounds[r].unit << 1 | pl.rounds[r].twist << 2);
    }
    // tab[..][4] of u16 = {idxA, idxB, lds(idxA), lds(idxB)} is read as uint2 {idxA | idxB << 16, ldsA | ldsB << 16}
    NTT_TRY(hipMalloc(&pd.d_tab, pl.tab.size() * sizeof(uint16_t)));
    NTT_TRY(hipMemcpy(pd.d_tab, pl.tab.data(), pl.tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    ctx->plans[kind * 16 + T] = pd;
}

void launch_pass(NttCtx* ctx, int kind, int T, Fr* d_out, const Fr* d_in, PassParams& P, unsigned grid, hipStream_t stream) {
    const NttPlanDev& pd = ctx->plans.at(kind * 16 + T);
    P.kind = kind == PLAN_DAS ? (int)KIND_A1 : kind;
    P.T = T;
    P.nrounds = pd.nrounds;
    for (int r = 0; r < pd.nrounds; ++r) {
        P.rd[r].bit = pd.rd[r][0];
        P.rd[r].sbit = pd.rd[r][1];
        P.rd[r].pos = pd.rd[r][2];
        P.rd[r].M = pd.rd[r][3];
        P.rd[r].barrier = pd.rd[r][4];
        P.rd[r].flags = pd.rd[r][5];
    }
    P.tab = (const uint2*)pd.d_tab;
    const size_t lds = (size_t)TILE * sizeof(u32) * fr29::L;
#define KZG_LAUNCH(K, V) hipLaunchKernelGGL((k_ntt_pass<K, V>), dim3(grid), dim3(NT), lds, stream, d_out, d_in, P)
#define KZG_LAUNCH_V(K) KZG_LAUNCH(K, 1)
    if (kind == PLAN_DAS) {
        hipLaunchKernelGGL(k_das_fused<1>, dim3(grid), dim3(NT), lds, stream, d_out, d_in, P);
    } else if (kind == KIND_A1) {
        KZG_LAUNCH_V(KIND_A1);
    } else if (kind == KIND_A2) {
        KZG_LAUNCH_V(KIND_A2);
    } else {
        KZG_LAUNCH_V(KIND_B);
    }
#undef KZG_LAUNCH_V
#undef KZG_LAUNCH
}

// what the last pass multiplies the results by
enum NttScale {
    SCALE_PLAIN = 0,    // forward transform: nothing; inverse transform: n^-1
    SCALE_FWD_NINV = 1, // forward transform times n^-1 (second half of the DAS extension)
    SCALE_INV_TWIST = 2 // inverse transform WITHOUT n^-1, result j times the 2n-th root w^j (first half of it)
};

// enqueue nbatch transforms of length n (device pointers; d_out must not alias d_in)
void ntt_enqueue(NttCtx* ctx, Fr* d_out, const Fr* d_in, size_t n, size_t nbatch, bool inverse, hipStream_t stream,
                 NttScale mode = SCALE_PLAIN) {
    PassParams P;
    memset(&P, 0, sizeof(P));
    const int logn = ilog2(n);
    P.n = (u32)n;
    // an inverse transform folds n^-1 into its last pass: data is d*2^256, so the multiplier n^-1*2^261 is
    // (n^-1 in blst Montgomery form) * 2^5
    if (mode == SCALE_INV_TWIST) {
        P.epilogue = 2;
        P.post = (const Fe*)ctx->d_roots;  // roots_of_unity[i] * 2^261, natural order, W + 1 entries
        P.post_stride = (u32)(ctx->W / (2 * n));
    } else if (inverse || mode == SCALE_FWD_NINV) {
        P.epilogue = 1;
        P.scale = times32(inv_len(n));
    }
    P.total = n * nbatch;
    P.tw = (const Fe*)(inverse ? ctx->d_tw_inv : ctx->d_tw_fwd);
    if (logn <= LOGT) {
        // the whole transform in one pass; a tile takes 4096 / n consecutive transforms
        P.last = 1;
        launch_pass(ctx, KIND_A1, logn, d_out, d_in, P, (unsigned)((P.total + TILE - 1) / TILE), stream);
    } else {
        const std::vector<int> Ts = nttplan::split_passes(logn);
        const u32 tiles = (u32)(n >> LOGT);
        P.tiles_per_xform = tiles;
        P.Lh = logn - Ts[0];
        P.last = 0;
        launch_pass(ctx, KIND_A2, Ts[0], d_out, d_in, P, (unsigned)(tiles * nbatch), stream);
        int s0 = Ts[0];
        for (size_t i = 1; i < Ts.size(); ++i) {
            P.s0 = s0;
            P.last = i + 1 == Ts.size();
            launch_pass(ctx, KIND_B, Ts[i], d_out, d_out, P, (unsigned)(tiles * nbatch), stream);
            s0 += Ts[i];
        }
    }
    NTT_TRY(hipGetLastError());
}

}  // namespace

extern "C" void* kzgamd_ntt_new(unsigned scale) { return kzgamd_ntt_new_ex(scale, nullptr); }

extern "C" void* kzgamd_ntt_new_ex(unsigned scale, const KzgAmdConfig* cfg) {
    kzgamd::Options opt;
    std::string err;
    if (!kzgamd::Options::resolve(opt, cfg, &err)) {
        fprintf(stderr, "kzg_mi355x: kzgamd_ntt_new: %s\n", err.c_str());
        return nullptr;
    }
    return kzgamd::ntt_create(scale, opt);
}

void* kzgamd::ntt_create(unsigned scale, const kzgamd::Options& opt) {
    if (scale >= 32) return nullptr;  // "Scale is expected to be within root of unity matrix row size"
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "kzg_mi355x: kzgamd_ntt_new: no gfx950 device visible\n");
        return nullptr;
    }
    auto* ctx = new NttCtx();
    try {
        int cur_dev = 0;
        NTT_TRY(hipGetDevice(&cur_dev));
        kzgamd::DeviceGuard placed(opt.device >= 0 ? opt.device : cur_dev);  // the caller's device is restored on return
        NTT_TRY(placed.err);
        NTT_TRY(hipGetDevice(&ctx->device));
        NTT_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->scale = scale;
        ctx->W = (size_t)1 << scale;
        ctx->g1_wide_max = (size_t)opt.t[kzgamd::T_G1_WIDE_MAX];
        ctx->g1_quad_max = (size_t)opt.t[kzgamd::T_G1_QUAD_MAX];
        ctx->g1_pair_max = (size_t)opt.t[kzgamd::T_G1_PAIR_MAX];
        ctx->combine = opt.t[kzgamd::T_COMBINE] != 0;
        kzgamd::expand_roots(ctx->roots, scale);
        // device twiddles in the 2^261 domain, w*2^261 = (w*2^256) * 2^5, already sliced into the 9 x 29-bit limbs
        // the butterflies multiply with (36 bytes per root instead of 32, ~27 instructions less per butterfly)
        std::vector<Fe> tw(ctx->W + 1);
        for (size_t i = 0; i <= ctx->W; ++i) tw[i] = fr29::unpack(times32(ctx->roots[i]));
        NTT_TRY(hipMalloc(&ctx->d_roots, (ctx->W + 1) * sizeof(Fe)));
        NTT_TRY(hipMemcpy(ctx->d_roots, tw.data(), (ctx->W + 1) * sizeof(Fe), hipMemcpyHostToDevice));
        // stage-major copies for the butterfly kernels: [(2^s - 1) + j] = w_{2^(s+1)}^j = roots[j * (W >> (s+1))],
        // s < scale, and the same with the inverse roots (roots[W - idx]); W - 1 entries each
        if (scale > 0) {
            std::vector<Fe> sm(ctx->W), smi(ctx->W);
            for (unsigned s = 0; s < scale; ++s)
                for (size_t j = 0; j < ((size_t)1 << s); ++j) {
                    const size_t idx = j * (ctx->W >> (s + 1));
                    sm[((size_t)1 << s) - 1 + j] = tw[idx];
                    smi[((size_t)1 << s) - 1 + j] = tw[ctx->W - idx];
                }
            NTT_TRY(hipMalloc(&ctx->d_tw_fwd, ctx->W * sizeof(Fe)));
            NTT_TRY(hipMalloc(&ctx->d_tw_inv, ctx->W * sizeof(Fe)));
            NTT_TRY(hipMemcpy(ctx->d_tw_fwd, sm.data(), ctx->W * sizeof(Fe), hipMemcpyHostToDevice));
            NTT_TRY(hipMemcpy(ctx->d_tw_inv, smi.data(), ctx->W * sizeof(Fe), hipMemcpyHostToDevice));
        }
        // every plan the passes can ask for (35 tables of <= 48 KiB), built once: launches never allocate
        for (int T = 0; T <= nttplan::LOGT; ++T) upload_plan(ctx, KIND_A1, T);
        for (int T = 0; T <= 10; ++T) {
            upload_plan(ctx, KIND_A2, T);
            upload_plan(ctx, KIND_B, T);
        }
        for (int T = 0; T <= nttplan::LOGT; ++T) upload_plan(ctx, PLAN_DAS, T);
    } catch (...) {
        delete ctx;
        return nullptr;
    }
    return ctx;
}

extern "C" void kzgamd_ntt_free(void* vctx) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx) return;
    kzgamd::DeviceGuard on_device(ctx->device);
    delete ctx;
}

extern "C" int kzgamd_ntt_fr_device(void* vctx, void* d_out, const void* d_in, size_t n, size_t nbatch, int inverse,
                                    void* stream) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx || !d_out || !d_in) return -1;
    if (n > ctx->W) return 1;
    if (n == 0 || (n & (n - 1))) return 2;
    if (d_out == d_in) return -3;
    try {
        kzgamd::DeviceGuard on_device(ctx->device);
        NTT_TRY(on_device.err);
        ntt_enqueue(ctx, (Fr*)d_out, (const Fr*)d_in, n, nbatch, inverse != 0, (hipStream_t)stream);
    } catch (const NttErr& e) {
        return -(int)e.e - 100;
    }
    return 0;
}

namespace {
// ---- combining of concurrent host-buffer calls on one handle (see NttCtx::Combine) ----
struct SlotPtrs {
    uint4* p[32];  // NttCtx::COMB_MAX
};
__global__ void __launch_bounds__(256) k_slots_gather(uint4* __restrict__ dst, SlotPtrs slots, size_t per, size_t nreq) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nreq * per) dst[t] = slots.p[t / per][t % per];
}
__global__ void __launch_bounds__(256) k_slots_scatter(SlotPtrs slots, const uint4* __restrict__ src, size_t per, size_t nreq) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nreq * per) slots.p[t / per][t % per] = src[t];
}
void das_enqueue(NttCtx* ctx, Fr* d_odds, const Fr* d_evens, Fr* d_tmp, size_t n, size_t nbatch, hipStream_t stream);

// one batch on `lane`: every request has the same kind and length; sets rc of each
void run_combined_batch(NttCtx* ctx, NttCtx::CombLane& lane, const std::vector<NttCtx::HostCall*>& batch) {
    const size_t nb = batch.size(), n = batch[0]->n;
    const int kind = batch[0]->kind;
    int rc = 0;
    try {
        kzgamd::DeviceGuard on_device(ctx->device);
        NTT_TRY(on_device.err);
        // each on its own test: an allocation that failed is tried again by the next batch, not taken for made
        const size_t cap = NttCtx::COMB_MAX * NttCtx::COMB_NMAX * sizeof(Fr);
        if (!lane.st) NTT_TRY(hipStreamCreateWithFlags(&lane.st, hipStreamNonBlocking));
        if (!lane.d_in) NTT_TRY(hipMalloc(&lane.d_in, cap));
        if (!lane.d_out) NTT_TRY(hipMalloc(&lane.d_out, cap));
        if (!lane.d_tmp) NTT_TRY(hipMalloc(&lane.d_tmp, cap));
        SlotPtrs sp;
        for (size_t j = 0; j < NttCtx::COMB_MAX; ++j) sp.p[j] = j < nb ? (uint4*)batch[j]->slot : nullptr;
        const size_t per = n * 2, total = nb * per;  // 16-byte words per request
        const unsigned grid = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(k_slots_gather, dim3(grid), dim3(256), 0, lane.st, (uint4*)lane.d_in, sp, per, nb);
        if (kind == 2) das_enqueue(ctx, lane.d_out, lane.d_in, lane.d_tmp, n, nb, lane.st);
        else ntt_enqueue(ctx, lane.d_out, lane.d_in, n, nb, kind == 1, lane.st);
        hipLaunchKernelGGL(k_slots_scatter, dim3(grid), dim3(256), 0, lane.st, sp, (const uint4*)lane.d_out, per, nb);
        NTT_TRY(hipGetLastError());
        NTT_TRY(hipStreamSynchronize(lane.st));
    } catch (const NttErr& e) {
        if (lane.st) (void)hipStreamSynchronize(lane.st);  // nothing may still read or write the callers' slots
        rc = -(int)e.e - 100;
    }
    for (auto* r : batch) r->rc = rc;
}

// returns false when the call cannot be combined (no slot): the caller takes the plain path
bool run_host_combined(NttCtx* ctx, void* out, const void* in, size_t n, int kind, int* rc_out) {
    NttCtx::HostCall me{out, in, n, kind};
    auto& q = ctx->comb;
    std::vector<NttCtx::HostCall*> batch;
    batch.reserve(NttCtx::COMB_MAX);
    std::unique_lock<std::mutex> lk(q.mu);
    if (q.free_slots.empty() && !q.pinned_failed && q.slots_allocated < NttCtx::COMB_SLOTS) {
        kzgamd::DeviceGuard on_device(ctx->device);
        q.slot_bytes = NttCtx::COMB_NMAX * sizeof(Fr);
        int grow = q.slots_allocated < 8 ? 4 : q.slots_allocated < 16 ? 8 : 16;
        if (grow > NttCtx::COMB_SLOTS - q.slots_allocated) grow = NttCtx::COMB_SLOTS - q.slots_allocated;
        unsigned char* chunk = nullptr;
        if (on_device.err != hipSuccess ||
            hipHostMalloc((void**)&chunk, (size_t)grow * q.slot_bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
            q.pinned_failed = true;
            (void)hipGetLastError();
        } else {
            q.slot_chunks.push_back(chunk);
            q.slots_allocated += grow;
            for (int i = grow; i-- > 0;) q.free_slots.push_back(chunk + (size_t)i * q.slot_bytes);
        }
    }
    if (q.free_slots.empty()) return false;
    me.slot = q.free_slots.back();
    q.free_slots.pop_back();
    lk.unlock();
    memcpy(me.slot, in, n * sizeof(Fr));
    lk.lock();
    q.pending.push_back(&me);
    while (!me.done) {
        if (q.leaders < NttCtx::COMB_LANES && !q.pending.empty()) {
            ++q.leaders;
            NttCtx::CombLane* lane = nullptr;
            for (auto& l : q.lanes)
                if (!l.busy) {
                    lane = &l;
                    break;
                }
            lane->busy = true;
            while (!q.pending.empty() && !me.done) {
                batch.clear();
                const size_t bn = q.pending.front()->n;
                const int bk = q.pending.front()->kind;
                for (auto it = q.pending.begin(); it != q.pending.end() && batch.size() < NttCtx::COMB_MAX;) {
                    if ((*it)->n == bn && (*it)->kind == bk) {
                        batch.push_back(*it);
                        it = q.pending.erase(it);
                    } else {
                        ++it;
                    }
                }
                lk.unlock();
                run_combined_batch(ctx, *lane, batch);
                lk.lock();
                for (auto* r : batch) r->done = true;
                q.cv.notify_all();
            }
            lane->busy = false;
            --q.leaders;
            q.cv.notify_all();
        } else {
            q.cv.wait(lk);
        }
    }
    lk.unlock();
    if (me.rc == 0) memcpy(out, me.slot, n * sizeof(Fr));
    lk.lock();
    q.free_slots.push_back(me.slot);
    lk.unlock();
    *rc_out = me.rc;
    return true;
}
}  // namespace

extern "C" int ntt_fr(void* vctx, blst_fr* out, const blst_fr* in, size_t n, int inverse) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx || !out || !in) return -1;
    if (n > ctx->W) return 1;                 // "Supplied list is longer than the available max width"
    if (n == 0 || (n & (n - 1))) return 2;    // "A list with power-of-two length expected"
    if (ctx->combine && n <= NttCtx::COMB_NMAX) {
        int rc = 0;
        if (run_host_combined(ctx, out, in, n, inverse != 0 ? 1 : 0, &rc)) return rc;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    try {
        kzgamd::DeviceGuard on_device(ctx->device);
        NTT_TRY(on_device.err);
        ctx->ensure(n);
        NTT_TRY(hipMemcpyAsync(ctx->d_a, in, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        ntt_enqueue(ctx, ctx->d_b, ctx->d_a, n, 1, inverse != 0, ctx->stream);
        NTT_TRY(hipMemcpyAsync(out, ctx->d_b, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        NTT_TRY(hipStreamSynchronize(ctx->stream));
    } catch (const NttErr& e) {
        return -(int)e.e - 100;
    }
    return 0;
}

// odds = FFT_n( w^j * IFFT_n(evens)_j ), w the 2n-th root: the values of the degree < n interpolant of
// `evens` at the odd positions of the size-2n domain — what das_fft_extension_stride computes with its
// fused butterfly network (data_availability_sampling.rs:14-72), including the final n^-1 (:95-97).  Like the
// reference's network this costs two half-size transforms and nothing else: the twist by w^j rides in the inverse
// transform's last-pass multiplication (instead of its n^-1), the n^-1 in the forward transform's.
namespace {
void das_enqueue(NttCtx* ctx, Fr* d_odds, const Fr* d_evens, Fr* d_tmp, size_t n, size_t nbatch, hipStream_t stream) {
    if (n <= (size_t)TILE) {
        // lists of <= 4096 elements: both transforms in ONE pass over the tile (nttplan::make_das_plan) — the data
        // crosses HBM once instead of twice and the tile's load / store skeleton is paid once
        PassParams P;
        memset(&P, 0, sizeof(P));
        P.n = (u32)n;
        P.total = n * nbatch;
        P.tw = (const Fe*)ctx->d_tw_inv;
        P.tw2 = (const Fe*)ctx->d_tw_fwd;
        P.post = (const Fe*)ctx->d_roots;
        P.post_stride = (u32)(ctx->W / (2 * n));
        P.epilogue = 1;
        P.scale = times32(inv_len(n));
        P.last = 1;
        launch_pass(ctx, PLAN_DAS, ilog2(n), d_odds, d_evens, P, (unsigned)((P.total + TILE - 1) / TILE), stream);
        NTT_TRY(hipGetLastError());
        return;
    }
    ntt_enqueue(ctx, d_tmp, d_evens, n, nbatch, true, stream, SCALE_INV_TWIST);
    ntt_enqueue(ctx, d_odds, d_tmp, n, nbatch, false, stream, SCALE_FWD_NINV);
}
}  // namespace

extern "C" int das_fft_extension(void* vctx, blst_fr* odds, const blst_fr* evens, size_t n) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx || !odds || !evens) return -1;
    if (n == 0) return 1;                  // "A non-zero list ab expected"
    if (n & (n - 1)) return 2;             // "A list with power-of-two length expected"
    if (n * 2 > ctx->W) return 3;          // "Supplied list is longer than the available max width"
    if (ctx->combine && n <= NttCtx::COMB_NMAX) {
        int rc = 0;
        if (run_host_combined(ctx, odds, evens, n, 2, &rc)) return rc;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    try {
        kzgamd::DeviceGuard on_device(ctx->device);
        NTT_TRY(on_device.err);
        ctx->ensure(2 * n);
        NTT_TRY(hipMemcpyAsync(ctx->d_a, evens, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        das_enqueue(ctx, ctx->d_a + n, ctx->d_a, ctx->d_b, n, 1, ctx->stream);
        NTT_TRY(hipMemcpyAsync(odds, ctx->d_a + n, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        NTT_TRY(hipStreamSynchronize(ctx->stream));
    } catch (const NttErr& e) {
        return -(int)e.e - 100;
    }
    return 0;
}

// Device-resident form: nbatch contiguous half-size lists in d_evens -> d_odds; d_scratch holds n * nbatch elements
// and must differ from both (d_odds may not alias d_evens either).  Same error codes as das_fft_extension.
extern "C" int kzgamd_das_fft_extension_device(void* vctx, void* d_odds, const void* d_evens, void* d_scratch, size_t n,
                                               size_t nbatch, void* stream) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx || !d_odds || !d_evens || !d_scratch) return -1;
    if (n == 0) return 1;
    if (n & (n - 1)) return 2;
    if (n * 2 > ctx->W) return 3;
    if (d_odds == d_evens || d_scratch == d_evens || d_scratch == d_odds) return -3;
    try {
        kzgamd::DeviceGuard on_device(ctx->device);
        NTT_TRY(on_device.err);
        das_enqueue(ctx, (Fr*)d_odds, (const Fr*)d_evens, (Fr*)d_scratch, n, nbatch, (hipStream_t)stream);
    } catch (const NttErr& e) {
        return -(int)e.e - 100;
    }
    return 0;
}

// The tile plan of (kind, T) as the kernel receives it — no GPU needed: rounds[r] = {pos, M, barrier_after, element bit},
// tab[(r * 1024 + thread) * 4 ..] = {idxA, idxB, swz(idxA), swz(idxB)}.  Returns the number of rounds, -1 on bad arguments.
extern "C" int kzgamd_ntt_plan_dump(int kind, int T, int* rounds /* 6 x 4 */, uint16_t* tab /* 6 x 1024 x 4 */) {
    if (kind < 0 || kind > 2 || T < 0 || T > (kind == nttplan::KIND_A1 ? nttplan::LOGT : 10)) return -1;
    const nttplan::Plan pl = nttplan::make_plan(kind, T);
    for (int r = 0; r < pl.nrounds; ++r) {
        if (rounds) {
            rounds[4 * r + 0] = pl.rounds[r].pos;
            rounds[4 * r + 1] = pl.rounds[r].M;
            rounds[4 * r + 2] = pl.rounds[r].barrier_after;
            rounds[4 * r + 3] = pl.elem_bit(r);
        }
    }
    if (tab) memcpy(tab, pl.tab.data(), pl.tab.size() * sizeof(uint16_t));
    return pl.nrounds;
}

extern "C" int kzgamd_ntt_das_plan_dump(int T, int* rounds /* 12 x 6 */, uint16_t* tab /* 12 x 1024 x 4 */) {
    if (T < 0 || T > nttplan::LOGT) return -1;
    const nttplan::Plan pl = nttplan::make_das_plan(T);
    for (int r = 0; r < pl.nrounds; ++r) {
        if (rounds) {
            rounds[6 * r + 0] = pl.rounds[r].pos;
            rounds[6 * r + 1] = pl.rounds[r].M;
            rounds[6 * r + 2] = pl.rounds[r].barrier_after;
            rounds[6 * r + 3] = pl.elem_bit(r);
            rounds[6 * r + 4] = pl.rounds[r].part | pl.rounds[r].unit << 1 | pl.rounds[r].twist << 2;
            rounds[6 * r + 5] = pl.sbit[r];
        }
    }
    if (tab) memcpy(tab, pl.tab.data(), pl.tab.size() * sizeof(uint16_t));
    return pl.nrounds;
}

extern "C" int kzgamd_ntt_roots(void* vctx, blst_fr* roots, blst_fr* reverse_roots, blst_fr* brp_roots) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx) return -1;
    const size_t W = ctx->W;
    if (roots) memcpy(roots, ctx->roots.data(), (W + 1) * sizeof(Fr));
    if (reverse_roots)
        for (size_t i = 0; i <= W; ++i) memcpy(&reverse_roots[i], &ctx->roots[W - i], sizeof(Fr));
    if (brp_roots)
        for (size_t i = 0; i < W; ++i) {
            size_t r = 0;
            for (unsigned b = 0; b < ctx->scale; ++b)
                if (i & ((size_t)1 << b)) r |= (size_t)1 << (ctx->scale - 1 - b);
            memcpy(&brp_roots[i], &ctx->roots[r], sizeof(Fr));
        }
    return 0;
}
