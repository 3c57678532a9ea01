// MI355X-native radix-2 NTT over Fr (B2): replaces FsFFTSettings::fft_fr
// (blst/src/fft_fr.rs:112-165) and das_fft_extension (blst/src/data_availability_sampling.rs:78-100).
//
// Contract identical to the reference: natural order in, natural order out, Montgomery blst_fr,
// roots taken with stride max_width/n from roots_of_unity (forward) or reverse_roots_of_unity
// (inverse), inverse scaled by n^-1.  The reference recurses (out-of-place DIT, even/odd split);
// here the same butterfly network is run iteratively:
//   n <= 4096 : one workgroup per transform, all log2(n) stages in LDS (4096 x 32 B = 128 KiB of the
//               160 KiB CDNA4 LDS), bit-reversal folded into the load;
//   n  > 4096 : pass 1 = the 12 low stages on contiguous 4096-blocks (same kernel), then up to 12 further
//               stages per pass on strided tiles (C positions x R rows = 4096 elements per workgroup): two
//               passes up to 2^24, three up to the 2^31 the roots table allows; every element crosses HBM
//               once per pass (64*n algorithmic bytes each).
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
#include "device_guard.h"
#include "ntt_internal.h"

using ff::Fr;
using fr29::Fe;
using ff::u32;
using ff::u64;

namespace {

constexpr int LOG_TILE = 12;
constexpr int TILE = 1 << LOG_TILE;  // elements per workgroup
// threads per workgroup = TILE >> MAXM: 2^MAXM elements per thread in a round of MAXM stages
//   MAXM = 3 (radix-8 rounds, 512 threads, 2 waves per SIMD: the 144 KiB tile allows one workgroup per CU)
//   MAXM = 2 (radix-4 rounds, 1024 threads, 4 waves per SIMD, half the registers, 1.5x the LDS traffic) was measured
//            7 % slower (98.8 vs 92.4 us for 256 transforms of 4096) and is not instantiated

__device__ __forceinline__ u32 brev(u32 v, int bits) { return bits == 0 ? 0u : __builtin_bitreverse32(v) >> (32 - bits); }

// LDS holds a tile in the 9 x 29-bit form, limb-major: sh[limb * cnt + swz(idx)].  The XOR swizzle folds
// index bits 5..11 into the bank bits, so that every access pattern of the kernels below — unit stride, the
// strides 8 / 64 / 512 of the radix-8 rounds, and the bit-reversed scatter of the loads — puts the 32 lanes of a
// ds_read_b32 / ds_write_b32 group on 32 distinct banks (4096 x 36 B = 144 KiB of the 160 KiB LDS).
__host__ __device__ constexpr int swz(int i) { return i ^ (((i >> 5) & 7) ^ (((i >> 6) & 3) << 3) ^ ((i >> 7) & 31)); }

__device__ __forceinline__ Fe lds_get(const u32* sh, int cnt, int sidx) {  // sidx already swizzled
    Fe r;
#pragma unroll
    for (int k = 0; k < fr29::L; ++k) r.v[k] = sh[k * cnt + sidx];
    return r;
}
__device__ __forceinline__ void lds_put(u32* sh, int cnt, int sidx, const Fe& a) {
#pragma unroll
    for (int k = 0; k < fr29::L; ++k) sh[k * cnt + sidx] = a.v[k];
}

// One round = M consecutive butterfly stages (B .. B+M-1 of the tile's DIT network) on 2^M elements held in
// registers: one LDS round trip and one barrier per M stages instead of per stage.  Virtual thread v owns the
// tile indices  base | (k << B),  k < 2^M,  base = v with M zero bits inserted at position B.  Stage B+q pairs
// k and k | 2^q; its twiddle depends on the low B+q bits of the index, so the round loads 2^M - 1 twiddles for
// its M * 2^(M-1) butterflies.  FIRST: stage 0 of a transform multiplies by w^0 = 1 — no multiplication.
template <int NT, int M, int B, bool FIRST, class TwFn>
__device__ __forceinline__ void round_regs(u32* sh, int cnt, int nelem, TwFn tw) {
    constexpr int E = 1 << M;
    for (int v = threadIdx.x; v < (nelem >> M); v += NT) {
        const int lo = v & ((1 << B) - 1);
        const int base = ((v >> B) << (B + M)) | lo;
        const int sbase = swz(base);  // swz is XOR-linear: swz(base | k << B) = swz(base) ^ swz(k << B)
        Fe e[E];
#pragma unroll
        for (int k = 0; k < E; ++k) e[k] = lds_get(sh, cnt, sbase ^ swz(k << B));
        // butterflies written out per stage (no loop nests for the compiler to leave rolled: a rolled loop would
        // index e[] dynamically and push it to scratch)
        // limbs are renormalised after the second stage of a round and at its end, not after every stage
#define KZG_BF(K0, K1, W)                                \
    {                                                    \
        const Fe tt_ = fr29::mul(e[K1], W);              \
        fr29::butterfly_lazy(e[K0], e[K1], tt_);         \
    }
#define KZG_BF1(K0, K1)                                  \
    {                                                    \
        const Fe y_ = e[K1];                             \
        fr29::butterfly_lazy(e[K0], e[K1], y_);          \
    }
/* a later stage of the first round whose twiddle is w^0 = 1: the operand is a lazy sum (< 4r), normalise it first */
#define KZG_BF1N(K0, K1)                                 \
    {                                                    \
        Fe y_ = e[K1];                                   \
        fr29::norm(y_);                                  \
        fr29::butterfly_lazy8(e[K0], e[K1], y_);         \
    }
#define KZG_NORM_ALL                  \
    _Pragma("unroll") for (int k_ = 0; k_ < E; ++k_) fr29::norm(e[k_]);
        if constexpr (M == 1) {
            if constexpr (FIRST && B == 0) {
                KZG_BF1(0, 1)
            } else {
                const Fe w = tw(B, lo, base);
                KZG_BF(0, 1, w)
            }
            KZG_NORM_ALL
        } else if constexpr (M == 2) {
            if constexpr (FIRST && B == 0) {
                KZG_BF1(0, 1)
                KZG_BF1(2, 3)
            } else {
                const Fe w = tw(B, lo, base);
                KZG_BF(0, 1, w)
                KZG_BF(2, 3, w)
            }
            {
                if constexpr (FIRST && B == 0) {
                    KZG_BF1N(0, 2)  // stage 1, position 0: w^0 = 1 again
                } else {
                    const Fe w0 = tw(B + 1, lo, base);
                    KZG_BF(0, 2, w0)
                }
                const Fe w1 = tw(B + 1, lo | (1 << B), base);
                KZG_BF(1, 3, w1)
            }
            KZG_NORM_ALL
        } else {
            if constexpr (FIRST && B == 0) {
                KZG_BF1(0, 1)
                KZG_BF1(2, 3)
                KZG_BF1(4, 5)
                KZG_BF1(6, 7)
            } else {
                const Fe w = tw(B, lo, base);
                KZG_BF(0, 1, w)
                KZG_BF(2, 3, w)
                KZG_BF(4, 5, w)
                KZG_BF(6, 7, w)
            }
            {
                // the first round of a transform: position 0 of stages 1 and 2 has the twiddle w^0 = 1 as well
                if constexpr (FIRST && B == 0) {
                    KZG_BF1N(0, 2)
                    KZG_BF1N(4, 6)
                } else {
                    const Fe w0 = tw(B + 1, lo, base);
                    KZG_BF(0, 2, w0)
                    KZG_BF(4, 6, w0)
                }
                const Fe w1 = tw(B + 1, lo | (1 << B), base);
                KZG_BF(1, 3, w1)
                KZG_BF(5, 7, w1)
            }
            KZG_NORM_ALL
            {
                if constexpr (FIRST && B == 0) {
                    KZG_BF1N(0, 4)
                } else {
                    const Fe w0 = tw(B + 2, lo, base);
                    KZG_BF(0, 4, w0)
                }
                const Fe w1 = tw(B + 2, lo | (1 << B), base);
                KZG_BF(1, 5, w1)
                const Fe w2 = tw(B + 2, lo | (2 << B), base);
                KZG_BF(2, 6, w2)
                const Fe w3 = tw(B + 2, lo | (3 << B), base);
                KZG_BF(3, 7, w3)
            }
            KZG_NORM_ALL
        }
#undef KZG_BF
#undef KZG_BF1
#undef KZG_BF1N
#undef KZG_NORM_ALL
#pragma unroll
        for (int k = 0; k < E; ++k) lds_put(sh, cnt, sbase ^ swz(k << B), e[k]);
    }
    __syncthreads();
}

// `stages` butterfly stages (1..12) of a DIT network on the tile in LDS: stage s pairs tile indices i and i + 2^s.
// tw(s, j, base): twiddle of stage s for pair position j = i mod 2^s (base = any index of the thread's group: the
// caller derives the tile column from its high bits).
template <bool FIRST, int MAXM, class TwFn>
__device__ __forceinline__ void tile_stages(u32* sh, int cnt, int nelem, int stages, TwFn tw) {
    constexpr int NT = TILE >> MAXM;
    if constexpr (MAXM == 3) {
#define KZG_ROUNDS(B_)                                                                  \
    if (stages >= (B_) + 3) round_regs<NT, 3, B_, FIRST>(sh, cnt, nelem, tw);           \
    else if (stages == (B_) + 2) round_regs<NT, 2, B_, FIRST>(sh, cnt, nelem, tw);      \
    else if (stages == (B_) + 1) round_regs<NT, 1, B_, FIRST>(sh, cnt, nelem, tw);
        KZG_ROUNDS(0)
        KZG_ROUNDS(3)
        KZG_ROUNDS(6)
        KZG_ROUNDS(9)
#undef KZG_ROUNDS
    } else {
#define KZG_ROUNDS(B_)                                                                  \
    if (stages >= (B_) + 2) round_regs<NT, 2, B_, FIRST>(sh, cnt, nelem, tw);           \
    else if (stages == (B_) + 1) round_regs<NT, 1, B_, FIRST>(sh, cnt, nelem, tw);
        KZG_ROUNDS(0)
        KZG_ROUNDS(2)
        KZG_ROUNDS(4)
        KZG_ROUNDS(6)
        KZG_ROUNDS(8)
        KZG_ROUNDS(10)
#undef KZG_ROUNDS
    }
}

struct NttParams {
    u32 n;          // transform length
    int logn;
    u32 W;          // roots table width (max_width)
    int inverse;
    Fr scale;       // final multiplier in the 2^261 domain: 2^261 mod r, times n^-1 on the last pass of an inverse
    int dbg;        // timing experiments only (KZGAMD_NTT_DBG): 1 = skip the butterfly stages, 2 = every twiddle = roots[0]
};

// Pass 1 (the whole transform when n <= TILE): stages 0 .. min(logn,12)-1.
//   n <= TILE: a workgroup takes C = cnt / n whole transforms (cnt = min(TILE, n * nbatch) elements): coalesced
//              natural-order loads, scattered into LDS at the bit-reversed index, coalesced natural-order stores.
//   n  > TILE: a workgroup takes TILE consecutive positions of the bit-reversed sequence (block `blk`), i.e. the
//              natural indices  o + (t << L),  o = brev_L(blk), L = logn - 12: 32-byte pieces 2^L elements apart.
//              The four blocks whose pieces share 128-byte lines (o, o^1, o^2, o^3) are given to workgroups
//              8 apart in launch order — same XCD, dispatched together — so the line is fetched from HBM once.
template <int MAXM>
__global__ void __launch_bounds__(TILE >> MAXM) k_ntt_low(Fr* __restrict__ out, const Fr* __restrict__ in,
                                                const Fe* __restrict__ roots, NttParams P, u32 blocks_per_xform, int nelem,
                                                size_t total) {
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    constexpr int NT = TILE >> MAXM;
    const int cnt = (nelem + 31) & ~31;  // limb stride: the swizzle permutes within aligned groups of 32
    const int stages = P.logn < LOG_TILE ? P.logn : LOG_TILE;
    const bool last_pass = P.logn <= LOG_TILE;
    size_t src0, dst0;
    int L = 0;
    if (P.n <= (u32)TILE) {
        src0 = dst0 = (size_t)blockIdx.x * (size_t)nelem;
    } else {
        const u32 xf = blockIdx.x / blocks_per_xform, g = blockIdx.x % blocks_per_xform;
        L = P.logn - LOG_TILE;
        u32 o = g;
        if (L >= 5) {
            const u32 xcd = g & 7, slot = g >> 3;
            o = (((slot >> 2) << 3 | xcd) << 2) | (slot & 3);
        }
        src0 = (size_t)xf * P.n + o;
        dst0 = (size_t)xf * P.n + (size_t)brev(o, L) * TILE;
    }
    const int mask = (1 << stages) - 1;
    for (int t = threadIdx.x; t < nelem; t += NT) {
        const size_t gi = src0 + ((size_t)t << L);
        Fe v;
        if (gi < total) v = fr29::unpack(in[gi]);
        else v = fr29::unpack(Fr::zero());
        lds_put(sh, cnt, swz((t & ~mask) | (int)brev((u32)(t & mask), stages)), v);
    }
    __syncthreads();
    // stage s: half = 2^s, twiddle w_n^(j * n / 2^(s+1)) = roots[j * (W / 2^(s+1))]
    // `roots` is the stage-major table of this direction: entry (2^s - 1) + j = w_{2^(s+1)}^(+-j), so the twiddles a
    // wave asks for are neighbours in memory whatever the ratio between the table width and n
    tile_stages<true, MAXM>(sh, cnt, nelem, (P.dbg & 1) ? 0 : stages, [&](int s, int j, int) -> Fe {
        return roots[(P.dbg & 2) ? 0u : ((1u << s) - 1u) + (u32)j];
    });
    // an inverse transform multiplies by n^-1 at the end of its last pass; everything else only needs the lazy value
    // (< 64r) brought back to [0, r)
    if (last_pass && P.inverse) {
        const Fe fin = fr29::unpack(P.scale);
        for (int t = threadIdx.x; t < nelem; t += NT)
            if (dst0 + t < total) out[dst0 + t] = fr29::finish(lds_get(sh, cnt, swz(t)), fin);
    } else {
        for (int t = threadIdx.x; t < nelem; t += NT)
            if (dst0 + t < total) out[dst0 + t] = fr29::reduce_lazy(lds_get(sh, cnt, swz(t)));
    }
}

// Passes 2, 3: `logR` stages starting at global stage `stage0` (a multiple of 12), in place.
// View the bit-reversed-order array as [hi][r][lo] with lo < 2^stage0, r < R = 2^logR: stage stage0+s pairs
// rows r and r + 2^s.  A workgroup takes C = 4096 / R consecutive lo positions of one hi block
// (C * 32 B contiguous per row), runs the logR stages on the tile, writes back.  Tile index = c * R + r.
template <int MAXM>
__global__ void __launch_bounds__(TILE >> MAXM) k_ntt_high(Fr* __restrict__ data, const Fe* __restrict__ roots, NttParams P,
                                                 u32 tiles_per_xform, int stage0, int logR, int last) {
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    constexpr int NT = TILE >> MAXM;
    const u32 xf = blockIdx.x / tiles_per_xform, tile = blockIdx.x % tiles_per_xform;
    const int C = TILE >> logR, logC = LOG_TILE - logR;
    Fr* base = data + (size_t)xf * P.n;
    const u32 lo_tiles = (1u << stage0) >> logC;  // tiles per hi block
    const u32 hi = tile / lo_tiles, lo0 = (tile % lo_tiles) << logC;
    const size_t origin = ((size_t)hi << (stage0 + logR)) + lo0;
    for (int e = threadIdx.x; e < TILE; e += NT) {
        const int c = e & (C - 1), r = e >> logC;  // consecutive lanes -> consecutive positions
        lds_put(sh, TILE, swz((c << logR) | r), fr29::unpack(base[origin + ((size_t)r << stage0) + c]));
    }
    __syncthreads();
    tile_stages<false, MAXM>(sh, TILE, TILE, (P.dbg & 1) ? 0 : logR, [&](int s, int j, int tidx) -> Fe {
        // global stage stage0+s: half = 2^(stage0+s); position mod half = (r mod 2^s) * 2^stage0 + lo
        const u32 lo = lo0 + (u32)(tidx >> logR);
        const u32 jg = ((u32)j << stage0) + lo;
        return roots[(P.dbg & 2) ? 0u : ((1u << (stage0 + s)) - 1u) + jg];
    });
    if (last && P.inverse) {
        const Fe fin = fr29::unpack(P.scale);
        for (int e = threadIdx.x; e < TILE; e += NT) {
            const int c = e & (C - 1), r = e >> logC;
            base[origin + ((size_t)r << stage0) + c] = fr29::finish(lds_get(sh, TILE, swz((c << logR) | r)), fin);
        }
    } else {
        for (int e = threadIdx.x; e < TILE; e += NT) {
            const int c = e & (C - 1), r = e >> logC;
            base[origin + ((size_t)r << stage0) + c] = fr29::reduce_lazy(lds_get(sh, TILE, swz((c << logR) | r)));
        }
    }
}

// DAS helper: data[i] *= roots[i * stride]  (the shift by the 2n-th root between the two NTTs)
__global__ void __launch_bounds__(256) k_twist(Fr* __restrict__ data, const Fe* __restrict__ roots, u32 n, u32 stride,
                                               size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    u32 i = (u32)(t % n);
    data[t] = fr29::finish(fr29::unpack(data[t]), roots[(size_t)i * stride]);
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

// enqueue nbatch transforms of length n (device pointers; out may equal in only when n > TILE is false... so never alias)
void ntt_enqueue(NttCtx* ctx, Fr* d_out, const Fr* d_in, size_t n, size_t nbatch, bool inverse, hipStream_t stream) {
    NttParams P;
    P.n = (u32)n;
    P.logn = ilog2(n);
    P.W = (u32)ctx->W;
    P.inverse = inverse ? 1 : 0;
    static const int dbg = getenv("KZGAMD_NTT_DBG") ? atoi(getenv("KZGAMD_NTT_DBG")) : 0;
    P.dbg = dbg;
    // 2^261 mod r as a plain residue (= "one" of the 2^261 domain); an inverse transform folds n^-1 in:
    // data is d*2^256, so the multiplier n^-1*2^261 is (n^-1 in blst Montgomery form) * 2^5
    P.scale = inverse ? times32(inv_len(n)) : one261();
    const size_t total = n * nbatch;
    const Fe* tw = (const Fe*)(inverse ? ctx->d_tw_inv : ctx->d_tw_fwd);
    if (n <= (size_t)TILE) {
        // whole transforms per workgroup: TILE / n of them (all of them when the batch is smaller than a tile)
        const size_t cnt = total < (size_t)TILE ? total : (size_t)TILE;
        const size_t wgs = (total + cnt - 1) / cnt;
        const size_t lds = ((cnt + 31) & ~(size_t)31) * sizeof(u32) * fr29::L;
        hipLaunchKernelGGL(k_ntt_low<3>, dim3((unsigned)wgs), dim3(TILE >> 3), lds, stream, d_out, d_in,
                               tw, P, 1u, (int)cnt, total);
    } else {
        const u32 blocks = (u32)(n >> LOG_TILE);
        hipLaunchKernelGGL(k_ntt_low<3>, dim3((unsigned)(blocks * nbatch)), dim3(TILE >> 3), (size_t)TILE * sizeof(u32) * fr29::L,
                               stream, d_out, d_in, tw, P, blocks, TILE, total);
    }
    // remaining stages, up to 12 per pass: 12..23, then 24..30
    for (int stage0 = LOG_TILE; stage0 < P.logn; stage0 += LOG_TILE) {
        const int logR = P.logn - stage0 < LOG_TILE ? P.logn - stage0 : LOG_TILE;
        const int last = stage0 + logR == P.logn;
        const u32 tiles = (u32)(n >> LOG_TILE);
        hipLaunchKernelGGL(k_ntt_high<3>, dim3((unsigned)(tiles * nbatch)), dim3(TILE >> 3), (size_t)TILE * sizeof(u32) * fr29::L,
                               stream, d_out, tw, P, tiles, stage0, logR, last);
    }
    NTT_TRY(hipGetLastError());
}

}  // namespace

extern "C" void* kzgamd_ntt_new(unsigned scale) {
    if (scale >= 32) return nullptr;  // "Scale is expected to be within root of unity matrix row size"
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "kzg_mi355x: kzgamd_ntt_new: no gfx950 device visible\n");
        return nullptr;
    }
    auto* ctx = new NttCtx();
    try {
        NTT_TRY(hipGetDevice(&ctx->device));
        NTT_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->scale = scale;
        ctx->W = (size_t)1 << scale;
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

extern "C" int ntt_fr(void* vctx, blst_fr* out, const blst_fr* in, size_t n, int inverse) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx || !out || !in) return -1;
    if (n > ctx->W) return 1;                 // "Supplied list is longer than the available max width"
    if (n == 0 || (n & (n - 1))) return 2;    // "A list with power-of-two length expected"
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
// fused butterfly network (data_availability_sampling.rs:14-72), including the final n^-1 (:95-97,
// absorbed here by the inverse transform).
extern "C" int das_fft_extension(void* vctx, blst_fr* odds, const blst_fr* evens, size_t n) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx || !odds || !evens) return -1;
    if (n == 0) return 1;                  // "A non-zero list ab expected"
    if (n & (n - 1)) return 2;             // "A list with power-of-two length expected"
    if (n * 2 > ctx->W) return 3;          // "Supplied list is longer than the available max width"
    std::lock_guard<std::mutex> lk(ctx->mu);
    try {
        kzgamd::DeviceGuard on_device(ctx->device);
        NTT_TRY(on_device.err);
        ctx->ensure(n);
        NTT_TRY(hipMemcpyAsync(ctx->d_a, evens, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        ntt_enqueue(ctx, ctx->d_b, ctx->d_a, n, 1, true, ctx->stream);
        hipLaunchKernelGGL(k_twist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_b,
                           (const Fe*)ctx->d_roots, (u32)n, (u32)(ctx->W / (2 * n)), n);
        ntt_enqueue(ctx, ctx->d_a, ctx->d_b, n, 1, false, ctx->stream);
        NTT_TRY(hipMemcpyAsync(odds, ctx->d_a, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        NTT_TRY(hipStreamSynchronize(ctx->stream));
    } catch (const NttErr& e) {
        return -(int)e.e - 100;
    }
    return 0;
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
