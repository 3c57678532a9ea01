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
#include "ntt_internal.h"

using ff::Fr;
using fr29::Fe;
using ff::u32;
using ff::u64;

namespace {

constexpr int LOG_TILE = 12;
constexpr int TILE = 1 << LOG_TILE;  // elements per workgroup
constexpr int NT = 1024;             // threads per workgroup

__device__ __forceinline__ u32 brev(u32 v, int bits) { return __builtin_bitreverse32(v) >> (32 - bits); }

// LDS holds elements in the 9 x 29-bit form, limb-major: sh[limb * cnt + idx] -> unit-stride lanes hit
// distinct banks (4096 x 36 B = 144 KiB of the 160 KiB LDS)
__device__ __forceinline__ Fe lds_get(const u32* sh, int cnt, int idx) {
    Fe r;
#pragma unroll
    for (int k = 0; k < fr29::L; ++k) r.v[k] = sh[k * cnt + idx];
    return r;
}
__device__ __forceinline__ void lds_put(u32* sh, int cnt, int idx, const Fe& a) {
#pragma unroll
    for (int k = 0; k < fr29::L; ++k) sh[k * cnt + idx] = a.v[k];
}

// `stages` butterfly stages of a DIT network on the `cnt` elements in LDS.
// Element e of the tile has global index  g = e_hi * gstride + goff  pattern handled by the caller
// through twiddle_index(); here: stage s pairs e and e + 2^s (s = 0 .. stages-1).
template <class TwFn>
__device__ __forceinline__ void lds_stages(u32* sh, int cnt, int stages, TwFn tw) {
    for (int s = 0; s < stages; ++s) {
        const int half = 1 << s;
        for (int b = threadIdx.x; b < cnt / 2; b += NT) {
            const int j = b & (half - 1);
            const int i0 = ((b >> s) << (s + 1)) | j, i1 = i0 + half;
            Fe x = lds_get(sh, cnt, i0), y = lds_get(sh, cnt, i1);
            Fe t = fr29::mul(y, tw(s, j, i0));
            fr29::butterfly(x, y, t);
            lds_put(sh, cnt, i0, x);
            lds_put(sh, cnt, i1, y);
        }
        __syncthreads();
    }
}

struct NttParams {
    u32 n;          // transform length
    int logn;
    u32 W;          // roots table width (max_width)
    int inverse;
    Fr scale;       // final multiplier in the 2^261 domain: 2^261 mod r, times n^-1 on the last pass of an inverse
};

// Pass 1 (and the whole transform when n <= TILE): block `blk` of min(n,TILE) consecutive
// positions of the bit-reversed sequence; stages 0 .. min(logn,12)-1.
__global__ void __launch_bounds__(NT) k_ntt_low(Fr* __restrict__ out, const Fr* __restrict__ in,
                                                const Fe* __restrict__ roots, NttParams P, u32 blocks_per_xform) {
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 xf = blockIdx.x / blocks_per_xform, blk = blockIdx.x % blocks_per_xform;
    const int cnt = P.n < (u32)TILE ? (int)P.n : TILE;
    const int stages = P.logn < LOG_TILE ? P.logn : LOG_TILE;
    const Fr* src = in + (size_t)xf * P.n;
    Fr* dst = out + (size_t)xf * P.n;
    for (int e = threadIdx.x; e < cnt; e += NT) {
        const u32 pos = blk * (u32)cnt + (u32)e;  // position in the bit-reversed sequence
        lds_put(sh, cnt, e, fr29::unpack(src[brev(pos, P.logn)]));
    }
    __syncthreads();
    // stage s: half = 2^s, twiddle w_n^(j * n / 2^(s+1)) = roots[j * (W / 2^(s+1))]
    const bool last_pass = P.logn <= LOG_TILE;
    lds_stages(sh, cnt, stages, [&](int s, int j, int) -> Fe {
        const u32 idx = (u32)j * (P.W >> (s + 1));
        return roots[P.inverse ? P.W - idx : idx];
    });
    const Fe fin = last_pass ? fr29::unpack(P.scale) : fr29::one();
    for (int e = threadIdx.x; e < cnt; e += NT) dst[blk * (u32)cnt + (u32)e] = fr29::finish(lds_get(sh, cnt, e), fin);
}

// Passes 2, 3: `logR` stages starting at global stage `stage0` (a multiple of 12), in place.
// View the bit-reversed-order array as [hi][r][lo] with lo < 2^stage0, r < R = 2^logR: stage stage0+s pairs
// rows r and r + 2^s.  A workgroup takes C = 4096 / R consecutive lo positions of one hi block
// (C * 32 B contiguous per row: coalesced while R <= 256), runs the logR stages in LDS, writes back.
__global__ void __launch_bounds__(NT) k_ntt_high(Fr* __restrict__ data, const Fe* __restrict__ roots, NttParams P,
                                                 u32 tiles_per_xform, int stage0, int logR, int last) {
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 xf = blockIdx.x / tiles_per_xform, tile = blockIdx.x % tiles_per_xform;
    const int R = 1 << logR, C = TILE >> logR, logC = LOG_TILE - logR;
    Fr* base = data + (size_t)xf * P.n;
    const u32 lo_tiles = (1u << stage0) >> logC;  // tiles per hi block
    const u32 hi = tile / lo_tiles, lo0 = (tile % lo_tiles) << logC;
    const size_t origin = ((size_t)hi << (stage0 + logR)) + lo0;
    // LDS element e = c * R + r  (row index fastest, so a butterfly pairs e and e + 2^s)
    for (int e = threadIdx.x; e < TILE; e += NT) {
        const int c = e & (C - 1), r = e >> logC;  // consecutive lanes -> consecutive positions
        lds_put(sh, TILE, c * R + r, fr29::unpack(base[origin + ((size_t)r << stage0) + c]));
    }
    __syncthreads();
    lds_stages(sh, TILE, logR, [&](int s, int j, int i0) -> Fe {
        // global stage stage0+s: half = 2^(stage0+s); position mod half = (r mod 2^s) * 2^stage0 + lo
        const u32 lo = lo0 + (u32)(i0 >> logR);
        const u32 jg = ((u32)j << stage0) + lo;
        const u32 idx = jg * (P.W >> (stage0 + s + 1));
        return roots[P.inverse ? P.W - idx : idx];
    });
    const Fe fin = last ? fr29::unpack(P.scale) : fr29::one();
    for (int e = threadIdx.x; e < TILE; e += NT) {
        const int c = e & (C - 1), r = e >> logC;
        base[origin + ((size_t)r << stage0) + c] = fr29::finish(lds_get(sh, TILE, c * R + r), fin);
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
    // 2^261 mod r as a plain residue (= "one" of the 2^261 domain); an inverse transform folds n^-1 in:
    // data is d*2^256, so the multiplier n^-1*2^261 is (n^-1 in blst Montgomery form) * 2^5
    P.scale = inverse ? times32(inv_len(n)) : one261();
    const size_t lds = (n < (size_t)TILE ? n : (size_t)TILE) * sizeof(u32) * fr29::L;
    const u32 blocks = n <= (size_t)TILE ? 1u : (u32)(n >> LOG_TILE);
    hipLaunchKernelGGL(k_ntt_low, dim3((unsigned)(blocks * nbatch)), dim3(NT), lds, stream, d_out, d_in,
                       (const Fe*)ctx->d_roots, P, blocks);
    // remaining stages, up to 12 per pass: 12..23, then 24..30
    for (int stage0 = LOG_TILE; stage0 < P.logn; stage0 += LOG_TILE) {
        const int logR = P.logn - stage0 < LOG_TILE ? P.logn - stage0 : LOG_TILE;
        const int last = stage0 + logR == P.logn;
        const u32 tiles = (u32)(n >> LOG_TILE);
        hipLaunchKernelGGL(k_ntt_high, dim3((unsigned)(tiles * nbatch)), dim3(NT), (size_t)TILE * sizeof(u32) * fr29::L, stream,
                           d_out, (const Fe*)ctx->d_roots, P, tiles, stage0, logR, last);
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
    } catch (...) {
        delete ctx;
        return nullptr;
    }
    return ctx;
}

extern "C" void kzgamd_ntt_free(void* ctx) { delete (NttCtx*)ctx; }

extern "C" int kzgamd_ntt_fr_device(void* vctx, void* d_out, const void* d_in, size_t n, size_t nbatch, int inverse,
                                    void* stream) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx || !d_out || !d_in) return -1;
    if (n > ctx->W) return 1;
    if (n == 0 || (n & (n - 1))) return 2;
    if (d_out == d_in) return -3;
    try {
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
        NTT_TRY(hipSetDevice(ctx->device));
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
        NTT_TRY(hipSetDevice(ctx->device));
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
