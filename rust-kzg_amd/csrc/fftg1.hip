// G1-valued radix-2 transform (B2, widening row): replaces FFTG1::fft_g1 for FsFFTSettings
// (blst/src/fft_g1.rs:13-83).  Same contract: natural order in and out, roots taken with stride
// max_width/n from roots_of_unity / reverse_roots_of_unity, inverse scaled by n^-1.
//
// The reference recurses (even/odd split) and spends its time in FsG1::mul, a 255-bit scalar
// multiplication per butterfly: n/2 * log2(n) of them.  Here the same butterfly network runs iteratively,
// one kernel per stage, one lane per butterfly (all transforms of a batch in one launch):
//     t = w^j * y  : fixed 4-bit windows, the 15 multiples of y in a per-lane table in HBM
//                    (64 windows: 252 doublings + <= 64 additions + 14 for the table, XYZZ coordinates)
//     x' = x + t,  y' = x - t
// Points live in HBM as XYZZ over the 14 x 28-bit field (g1_28.cuh) between stages; blst Jacobian at the
// boundary.  The work is integer-VALU bound (~3400 field multiplications per butterfly); HBM traffic is
// negligible next to it (448 B in/out plus a 3.4 KB table per butterfly).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/kzg_mi355x.h"
#include "ff.cuh"
#include "g1_28.cuh"
#include "ntt_internal.h"

using ff::Fr;
using ff::u32;
using ff::u64;
using g1::Xyzz;

namespace {

constexpr int WIN = 4;              // scalar window
constexpr int NTAB = (1 << WIN) - 1;  // multiples 1..15
constexpr int NTHREADS = 64;        // one wave per workgroup: long serial lanes, spread over all CUs

struct Scalar {
    u32 w[8];
};

__device__ __forceinline__ u32 brev(u32 v, int bits) { return bits == 0 ? 0u : __builtin_bitreverse32(v) >> (32 - bits); }

// The point routines are kept out of line here: a lane executes ~330 of them per butterfly, and inlining each
// (20-35 KB of code apiece) buys nothing against their ~6000-instruction bodies.
__device__ __noinline__ void pt_dbl(Xyzz& a) {
    if (!g1::is_inf(a)) g1::dbl(a);
}
__device__ __noinline__ void pt_add(Xyzz& a, const Xyzz& b) { g1::dadd(a, b); }

// acc = k * acc, k a canonical 256-bit scalar (little-endian words); tab = this lane's 15 table slots,
// slot e at tab[e * tab_stride]
__device__ void scalar_mul(Xyzz& acc, const Scalar& k, Xyzz* tab, size_t tab_stride) {
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
    for (int w = 256 / WIN - 1; w >= 0; --w) {
        if (!g1::is_inf(acc)) {
            pt_dbl(acc);
            pt_dbl(acc);
            pt_dbl(acc);
            pt_dbl(acc);
        }
        const u32 d = (k.w[w >> 3] >> ((w & 7) * WIN)) & (u32)NTAB;
        if (d) {
            Xyzz q = tab[(size_t)(d - 1) * tab_stride];
            pt_add(acc, q);
        }
    }
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
// twiddle w_n^(j * n / 2^(s+1)) = roots[j * (W >> (s+1))]  (fft_g1.rs:22-29 unrolled)
__global__ void __launch_bounds__(NTHREADS) k_g1_stage(Xyzz* __restrict__ data, Xyzz* __restrict__ tab,
                                                       const Scalar* __restrict__ kroots, u32 n, int s, u32 W, int inverse,
                                                       size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const u32 halfn = n >> 1;
    const size_t xf = t / halfn;
    const u32 b = (u32)(t % halfn);
    const u32 half = 1u << s, j = b & (half - 1);
    const u32 i0 = ((b >> s) << (s + 1)) | j, i1 = i0 + half;
    Xyzz* base = data + xf * n;
    Xyzz y = base[i1];
    const u32 idx = j * (W >> (s + 1));
    if (idx != 0) scalar_mul(y, kroots[inverse ? W - idx : idx], tab + t, total);
    Xyzz x = base[i0];
    Xyzz d = x;
    pt_add(x, y);
    y.y = fp28::neg<8>(y.y);  // -t : Y < 8p is within what dadd/dbl accept
    pt_add(d, y);
    base[i0] = x;
    base[i1] = d;
}

// XYZZ -> blst Jacobian; an inverse transform multiplies by n^-1 first (fft_g1.rs:72-79)
__global__ void __launch_bounds__(NTHREADS) k_g1_store(ff::Fp* __restrict__ out, const Xyzz* __restrict__ data,
                                                       Xyzz* __restrict__ tab, Scalar inv_n, int scale, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    Xyzz p = data[t];
    if (scale) scalar_mul(p, inv_n, tab + t, total);
    g1::to_blst_jacobian(out + t * 3, p);
}

int ilog2(size_t n) {
    int l = 0;
    while (((size_t)1 << l) < n) ++l;
    return l;
}

void ensure_g1(NttCtx* ctx, size_t total, size_t tab_lanes) {
    if (!ctx->d_kroots) {
        std::vector<Fr> plain(ctx->W + 1);
        for (size_t i = 0; i <= ctx->W; ++i) plain[i] = ff::from_mont(ctx->roots[i]);
        NTT_TRY(hipMalloc(&ctx->d_kroots, (ctx->W + 1) * sizeof(Fr)));
        NTT_TRY(hipMemcpy(ctx->d_kroots, plain.data(), (ctx->W + 1) * sizeof(Fr), hipMemcpyHostToDevice));
    }
    if (total > ctx->cap_g1) {
        if (ctx->d_p1) (void)hipFree(ctx->d_p1);
        if (ctx->d_pts) (void)hipFree(ctx->d_pts);
        ctx->d_p1 = ctx->d_pts = nullptr;
        ctx->cap_g1 = 0;
        NTT_TRY(hipMalloc(&ctx->d_p1, total * sizeof(blst_p1)));
        NTT_TRY(hipMalloc(&ctx->d_pts, total * sizeof(Xyzz)));
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

}  // namespace

extern "C" int kzgamd_fft_g1_batch(void* vctx, blst_p1* out, const blst_p1* in, size_t n, size_t nbatch, int inverse) {
    NttCtx* ctx = (NttCtx*)vctx;
    if (!ctx || !out || !in) return -1;
    if (n > ctx->W) return 1;               // "Supplied list is longer than the available max width"
    if (n == 0 || (n & (n - 1))) return 2;  // "A list with power-of-two length expected"
    if (nbatch == 0) return 0;
    std::lock_guard<std::mutex> lk(ctx->mu);
    try {
        NTT_TRY(hipSetDevice(ctx->device));
        const size_t total = n * nbatch, bf = total / 2;
        const int logn = ilog2(n);
        ensure_g1(ctx, total, inverse ? total : (bf ? bf : 1));
        Xyzz* pts = (Xyzz*)ctx->d_pts;
        Xyzz* tab = (Xyzz*)ctx->d_tab;
        hipStream_t st = ctx->stream;
        NTT_TRY(hipMemcpyAsync(ctx->d_p1, in, total * sizeof(blst_p1), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_g1_load, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pts, (const ff::Fp*)ctx->d_p1,
                           (u32)n, logn, total);
        for (int s = 0; s < logn; ++s)
            hipLaunchKernelGGL(k_g1_stage, dim3((unsigned)((bf + NTHREADS - 1) / NTHREADS)), dim3(NTHREADS), 0, st, pts, tab,
                               (const Scalar*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0, bf);
        Scalar inv_n;
        memset(&inv_n, 0, sizeof inv_n);
        if (inverse) {
            Fr v = Fr::zero();
            v.v[0] = (u32)n;
            v.v[1] = (u32)((u64)n >> 32);
            const Fr iv = ff::from_mont(ff::inverse_bgcd(ff::to_mont(v)));
            for (int i = 0; i < 8; ++i) inv_n.w[i] = iv.v[i];
        }
        hipLaunchKernelGGL(k_g1_store, dim3((unsigned)((total + NTHREADS - 1) / NTHREADS)), dim3(NTHREADS), 0, st,
                           (ff::Fp*)ctx->d_p1, (const Xyzz*)pts, tab, inv_n, inverse && n > 1 ? 1 : 0, total);
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
