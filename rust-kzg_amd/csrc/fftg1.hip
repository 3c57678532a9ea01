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

#include "../../include/kzg_mi355x.h"
#include "ff.hip.h"
#include "g1_28.hip.h"
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
__global__ void __launch_bounds__(NTHREADS) k_g1_stage(Xyzz* __restrict__ dst, const Xyzz* __restrict__ data, Xyzz* __restrict__ tab,
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
__global__ void __launch_bounds__(NTHREADS) k_g1_store(ff::Fp* __restrict__ out, const Xyzz* __restrict__ data,
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
__global__ void __launch_bounds__(NTHREADS) k_g1_scale_xyzz(Xyzz* __restrict__ data, Xyzz* __restrict__ tab, RootSplit inv_n,
                                                            size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // total = 2 * points
    if (t >= total) return;
    const int half = (int)(t & 1);
    Xyzz p = data[t >> 1];
    glv_mul_pair(p, inv_n, half, tab + t, total);
    if (half == 0) data[t >> 1] = p;
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
        for (int s = 0; s < logn; ++s)
            hipLaunchKernelGGL(k_g1_stage, dim3((unsigned)((2 * bf + NTHREADS - 1) / NTHREADS)), dim3(NTHREADS), 0, st,
                               bufs[(s + 1) & 1], (const Xyzz*)bufs[s & 1], tab, (const RootSplit*)ctx->d_kroots, (u32)n, s,
                               (u32)ctx->W, inverse ? 1 : 0, 2 * bf);
        Xyzz* res = bufs[logn & 1];
        if (inverse && n > 1 && scale_inverse) {
            Fr v = Fr::zero();
            v.v[0] = (u32)n;
            v.v[1] = (u32)((u64)n >> 32);
            const RootSplit inv_n = split_scalar(ff::from_mont(ff::inverse_bgcd(ff::to_mont(v))));
            hipLaunchKernelGGL(k_g1_scale_xyzz, dim3((unsigned)((2 * total + NTHREADS - 1) / NTHREADS)), dim3(NTHREADS), 0, st, res,
                               tab, inv_n, 2 * total);
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
        for (int s = 0; s < logn; ++s)
            hipLaunchKernelGGL(k_g1_stage, dim3((unsigned)((2 * bf + NTHREADS - 1) / NTHREADS)), dim3(NTHREADS), 0, st,
                               pts + ((s + 1) & 1) * total, (const Xyzz*)(pts + (s & 1) * total), tab,
                               (const RootSplit*)ctx->d_kroots, (u32)n, s, (u32)ctx->W, inverse ? 1 : 0, 2 * bf);
        RootSplit inv_n;
        memset(&inv_n, 0, sizeof inv_n);
        if (inverse) {
            Fr v = Fr::zero();
            v.v[0] = (u32)n;
            v.v[1] = (u32)((u64)n >> 32);
            inv_n = split_scalar(ff::from_mont(ff::inverse_bgcd(ff::to_mont(v))));
        }
        hipLaunchKernelGGL(k_g1_store, dim3((unsigned)((2 * total + NTHREADS - 1) / NTHREADS)), dim3(NTHREADS), 0, st,
                           (ff::Fp*)ctx->d_p1, (const Xyzz*)(pts + (logn & 1) * total), tab, inv_n, inverse && n > 1 ? 1 : 0, 2 * total);
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
