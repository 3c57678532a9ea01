// c-kzg-4844 surface (B3) for the proving path, MI355X-native.
// Replaces blst/src/eip_4844.rs:160-530 + the generic protocol code it calls
// (kzg/src/eip_4844.rs).  The CKZGSettings struct has the reference's layout
// (kzg/src/eth/c_bindings.rs:55-108); the device-resident state (fixed-base MSM table,
// roots) hangs off a registry keyed by the settings' g1_values_lagrange_brp pointer —
// the same trick the reference uses for its precomputation tables
// (PrecomputationTableManager, kzg/src/eip_4844.rs:64-146).
#include "ckzg_shared.h"

namespace {

// ---------------------------------------------------------------- kernels

__global__ void __launch_bounds__(128) k_uncompress(AffPt* __restrict__ out, int* __restrict__ bad,
                                                    const unsigned char* __restrict__ in, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned char buf[48];
    for (int k = 0; k < 48; ++k) buf[k] = in[48 * i + k];
    AffPt p;
    if (!g1io::uncompress(p, buf)) atomicAdd(bad, 1);
    out[i] = p;
}

// AffPt -> blst_p1 (Jacobian, Z = one; infinity = all-zero) for the CKZGSettings arrays
__global__ void __launch_bounds__(256) k_affpt_to_blst_p1(ff::Fp* __restrict__ out, const AffPt* __restrict__ in, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    AffPt p = in[i];
    if (p.flags & 1) {
        out[3 * i] = ff::Fp::zero();
        out[3 * i + 1] = ff::Fp::zero();
        out[3 * i + 2] = ff::Fp::zero();
    } else {
        out[3 * i] = fp28::to_blst(p.x);
        out[3 * i + 1] = fp28::to_blst(p.y);
        out[3 * i + 2] = ff::Fp::one();
    }
}

// blob bytes (4096 x 32 B big-endian) -> canonical little-endian scalars; status[b] = 1 if any element >= r
// (bytes_to_blob / FsFr::from_bytes, kzg/src/eip_4844.rs:867-880, blst/src/types/fr.rs:64-86)
__global__ void __launch_bounds__(256) k_blob_to_scalars(u32* __restrict__ out, int* __restrict__ status,
                                                         const u32* __restrict__ blobs, size_t nblobs) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * N) return;
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = __builtin_bswap32(blobs[t * 8 + (7 - k)]);
    // w >= r ?
    u64 borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        u64 d = (u64)w[k] - ff::FrParams::p(k) - borrow;
        borrow = (d >> 32) & 1;
    }
    if (!borrow) status[t / N] = 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) out[t * 8 + k] = w[k];
}




// The blobs of a lane batch stay where their callers staged them (page-locked host memory, one slot per caller): the
// kernels that read raw blob bytes take one pointer per blob and fetch them over PCIe themselves — no host-side
// gathering into one buffer, no copy operation on the stream.
struct BlobPtrs {
    const u32* p[16];  // KzgAmdSettings::LANE_MAX_BLOBS
};
__global__ void __launch_bounds__(256) k_blob_to_scalars_ptrs(u32* __restrict__ out, int* __restrict__ status, BlobPtrs blobs,
                                                              size_t nblobs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * N) return;
    const uint4* src = reinterpret_cast<const uint4*>(blobs.p[t / N] + (t % N) * 8);
    const uint4 lo = src[0], hi = src[1];
    const u32 raw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = __builtin_bswap32(raw[7 - k]);
    u64 borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        u64 d = (u64)w[k] - ff::FrParams::p(k) - borrow;
        borrow = (d >> 32) & 1;
    }
    if (!borrow) status[t / N] = 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) out[t * 8 + k] = w[k];
}
// the same blobs gathered into one device buffer (the proving kernels read a blob more than once)
__global__ void __launch_bounds__(256) k_gather_blobs(uint4* __restrict__ out, BlobPtrs blobs, size_t nblobs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr size_t PER = BYTES_PER_BLOB / 16;
    if (t >= nblobs * PER) return;
    out[t] = reinterpret_cast<const uint4*>(blobs.p[t / PER])[t % PER];
}


// compute_kzg_proof_rust up to the MSM (kzg/src/eip_4844.rs:437-510): y = p(z) by the barycentric
// formula (evaluate_polynomial_in_evaluation_form :954-1003) and the quotient polynomial in
// evaluation form, including the z-inside-the-domain column (:484-510).  One workgroup per blob;
// the 4096 inversions are one Fermat inversion per blob via a block-wide product scan
// (the reference's fr_batch_inv :882-914 is the same trick, serial).
// Outputs: q as canonical little-endian scalars (ready for the MSM), y canonical.
__global__ void __launch_bounds__(QT) k_quotient(u32* __restrict__ q_out, u32* __restrict__ y_out, int* __restrict__ status,
                                                 const u32* __restrict__ blobs, const u32* __restrict__ z_be,
                                                 const ff::Fr* __restrict__ roots_brp, ff::Fr ninv) {
    __shared__ ff::Fr sh_a[QT];
    __shared__ ff::Fr sh_b[QT];
    __shared__ ff::Fr sh_misc[4];  // 0: total^-1, 1: y, 2: z^-1 (domain case)
    __shared__ int sh_m, sh_bad;
    const int t = threadIdx.x;
    const size_t blob = blockIdx.x;
    const u32* bw = blobs + blob * (N * 8);
    if (t == 0) {
        sh_m = -1;
        sh_bad = 0;
    }
    __syncthreads();
    bool zok;
    ff::Fr z = ff::to_mont(fr_load_be(z_be + blob * 8, &zok));
    if (!zok && t == 0) sh_bad = 1;

    // The thread's QE prefix products, then its QE inverses, are kept in the element's own 32-byte slot of q_out until the
    // quotient overwrites them (same lane, same address).  (Per-thread arrays instead were 536 bytes of scratch per lane —
    // the loops are too large to unroll with the multiplications inlined, so the arrays were indexed dynamically; removed
    // in round 5.)
    uint4* qslot = reinterpret_cast<uint4*>(q_out + blob * (N * 8));
    auto put = [&](int, int i, const ff::Fr& x) {
        qslot[2 * i] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        qslot[2 * i + 1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    };
    auto get = [&](int, int i) {
        const uint4 lo = qslot[2 * i], hi = qslot[2 * i + 1];
        ff::Fr x;
        x.v[0] = lo.x, x.v[1] = lo.y, x.v[2] = lo.z, x.v[3] = lo.w;
        x.v[4] = hi.x, x.v[5] = hi.y, x.v[6] = hi.z, x.v[7] = hi.w;
        return x;
    };
    // d_k = z - w_i, prefix products within the thread
    ff::Fr prod = ff::Fr::one();
#pragma unroll
    for (int k = 0; k < QE; ++k) {
        const int i = k * QT + t;
        ff::Fr d = ff::sub(z, roots_brp[i]);
        if (d.is_zero()) {
            sh_m = i;
            d = ff::Fr::one();
        }
        put(k, i, prod);
        prod = fmul(prod, d);
    }
    // block-wide inclusive prefix (sh_a) and suffix (sh_b) products of the per-thread products
    sh_a[t] = prod;
    sh_b[t] = prod;
    __syncthreads();
    for (int off = 1; off < QT; off <<= 1) {
        ff::Fr pa = sh_a[t], pb = sh_b[t];
        if (t >= off) pa = fmul(sh_a[t - off], pa);
        if (t + off < QT) pb = fmul(pb, sh_b[t + off]);
        __syncthreads();
        sh_a[t] = pa;
        sh_b[t] = pb;
        __syncthreads();
    }
    if (t == 0) sh_misc[0] = fr_inverse(sh_a[QT - 1]);
    __syncthreads();
    ff::Fr inv = sh_misc[0];
    if (t > 0) inv = fmul(inv, sh_a[t - 1]);
    if (t + 1 < QT) inv = fmul(inv, sh_b[t + 1]);  // inv = 1 / P_t
    const int m = sh_m;
    __syncthreads();

    // back-substitution: inv_k = 1/d_k; barycentric sum  sum p_i w_i / (z - w_i)
    ff::Fr acc = ff::Fr::zero();
    bool bad = false;
#pragma unroll
    for (int k = QE - 1; k >= 0; --k) {
        const int i = k * QT + t;
        const ff::Fr w = roots_brp[i];
        ff::Fr d = ff::sub(z, w);
        if (i == m) d = ff::Fr::one();
        const ff::Fr inv_k = fmul(inv, get(k, i));
        put(k, i, inv_k);
        inv = fmul(inv, d);
        bool ok;
        // the blob element stays canonical: a Montgomery product with one canonical operand is the canonical product,
        // so neither the elements nor the results below need a conversion multiplication
        const ff::Fr p = fr_load_be(bw + (size_t)i * 8, &ok);
        bad |= !ok;
        acc = ff::add(acc, fmul(fmul(inv_k, w), p));
    }
    if (bad) sh_bad = 1;
    sh_a[t] = acc;
    __syncthreads();
    for (int off = QT / 2; off > 0; off >>= 1) {
        if (t < off) sh_a[t] = ff::add(sh_a[t], sh_a[t + off]);
        __syncthreads();
    }
    if (t == 0) {
        ff::Fr y;  // canonical
        if (m >= 0) {
            bool ok;
            y = fr_load_be(bw + (size_t)m * 8, &ok);
            sh_misc[2] = fr_inverse(z);
        } else {
            // out = sum / N * (z^N - 1)
            ff::Fr zn = z;
            for (int k = 0; k < 12; ++k) zn = fmul(zn, zn);
            y = fmul(fmul(sh_a[0], ninv), ff::sub(zn, ff::Fr::one()));  // ninv = 1/N, computed once on the host
        }
        sh_misc[1] = y;
#pragma unroll
        for (int k = 0; k < 8; ++k) y_out[blob * 8 + k] = y.v[k];
        if (sh_bad) status[blob] = 1;
    }
    __syncthreads();
    const ff::Fr y = sh_misc[1];
    // q_i = (p_i - y) / (w_i - z) = (y - p_i) * inv_i ;  domain case: column m gets
    // sum_{i != m} (p_i - y) * w_i / (z * (z - w_i))
    ff::Fr col = ff::Fr::zero();
#pragma unroll
    for (int k = 0; k < QE; ++k) {
        const int i = k * QT + t;
        if (i == m) continue;
        bool ok;
        const ff::Fr p = fr_load_be(bw + (size_t)i * 8, &ok);
        const ff::Fr ymp = ff::sub(y, p);            // canonical
        const ff::Fr inv_k = get(k, i);
        const ff::Fr qc = fmul(ymp, inv_k);       // canonical x Montgomery -> canonical
        if (m >= 0) col = ff::add(col, fmul(fmul(ff::neg(ymp), roots_brp[i]), inv_k));
        qslot[2 * i] = make_uint4(qc.v[0], qc.v[1], qc.v[2], qc.v[3]);
        qslot[2 * i + 1] = make_uint4(qc.v[4], qc.v[5], qc.v[6], qc.v[7]);
    }
    if (m >= 0) {
        sh_a[t] = col;
        __syncthreads();
        for (int off = QT / 2; off > 0; off >>= 1) {
            if (t < off) sh_a[t] = ff::add(sh_a[t], sh_a[t + off]);
            __syncthreads();
        }
        if (t == 0) {
            const ff::Fr qc = fmul(sh_a[0], sh_misc[2]);  // canonical column sum x z^-1 (Montgomery)
#pragma unroll
            for (int l = 0; l < 8; ++l) q_out[(blob * N + m) * 8 + l] = qc.v[l];
        }
    }
}

// The same computation for a handful of blobs (single compute_kzg_proof / compute_blob_kzg_proof calls), where one
// workgroup per blob is a latency chain: QS workgroups per blob, one element per lane, in two phases.
//   k_quotient_a : d_i = z - w_i, inverses by a product scan per workgroup (one binary-Euclid inversion each, all
//                  concurrent), partial barycentric sums, inv_i kept in `scratch`
//   k_quotient_b : y from the QS partial sums, q_i = (y - p_i) * inv_i; in the z-inside-the-domain case the column
//                  sum is finished by the last workgroup of the blob to arrive (atomic ticket)
// scratch per blob: N inverses, QS partial sums, QS partial column sums (Fr), then m (int) and a ticket (u32).
constexpr int QS = (int)(N / QT);  // workgroups per blob (8)
constexpr size_t QSCR_FR = N + 2 * QS;                   // Fr slots per blob
constexpr size_t QSCR_BYTES = QSCR_FR * sizeof(ff::Fr) + 16;  // + m, ticket

// m = -1 (z outside the domain), ticket = 0 for every blob of a k_quotient_a / _b pair
__global__ void __launch_bounds__(64) k_quotient_init(unsigned char* __restrict__ scratch, size_t nblobs) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblobs) return;
    int* w = reinterpret_cast<int*>(scratch + b * QSCR_BYTES + QSCR_FR * sizeof(ff::Fr));
    w[0] = -1;
    w[1] = w[2] = w[3] = 0;
}

__global__ void __launch_bounds__(QT) k_quotient_a(unsigned char* __restrict__ scratch, int* __restrict__ status,
                                                   const u32* __restrict__ blobs, const u32* __restrict__ z_be,
                                                   const ff::Fr* __restrict__ roots_brp) {
    __shared__ ff::Fr sh_a[QT];
    __shared__ ff::Fr sh_b[QT];
    __shared__ ff::Fr sh_inv;
    const int t = threadIdx.x;
    const size_t blob = blockIdx.x / QS;
    const int blk = (int)(blockIdx.x % QS);
    const int i = blk * QT + t;
    unsigned char* sc = scratch + blob * QSCR_BYTES;
    ff::Fr* sc_fr = reinterpret_cast<ff::Fr*>(sc);
    int* sc_m = reinterpret_cast<int*>(sc + QSCR_FR * sizeof(ff::Fr));
    bool zok;
    const ff::Fr z = ff::to_mont(fr_load_be(z_be + blob * 8, &zok));
    const ff::Fr w = roots_brp[i];
    ff::Fr d = ff::sub(z, w);
    if (d.is_zero()) {
        *sc_m = i;  // at most one lane of one workgroup
        d = ff::Fr::one();
    }
    sh_a[t] = d;
    sh_b[t] = d;
    __syncthreads();
    for (int off = 1; off < QT; off <<= 1) {
        ff::Fr pa = sh_a[t], pb = sh_b[t];
        if (t >= off) pa = fmul(sh_a[t - off], pa);
        if (t + off < QT) pb = fmul(pb, sh_b[t + off]);
        __syncthreads();
        sh_a[t] = pa;
        sh_b[t] = pb;
        __syncthreads();
    }
    if (t == 0) sh_inv = fr_inverse(sh_a[QT - 1]);
    __syncthreads();
    ff::Fr inv = sh_inv;
    if (t > 0) inv = fmul(inv, sh_a[t - 1]);
    if (t + 1 < QT) inv = fmul(inv, sh_b[t + 1]);  // 1 / d_i
    __syncthreads();
    bool ok;
    const ff::Fr p = fr_load_be(blobs + (blob * N + (size_t)i) * 8, &ok);  // canonical, as in k_quotient
    if (!ok || !zok) status[blob] = 1;
    sc_fr[i] = inv;
    sh_a[t] = fmul(fmul(inv, w), p);
    __syncthreads();
    for (int off = QT / 2; off > 0; off >>= 1) {
        if (t < off) sh_a[t] = ff::add(sh_a[t], sh_a[t + off]);
        __syncthreads();
    }
    if (t == 0) sc_fr[N + blk] = sh_a[0];
}

__global__ void __launch_bounds__(QT) k_quotient_b(u32* __restrict__ q_out, u32* __restrict__ y_out,
                                                   unsigned char* __restrict__ scratch, const u32* __restrict__ blobs,
                                                   const u32* __restrict__ z_be, const ff::Fr* __restrict__ roots_brp,
                                                   ff::Fr ninv) {
    __shared__ ff::Fr sh_a[QT];
    __shared__ ff::Fr sh_y, sh_zinv;
    __shared__ u32 sh_ticket;
    const int t = threadIdx.x;
    const size_t blob = blockIdx.x / QS;
    const int blk = (int)(blockIdx.x % QS);
    const int i = blk * QT + t;
    unsigned char* sc = scratch + blob * QSCR_BYTES;
    ff::Fr* sc_fr = reinterpret_cast<ff::Fr*>(sc);
    const int m = *reinterpret_cast<const int*>(sc + QSCR_FR * sizeof(ff::Fr));
    u32* ticket = reinterpret_cast<u32*>(sc + QSCR_FR * sizeof(ff::Fr) + 4);
    const u32* bw = blobs + blob * (N * 8);
    bool ok;
    const ff::Fr z = ff::to_mont(fr_load_be(z_be + blob * 8, &ok));
    if (t == 0) {
        ff::Fr y;  // canonical
        if (m >= 0) {
            y = fr_load_be(bw + (size_t)m * 8, &ok);
        } else {
            ff::Fr sum = sc_fr[N];
            for (int k = 1; k < QS; ++k) sum = ff::add(sum, sc_fr[N + k]);
            ff::Fr zn = z;
            for (int k = 0; k < 12; ++k) zn = fmul(zn, zn);
            y = fmul(fmul(sum, ninv), ff::sub(zn, ff::Fr::one()));  // sum / N * (z^N - 1)
        }
        sh_y = y;
        if (blk == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) y_out[blob * 8 + k] = y.v[k];
        }
    }
    __syncthreads();
    const ff::Fr y = sh_y;
    const ff::Fr inv = sc_fr[i];
    const ff::Fr p = fr_load_be(bw + (size_t)i * 8, &ok);
    const ff::Fr ymp = ff::sub(y, p);
    if (i != m) {
        const ff::Fr qc = fmul(ymp, inv);  // canonical x Montgomery
#pragma unroll
        for (int l = 0; l < 8; ++l) q_out[(blob * N + i) * 8 + l] = qc.v[l];
    }
    if (m < 0) return;  // (uniform over the blob)
    // domain case: column m gets  sum_{i != m} (p_i - y) * w_i / (z * (z - w_i))
    sh_a[t] = i == m ? ff::Fr::zero() : fmul(fmul(ff::neg(ymp), roots_brp[i]), inv);
    __syncthreads();
    for (int off = QT / 2; off > 0; off >>= 1) {
        if (t < off) sh_a[t] = ff::add(sh_a[t], sh_a[t + off]);
        __syncthreads();
    }
    if (t == 0) {
        sc_fr[N + QS + blk] = sh_a[0];
        __threadfence();
        sh_ticket = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    if (sh_ticket != (u32)QS - 1 || t != 0) return;
    __threadfence();
    ff::Fr col = sc_fr[N + QS];
    for (int k = 1; k < QS; ++k) col = ff::add(col, sc_fr[N + QS + k]);
    const ff::Fr qc = fmul(col, fr_inverse(z));
#pragma unroll
    for (int l = 0; l < 8; ++l) q_out[(blob * N + m) * 8 + l] = qc.v[l];
}

// ---- Fiat-Shamir challenge on the device (compute_challenge_rust, kzg/src/eip_4844.rs:920-945) ----
// SHA-256 is a serial chain over the 2050 blocks of  domain | 0 | 4096 | blob | commitment : one lane per blob, ~9 ms
// of latency whatever the batch, a percent of the chip's VALU time.  It pays when batches are pipelined on several
// streams (the hash of one batch runs under the MSM of another) and when host cores are scarce (eight ranks per node);
// the host-buffer entry points keep hashing small batches on the host's SHA units, where one blob takes 75 us.
__device__ __forceinline__ u32 sha_rotr(u32 x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
// gfx950's three-input bit operation (v_bitop3_b32, truth table in the immediate): a ^ b ^ c, Ch and Maj are ONE
// instruction each — 14 instead of 18 per compression round (the compiler finds Ch and Maj by itself, not the xors)
__device__ __forceinline__ u32 sha_xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ u32 sha_ch(u32 e, u32 f, u32 g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xca); }   // e ? f : g
__device__ __forceinline__ u32 sha_maj(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8); }
__device__ __forceinline__ void sha256_block(u32 h[8], u32 w[16]) {
    constexpr u32 K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u,
        0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u,
        0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u,
        0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
        0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u,
        0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au,
        0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u,
        0xc67178f2u};
    u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int t = 0; t < 64; ++t) {
        if (t >= 16) {
            const u32 w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
            const u32 s0 = sha_xor3(sha_rotr(w15, 7), sha_rotr(w15, 18), w15 >> 3);
            const u32 s1 = sha_xor3(sha_rotr(w2, 17), sha_rotr(w2, 19), w2 >> 10);
            w[t & 15] += s0 + w[(t - 7) & 15] + s1;
        }
        const u32 S1 = sha_xor3(sha_rotr(e, 6), sha_rotr(e, 11), sha_rotr(e, 25));
        const u32 ch = sha_ch(e, f, g);
        const u32 t1 = hh + S1 + ch + K[t] + w[t & 15];
        const u32 S0 = sha_xor3(sha_rotr(a, 2), sha_rotr(a, 13), sha_rotr(a, 22));
        const u32 maj = sha_maj(a, b, c);
        const u32 t2 = S0 + maj;
        hh = g;
        g = f;
        f = e;
        e = d + t1;
        d = c;
        c = b;
        b = a;
        a = t1 + t2;
    }
    h[0] += a;
    h[1] += b;
    h[2] += c;
    h[3] += d;
    h[4] += e;
    h[5] += f;
    h[6] += g;
    h[7] += hh;
}

// z_be[b] = hash_to_bls_field(sha256("FSBLOBVERIFY_V1_" | u64_be(0) | u64_be(4096) | blob_b | commitment_b)), 32 bytes
// big-endian.  Message = 8 + 32768 + 12 words; 2049 full blocks + one block of 4 words, padding and the bit length.
__global__ void __launch_bounds__(64) k_challenge_sha256(u32* __restrict__ z_be, const u32* __restrict__ blobs,
                                                         const u32* __restrict__ commitments, size_t n) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    __builtin_amdgcn_s_setprio(3);  // a handful of long serial waves next to throughput kernels: never starve them
    const u32* blob = blobs + b * (N * 8);
    const u32* cm = commitments + b * 12;
    u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    u32 w[16];
    // block 0: the 32-byte header and the first 8 words of the blob
    w[0] = 0x4653424cu;  // "FSBL"
    w[1] = 0x4f425645u;  // "OBVE"
    w[2] = 0x52494659u;  // "RIFY"
    w[3] = 0x5f56315fu;  // "_V1_"
    w[4] = 0;
    w[5] = 0;
    w[6] = 0;
    w[7] = (u32)N;
#pragma unroll
    for (int k = 0; k < 8; ++k) w[8 + k] = __builtin_bswap32(blob[k]);
    sha256_block(h, w);
    // blocks 1 .. 2047: blob words 16 blk - 8 .. 16 blk + 7
#pragma unroll 1
    for (int blk = 1; blk < 2048; ++blk) {
        const uint4* src = reinterpret_cast<const uint4*>(blob + 16 * blk - 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = src[q];
            w[4 * q] = __builtin_bswap32(v.x);
            w[4 * q + 1] = __builtin_bswap32(v.y);
            w[4 * q + 2] = __builtin_bswap32(v.z);
            w[4 * q + 3] = __builtin_bswap32(v.w);
        }
        sha256_block(h, w);
    }
    // block 2048: the last 8 words of the blob and the first 8 of the commitment
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = __builtin_bswap32(blob[N * 8 - 8 + k]);
#pragma unroll
    for (int k = 0; k < 8; ++k) w[8 + k] = __builtin_bswap32(cm[k]);
    sha256_block(h, w);
    // block 2049: the last 4 words of the commitment, 0x80, zeros, the length in bits (131152 * 8)
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = __builtin_bswap32(cm[8 + k]);
    w[4] = 0x80000000u;
#pragma unroll
    for (int k = 5; k < 15; ++k) w[k] = 0;
    w[14] = 0;
    w[15] = (u32)((32 + BYTES_PER_BLOB + 48) * 8);
    sha256_block(h, w);
    // hash_to_bls_field: the digest as a big-endian integer, reduced mod r
    ff::Fr v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v.v[i] = h[7 - i];
    const ff::Fr red = ff::from_mont(ff::mul(v, ff::Fr::r2()));
#pragma unroll
    for (int i = 0; i < 8; ++i) z_be[b * 8 + i] = __builtin_bswap32(red.v[7 - i]);
}

// The same hash with FOUR lanes per blob (16 blobs per wave).  SHA-256 has two halves: the message schedule of a block
// (W[16..63], 48 steps that depend on the block's words only) and the compression (64 rounds that depend on the
// previous block's state).  Only the compression is a chain.  The four lanes of a blob take the next four blocks, each
// expands ITS block's schedule and leaves W[t] + K[t] in LDS — one pass of the schedule code serves four blocks — and
// then lane 0 of the four runs the 4 x 64 rounds with one LDS operand per round.  Per block the wave issues ~150 + 980
// instructions instead of ~2000, so the chain is about half as long (a lone wave issues an instruction every ~5 cycles
// whatever its lane count), on four times the waves: 64 waves per 1024 blobs, ~3 % of what the batch's MSM issues.
// (One blob per WAVE would shorten the chain no further — the compression rounds are the chain — and cost 16 x more.)
constexpr int SHA_Q = 4;                 // lanes (= blocks in flight) per blob
constexpr int SHA_BLOBS = 64 / SHA_Q;    // blobs per wave
constexpr int SHA_BLOCKS = 2050;         // (32 + 131072 + 48 + 1 + 8 bytes) rounded up to 64-byte blocks
__device__ __forceinline__ void sha256_rounds_wk(u32 h[8], const uint4* __restrict__ wk) {
    u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int t4 = 0; t4 < 16; ++t4) {
        const uint4 v = wk[t4];
        const u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 S1 = sha_xor3(sha_rotr(e, 6), sha_rotr(e, 11), sha_rotr(e, 25));
            const u32 ch = sha_ch(e, f, g);
            const u32 t1 = hh + S1 + ch + x[k];
            const u32 S0 = sha_xor3(sha_rotr(a, 2), sha_rotr(a, 13), sha_rotr(a, 22));
            const u32 maj = sha_maj(a, b, c);
            hh = g;
            g = f;
            f = e;
            e = d + t1;
            d = c;
            c = b;
            b = a;
            a = t1 + S0 + maj;
        }
    }
    h[0] += a;
    h[1] += b;
    h[2] += c;
    h[3] += d;
    h[4] += e;
    h[5] += f;
    h[6] += g;
    h[7] += hh;
}
// word m (0 .. 16 * SHA_BLOCKS - 1) of the padded message, big-endian
__device__ __forceinline__ u32 sha_msg_word(int m, const u32* __restrict__ blob, const u32* __restrict__ cm) {
    constexpr int NB = N * 8;  // words of a blob
    if (m < 8) return m == 0 ? 0x4653424cu : m == 1 ? 0x4f425645u : m == 2 ? 0x52494659u : m == 3 ? 0x5f56315fu : m == 7 ? (u32)N : 0u;
    if (m < 8 + NB) return __builtin_bswap32(blob[m - 8]);
    if (m < 8 + NB + 12) return __builtin_bswap32(cm[m - 8 - NB]);
    if (m == 8 + NB + 12) return 0x80000000u;
    if (m == 16 * SHA_BLOCKS - 1) return (u32)((32 + BYTES_PER_BLOB + 48) * 8);
    return 0u;
}
__global__ void __launch_bounds__(64) k_challenge_sha256_quad(u32* __restrict__ z_be, const u32* __restrict__ blobs,
                                                              const u32* __restrict__ commitments, size_t n) {
    // per blob: SHA_Q blocks x 16 uint4 of W + K, padded by one uint4 so that the 16 lanes that compress read 64 banks
    __shared__ uint4 wk[SHA_BLOBS][SHA_Q * 16 + 1];
    constexpr u32 K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u,
        0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u,
        0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u,
        0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
        0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u,
        0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au,
        0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u,
        0xc67178f2u};
    const int lane = threadIdx.x, q = lane / SHA_Q, j = lane % SHA_Q;
    const size_t b_raw = (size_t)blockIdx.x * SHA_BLOBS + q;
    const bool live = b_raw < n;
    const size_t b = live ? b_raw : n - 1;  // lanes past the end hash the last blob again and store nothing
    __builtin_amdgcn_s_setprio(3);  // a handful of long serial waves next to throughput kernels: never starve them
    const u32* blob = blobs + b * (N * 8);
    const u32* cm = commitments + b * 12;
    u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    constexpr int GROUPS = (SHA_BLOCKS + SHA_Q - 1) / SHA_Q;
#pragma unroll 1
    for (int g = 0; g < GROUPS; ++g) {
        const int blk = g * SHA_Q + j;
        u32 w[16];
        if (blk >= 1 && blk < 2048) {  // sixteen words of the blob: words 16 blk - 8 .. 16 blk + 7
            const uint4* src = reinterpret_cast<const uint4*>(blob + 16 * blk - 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 v = src[k];
                w[4 * k] = __builtin_bswap32(v.x);
                w[4 * k + 1] = __builtin_bswap32(v.y);
                w[4 * k + 2] = __builtin_bswap32(v.z);
                w[4 * k + 3] = __builtin_bswap32(v.w);
            }
        } else {  // the header block, the two blocks with the commitment and the padding, blocks past the end (unused)
#pragma unroll
            for (int k = 0; k < 16; ++k) w[k] = blk < SHA_BLOCKS ? sha_msg_word(16 * blk + k, blob, cm) : 0u;
        }
        fpw::wave_sync();  // the compression of the previous group has read its operands
        uint4* dst = &wk[q][j * 16];
#pragma unroll
        for (int t4 = 0; t4 < 16; ++t4) {
            u32 x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = 4 * t4 + k;
                if (t >= 16) {
                    const u32 w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
                    const u32 s0 = sha_xor3(sha_rotr(w15, 7), sha_rotr(w15, 18), w15 >> 3);
                    const u32 s1 = sha_xor3(sha_rotr(w2, 17), sha_rotr(w2, 19), w2 >> 10);
                    w[t & 15] += s0 + w[(t - 7) & 15] + s1;
                }
                x[k] = w[t & 15] + K[t];
            }
            dst[t4] = make_uint4(x[0], x[1], x[2], x[3]);
        }
        fpw::wave_sync();
        if (j == 0) {
            const int left = SHA_BLOCKS - g * SHA_Q;
#pragma unroll 1
            for (int jj = 0; jj < SHA_Q && jj < left; ++jj) sha256_rounds_wk(h, &wk[q][jj * 16]);
        }
    }
    if (j != 0 || !live) return;
    // hash_to_bls_field: the digest as a big-endian integer, reduced mod r
    ff::Fr v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v.v[i] = h[7 - i];
    const ff::Fr red = ff::from_mont(ff::mul(v, ff::Fr::r2()));
#pragma unroll
    for (int i = 0; i < 8; ++i) z_be[b * 8 + i] = __builtin_bswap32(red.v[7 - i]);
}
// launch of the challenge hashes of n blobs on `st`.  Tuning key sha_lanes: 4 = four lanes per blob (the chain ~0.6 as
// long, 2.4 x the instructions), 1 = one lane per blob, 0 (default) = by batch size: four lanes up to 512 blobs — a call
// that waits for its hashes — one lane for the large batches of a pipelined caller, whose hashes run under other
// batches' MSMs and only cost what they issue (measured, 1024 blobs on four streams: 77.7 k proofs/s against 73.5 k).
static inline void launch_challenge_sha256(u32* z_be, const u32* blobs, const u32* commitments, size_t n, int lanes, hipStream_t st) {
    if (lanes == 0) lanes = n <= 512 ? 4 : 1;
    if (lanes == 1)
        hipLaunchKernelGGL(k_challenge_sha256, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, z_be, blobs, commitments, n);
    else
        hipLaunchKernelGGL(k_challenge_sha256_quad, dim3((unsigned)((n + SHA_BLOBS - 1) / SHA_BLOBS)), dim3(64), 0, st, z_be, blobs,
                           commitments, n);
}

// the r-torsion test of a decoded point: phi(P) == -[x^2]P, phi(x,y) = (beta*x, y), x the BLS parameter (the in-tree
// statement of the same test: zkcrypto/bls12_381/src/g1.rs:401-435).  Two 64-bit scalar multiplications instead of one
// by the 255-bit group order.
__device__ __forceinline__ bool affpt_in_g1(const AffPt& p) {
    if (p.flags & 1) return true;
    const unsigned long long BLS_X = 0xd201000000010000ull;  // |x|; the sign cancels in x^2
    g1::Xyzz q1, q2;
    g1::set_inf(q1);
    for (int bit = 63; bit >= 0; --bit) {
        if (!g1::is_inf(q1)) g1::dbl(q1);
        if ((BLS_X >> bit) & 1) g1::madd(q1, p.x, p.y);
    }
    g1::set_inf(q2);
    for (int bit = 63; bit >= 0; --bit) {
        if (!g1::is_inf(q2)) g1::dbl(q2);
        if ((BLS_X >> bit) & 1) g1::dadd(q2, q1);
    }
    if (g1::is_inf(q2)) return false;
    fp28::Fe beta;
    {
        constexpr u32 t[14] = {0xa75929au, 0x681b798u, 0x22a3e9du, 0xabc02bfu, 0x4e5bb45u, 0x55e6e7eu, 0x4814117u,
                               0x6d04f1bu, 0xae3387du, 0x54acb0cu, 0xa4c74bu, 0x56138b5u, 0xb64e066u, 0x76f2u};
#pragma unroll
        for (int k = 0; k < 14; ++k) beta.v[k] = t[k];  // cube root of unity, Montgomery 2^392
    }
    // phi(P) == -Q2  <=>  beta*x*ZZ == X  and  y*ZZZ == -Y
    const fp28::Fe dx = fp28::sub<16>(fp28::mul(fp28::mul(beta, p.x), q2.zz), q2.x);
    const fp28::Fe dy = fp28::addn(fp28::mul(p.y, q2.zzz), q2.y);
    return fp28::is_zero_mod_p(dx) && fp28::is_zero_mod_p(dy);
}

// compressed bytes -> table slots + status: 0 ok, 1 not a valid encoding, 2 on the curve but outside the r-torsion
// subgroup (one square root per point: the decode and the membership test of batched verification in one kernel)
template <bool CHECK>
__global__ void __launch_bounds__(64) k_decode_check_g1(AffPt* __restrict__ out, int* __restrict__ status,
                                                        const unsigned char* __restrict__ in, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned char buf[48];
    for (int k = 0; k < 48; ++k) buf[k] = in[48 * i + k];
    AffPt p;
    if (!g1io::uncompress(p, buf)) {
        status[i] = 1;
        return;
    }
    out[i] = p;
    status[i] = !CHECK || affpt_in_g1(p) ? 0 : 2;
}

// The decode of k_decode_check_g1<false> with the square root limb-parallel, FOUR POINTS PER WAVE (one per DPP row of
// fpw.hip.h).  y = (x^3 + 4)^((p+1)/4) is a chain of 380 squarings that nothing shortens: in one lane 570 multiplications
// of ~1 us each (the plain square-and-multiply of g1io::pow_sat), here 4-bit windows — 14 table products, then 4
// squarings + at most one product per nibble of the exponent, 95 nibbles — of one row-parallel multiplication step each.
// Lane 0 of a row parses its point and computes x^3 + 4 before the chain, and checks the root, picks the sign and writes
// the slot after it: the same single-lane code as g1io::uncompress on either side of the chain.
__global__ void __launch_bounds__(64) k_decode_g1_wide(AffPt* __restrict__ out, int* __restrict__ status,
                                                       const unsigned char* __restrict__ in, size_t n) {
    __shared__ u32 sh_in[4][16], sh_out[4][16];
    const int lane = threadIdx.x, row = lane >> 4, li = lane & 15;
    const size_t i = (size_t)blockIdx.x * 4 + row;
    const bool mine = li == 0 && i < n;
    bool chain = false, sort = false;
    fp28::Fe x = fp28::zero(), y2 = fp28::zero();
    sh_in[row][li] = 0;
    fpw::wave_sync();
    if (mine) {
        unsigned char buf[48];
        for (int k = 0; k < 48; ++k) buf[k] = in[48 * i + k];
        const bool compressed = (buf[0] >> 7) & 1, infinity = (buf[0] >> 6) & 1;
        sort = (buf[0] >> 5) & 1;
        buf[0] &= 0x1f;
        const ff::Fp xs = g1io::be48_to_sat(buf);
        int st = 1;  // not a valid encoding, unless ...
        if (compressed && infinity) {
            if (!sort && xs.is_zero()) {
                AffPt o;
                o.flags = 1;
                o.pad[0] = o.pad[1] = o.pad[2] = 0;
                o.x = fp28::zero();
                o.y = fp28::zero();
                out[i] = o;
                st = 0;
            }
        } else if (compressed && !g1io::sat_geq_p(xs)) {
            x = g1io::from_plain(xs);
            fp28::Fe b4;
#pragma unroll
            for (int k = 0; k < 14; ++k) b4.v[k] = g1io::b4_392_l(k);
            y2 = fp28::addn(fp28::mul(fp28::sqr(x), x), b4);  // x^3 + 4, < 4p
#pragma unroll
            for (int k = 0; k < 14; ++k) sh_in[row][k] = y2.v[k];
            chain = true;
        }
        if (!chain) status[i] = st;
    }
    fpw::wave_sync();
    // the chain (rows without a point run it on zero)
    const fpw::Lane lc = fpw::lane_consts(lane);
    const u32 a = sh_in[row][li];
    u32 t[16];
    t[0] = 0;
    t[1] = a;
#pragma unroll
    for (int k = 2; k < 16; ++k) t[k] = fpw::wmul4(t[k - 1], a, lc);
    auto nibble = [](int j) -> u32 { return (g1io::p_sqrt_exp_l(j >> 3) >> (4 * (j & 7))) & 15u; };
    auto pick = [&](u32 d) -> u32 {
        u32 v = t[1];
#pragma unroll
        for (int k = 2; k < 16; ++k) v = d == (u32)k ? t[k] : v;
        return v;
    };
    u32 acc = pick(nibble(94));  // the top nibble of (p+1)/4 is not zero
#pragma unroll 1
    for (int j = 93; j >= 0; --j) {
#pragma unroll 1
        for (int q = 0; q < 4; ++q) acc = fpw::wmul4(acc, acc, lc);
        const u32 d = nibble(j);
        if (d) acc = fpw::wmul4(acc, pick(d), lc);
    }
    sh_out[row][li] = fpw::wnorm_full(acc, lc);
    fpw::wave_sync();
    if (mine && chain) {
        fp28::Fe y;
#pragma unroll
        for (int k = 0; k < 14; ++k) y.v[k] = sh_out[row][k];
        const fp28::Fe chk = fp28::sub<8>(fp28::sqr(y), y2);
        if (!fp28::is_zero_mod_p(chk)) {
            status[i] = 1;  // x^3 + 4 is not a square: no such point
        } else {
            y = fp28::canon(y);
            if (g1io::is_lex_largest(g1io::to_plain(y)) != sort) y = fp28::canon(fp28::neg<2>(y));
            AffPt o;
            o.flags = 0;
            o.pad[0] = o.pad[1] = o.pad[2] = 0;
            o.x = fp28::canon(x);
            o.y = y;
            out[i] = o;
            status[i] = 0;
        }
    }
}

// The membership test of affpt_in_g1 with ONE WAVE PER POINT (g1w: the limbs of a coordinate across the lanes of a DPP
// row, the independent products of a point formula in the four rows): 2 x (63 doublings + 5 additions) of 3 / 4
// multiplication steps each instead of ~1 000 single-lane multiplications — the test is a latency chain (1.1 ms in one
// lane) in front of every batched verification, and a few hundred points leave the chip empty anyway.
// status[i] != 0 (not decoded) is left alone; a point that fails gets 2.
__device__ __forceinline__ void wide_mul_by_abs_x(g1w::WPt& acc, const g1w::WPt& b, const fpw::Lane& lc, u32* sh, int lane) {
    // |x| = 0xd201000000010000: bits 63, 62, 60, 57, 48, 16
    acc = b;
    const int runs[6] = {1, 2, 3, 9, 32, 16};
#pragma unroll 1
    for (int r = 0; r < 6; ++r) {
#pragma unroll 1
        for (int k = 0; k < runs[r]; ++k) g1w::dbl(acc, lc, lane);  // odd group order: never infinity
        if (r < 5) g1w::dadd(acc, b, lc, sh, lane);
    }
}
__global__ void __launch_bounds__(64) k_affpts_in_g1_wide(int* __restrict__ status, const AffPt* __restrict__ pts, size_t n) {
    __shared__ u32 sh[16];
    const size_t i = blockIdx.x;
    const int lane = threadIdx.x;
    if (i >= n || status[i] != 0) return;
    const u32* w = reinterpret_cast<const u32*>(pts + i);
    if (w[2 * fp28::L] & 1u) return;  // infinity
    const fpw::Lane lc = fpw::lane_consts(lane);
    const int li = lane & 15;
    g1w::WPt P, q1, q2;
    P.x = li < fp28::L ? w[li] : 0u;
    P.y = li < fp28::L ? w[fp28::L + li] : 0u;
    P.zz = P.zzz = fpw::to_wide(fp28::one(), sh, lane);
    wide_mul_by_abs_x(q1, P, lc, sh, lane);
    wide_mul_by_abs_x(q2, q1, lc, sh, lane);
    bool ok = !g1w::is_inf(q2);
    if (ok) {
        fp28::Fe beta;
        {
            constexpr u32 t[14] = {0xa75929au, 0x681b798u, 0x22a3e9du, 0xabc02bfu, 0x4e5bb45u, 0x55e6e7eu, 0x4814117u,
                                   0x6d04f1bu, 0xae3387du, 0x54acb0cu, 0xa4c74bu, 0x56138b5u, 0xb64e066u, 0x76f2u};
#pragma unroll
            for (int k = 0; k < 14; ++k) beta.v[k] = t[k];
        }
        const u32 bw = fpw::to_wide(beta, sh, lane);
        // phi(P) == -Q2  <=>  beta*x*ZZ == X  and  y*ZZZ == -Y   (as in affpt_in_g1)
        const u32 dx = fpw::wsub32(fpw::wmul(fpw::wmul(bw, P.x, lc), q2.zz, lc), q2.x, lc);
        const u32 dy = fpw::waddn(fpw::wmul(P.y, q2.zzz, lc), q2.y, lc);
        ok = g1w::is_zero_mod_p(dx, sh, lane) && g1w::is_zero_mod_p(dy, sh, lane);
    }
    if (!ok && lane == 0) status[i] = 2;
}

}  // namespace
namespace ckz {
// `decoded` (optional) is recorded when the slots are written — before the membership test in the two-kernel form, so a
// consumer that only needs the points (the MSM of a verification call, whose result is thrown away if a test fails) can
// start while the test still runs
void decode_check_enqueue(AffPt* d_pts, int* d_stat, const unsigned char* d_bytes, size_t np, hipStream_t st, bool wide,
                          hipEvent_t decoded) {
    const dim3 grid((unsigned)((np + 63) / 64));
    if (wide && np <= WIDE_CHECK_MAX) {
        hipLaunchKernelGGL(k_decode_g1_wide, dim3((unsigned)((np + 3) / 4)), dim3(64), 0, st, d_pts, d_stat, d_bytes, np);
        if (decoded) (void)hipEventRecord(decoded, st);
        hipLaunchKernelGGL(k_affpts_in_g1_wide, dim3((unsigned)np), dim3(64), 0, st, d_stat, (const AffPt*)d_pts, np);
    } else {
        hipLaunchKernelGGL(k_decode_check_g1<true>, grid, dim3(64), 0, st, d_pts, d_stat, d_bytes, np);
        if (decoded) (void)hipEventRecord(decoded, st);
    }
}
}  // namespace ckz
namespace {

// commitment bytes -> status: 0 ok (valid encoding, and infinity or in the r-torsion subgroup), 1 bad
// (FsG1::from_bytes + `!is_inf && !is_valid`, kzg/src/eip_4844.rs:556-558,577)
__global__ void __launch_bounds__(64) k_check_commitments(int* __restrict__ status, const unsigned char* __restrict__ in,
                                                          size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned char buf[48];
    for (int k = 0; k < 48; ++k) buf[k] = in[48 * i + k];
    AffPt p;
    if (!g1io::uncompress(p, buf) || !affpt_in_g1(p)) status[i] = 1;
}



std::mutex g_registry_mu;
std::map<const void*, KzgAmdSettings*> g_registry;

}  // namespace
namespace ckz {
KzgAmdSettings* lookup(const CKZGSettings* s) {
    if (!s || !s->g1_values_lagrange_brp) return nullptr;
    std::lock_guard<std::mutex> lk(g_registry_mu);
    auto it = g_registry.find(s->g1_values_lagrange_brp);
    return it == g_registry.end() ? nullptr : it->second;
}

// A lane of `parent`: own streams, staging and mutex; tables and engine handles borrowed (see KzgAmdSettings::lanes)
KzgAmdSettings* make_lane(KzgAmdSettings* parent) {
    std::unique_ptr<KzgAmdSettings> ln(new KzgAmdSettings());
    ln->apply_options(parent->opt);
    ln->is_lane = true;
    ln->device = parent->device;
    kzgamd::DeviceGuard on_device(parent->device);
    CK_HIP(on_device.err);
    // ONE stream per lane: the runtime deals streams round-robin onto its (by default four) hardware queues, and a
    // second, idle stream per lane would put every lane's working stream on the same two queues
    CK_HIP(hipStreamCreateWithFlags(&ln->stream, hipStreamNonBlocking));
    ln->stream2 = ln->stream;
    ln->msm = parent->msm;
    // the MSM workspace of this lane's stream, sized now for every batch size a lane runs: no call on a lane allocates
    for (size_t nb = KzgAmdSettings::LANE_MAX_BLOBS; nb >= 1; --nb) {
        kzgamd::msm_lock(parent->msm);
        try {
            kzgamd::msm_enqueue(parent->msm, nullptr, nullptr, N, nb, 0, ln->stream, kzgamd::OUT_COMPRESSED, true);
        } catch (...) {
            kzgamd::msm_unlock(parent->msm);
            throw;
        }
        kzgamd::msm_unlock(parent->msm);
    }
    ln->d_brp_roots = parent->d_brp_roots;
    ln->brp_roots = parent->brp_roots;
    ln->busy.store(true);
    parent->lanes.push_back(std::move(ln));
    return parent->lanes.back().get();
}
}  // namespace ckz
namespace {

// ---- trusted setup text (kzg/src/eip_4844.rs:151-228) ----
bool is_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
int hexval(unsigned char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}
bool scan_number(const std::string& s, size_t& off, size_t& out) {
    while (off < s.size() && is_ws((unsigned char)s[off])) ++off;
    if (off >= s.size()) return false;
    size_t start = off;
    while (off < s.size() && s[off] >= '0' && s[off] <= '9') ++off;
    if (off >= s.size() || off == start || off - start > 18) return false;
    out = 0;
    for (size_t i = start; i < off; ++i) out = out * 10 + (size_t)(s[i] - '0');
    return true;
}
bool scan_hex_byte(const std::string& s, size_t& off, uint8_t& out) {
    while (off < s.size() && is_ws((unsigned char)s[off])) ++off;
    if (off >= s.size()) return false;
    int hi = hexval((unsigned char)s[off]);
    if (hi < 0) return false;
    if (off + 1 < s.size() && hexval((unsigned char)s[off + 1]) >= 0) {
        out = (uint8_t)(hi * 16 + hexval((unsigned char)s[off + 1]));
        off += 2;
    } else {
        out = (uint8_t)hi;
        off += 1;
    }
    return true;
}
void parse_setup_text(const std::string& text, std::vector<uint8_t>& g1_mono, std::vector<uint8_t>& g1_lag,
                      std::vector<uint8_t>& g2_mono) {
    size_t off = 0, n1 = 0, n2 = 0;
    CK_REQUIRE(scan_number(text, off, n1) && n1 == N, "Incorrect trusted setup format");
    CK_REQUIRE(scan_number(text, off, n2) && n2 == NUM_G2, "Incorrect trusted setup format");
    g1_lag.resize(48 * N);
    g2_mono.resize(96 * NUM_G2);
    g1_mono.resize(48 * N);
    for (auto& b : g1_lag) CK_REQUIRE(scan_hex_byte(text, off, b), "Incorrect trusted setup format");
    for (auto& b : g2_mono) CK_REQUIRE(scan_hex_byte(text, off, b), "Incorrect trusted setup format");
    for (auto& b : g1_mono) CK_REQUIRE(scan_hex_byte(text, off, b), "Incorrect trusted setup format");
}

template <class T>
T* leak_array(size_t n) {
    T* p = (T*)calloc(n, sizeof(T));
    if (!p) throw CkErr{C_KZG_MALLOC, "out of memory"};
    return p;
}

void zero_settings(CKZGSettings* out) { memset(out, 0, sizeof *out); }

void free_host_arrays(CKZGSettings* s) {
    free(s->roots_of_unity);
    free(s->brp_roots_of_unity);
    free(s->reverse_roots_of_unity);
    free(s->g1_values_monomial);
    free(s->g1_values_lagrange_brp);
    free(s->g2_values_monomial);
    if (s->x_ext_fft_columns) {
        for (size_t i = 0; i < 2 * CELLS_PER_BLOB; ++i) free(s->x_ext_fft_columns[i]);
        free(s->x_ext_fft_columns);
    }
    zero_settings(s);
}

// load_trusted_setup_rust (kzg/src/eip_4844.rs:1022-1086) with the G1 work on the device
void load_impl(CKZGSettings* out, const uint8_t* g1_mono, size_t n1m, const uint8_t* g1_lag, size_t n1l,
               const uint8_t* g2_mono, size_t n2, const KzgAmdConfig* cfg) {
    kzgamd::Options opt;
    {
        std::string err;
        if (!kzgamd::Options::resolve(opt, cfg, &err)) throw CkErr{C_KZG_BADARGS, err};
    }
    CK_REQUIRE(n1m / 48 == N && n1m % 48 == 0, "Invalid number of G1 points");
    CK_REQUIRE(n1l / 48 == N && n1l % 48 == 0, "Invalid number of G1 points");
    CK_REQUIRE(n2 / 96 == NUM_G2 && n2 % 96 == 0, "Invalid number of G2 points");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw CkErr{C_KZG_ERROR, "no gfx950 device visible"};
    // G2 points: decoded and checked to be on the twist (TG2::from_bytes = blst_p2_uncompress, blst/src/types/g2.rs:52-75)
    // on the host; they feed the pairing checks of the verify_* entry points and the Lagrange-form check below
    std::vector<kzgamd::pairing::G2Jac> g2(NUM_G2);
    for (size_t i = 0; i < NUM_G2; ++i)
        CK_REQUIRE(kzgamd::pairing::g2_uncompress(g2[i], g2_mono + 96 * i), "Failed to uncompress G2 point");

    auto* dev = new KzgAmdSettings();
    dev->apply_options(opt);
    unsigned char* d_bytes = nullptr;
    AffPt* d_pts = nullptr;
    int* d_bad = nullptr;
    ff::Fp* d_p1 = nullptr;
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    kzgamd::DeviceGuard placed(opt.device >= 0 ? opt.device : cur_dev);  // the caller's device is restored on return
    try {
        CK_HIP(placed.err);
        CK_HIP(hipGetDevice(&dev->device));
        CK_HIP(hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking));
        // stream2 carries the long one-lane latency chains that nothing waits for until the end of a call (the
        // commitment checks of a proof batch: 1.9 ms): a low-priority stream gets a hardware queue of its own — on a
        // queue shared with a pipeline stream it held that stream's chunk back until it was done (kernel trace of a
        // 256-blob proof call: chunk 2 started when k_check_commitments ended)
        {
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
            CK_HIP(hipStreamCreateWithPriority(&dev->stream2, hipStreamNonBlocking, least));
        }
        // bytes: [0,N) monomial, [N,2N) Lagrange in bit-reversed order (reverse_bit_order, eip_4844.rs:1070)
        std::vector<uint8_t> stage(2 * N * 48);
        memcpy(stage.data(), g1_mono, N * 48);
        for (size_t i = 0; i < N; ++i) memcpy(&stage[(N + i) * 48], g1_lag + 48 * reverse_bits(i, 12), 48);
        CK_HIP(hipMalloc(&d_bytes, stage.size()));
        CK_HIP(hipMalloc(&d_pts, 2 * N * sizeof(AffPt)));
        CK_HIP(hipMalloc(&d_bad, sizeof(int)));
        CK_HIP(hipMalloc(&d_p1, 2 * N * 144));
        CK_HIP(hipMemcpyAsync(d_bytes, stage.data(), stage.size(), hipMemcpyHostToDevice, dev->stream));
        CK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), dev->stream));
        hipLaunchKernelGGL(k_uncompress, dim3((unsigned)((2 * N + 127) / 128)), dim3(128), 0, dev->stream, d_pts, d_bad,
                           d_bytes, 2 * N);
        hipLaunchKernelGGL(k_affpt_to_blst_p1, dim3((unsigned)((2 * N + 255) / 256)), dim3(256), 0, dev->stream, d_p1,
                           d_pts, 2 * N);
        int bad = 0;
        CK_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, dev->stream));
        CK_HIP(hipStreamSynchronize(dev->stream));
        CK_REQUIRE(bad == 0, "Invalid G1 point in trusted setup");

        out->g1_values_monomial = leak_array<blst_p1>(N);
        out->g1_values_lagrange_brp = leak_array<blst_p1>(N);
        CK_HIP(hipMemcpy(out->g1_values_monomial, d_p1, N * 144, hipMemcpyDeviceToHost));
        CK_HIP(hipMemcpy(out->g1_values_lagrange_brp, d_p1 + 3 * N, N * 144, hipMemcpyDeviceToHost));

        // fixed-base MSM table over the bit-reversed Lagrange points (FsKZGSettings::new ->
        // prepare_msm, blst/src/types/kzg_settings.rs:109-123)
        dev->opt.device = -1;  // dev->device names the GPU from here on
        dev->msm = kzgamd::msm_create(d_pts + N, N, true, true, true, kzgamd::G1_TRUSTED, &dev->opt);
        CK_HIP(hipMalloc(&dev->d_monomial, N * sizeof(AffPt)));
        CK_HIP(hipMemcpy(dev->d_monomial, d_pts, N * sizeof(AffPt), hipMemcpyDeviceToDevice));

        // is_trusted_setup_in_lagrange_form (eip_4844.rs:1005-1020), the reference's own test: a file whose
        // "Lagrange" section is the monomial setup satisfies e(L_1, G2_0) == e(L_0, G2_1) (L_1 = tau * L_0).
        // L_0, L_1 are the first two points in FILE order: positions 0 and brp(1) = N/2 of the bit-reversed array.
        out->g2_values_monomial = leak_array<blst_p2>(NUM_G2);
        memcpy(out->g2_values_monomial, g2.data(), NUM_G2 * sizeof(blst_p2));
        dev->g2_monomial = g2;
        CK_REQUIRE(!kzgamd::pairing::pairings_verify(&out->g1_values_lagrange_brp[N / 2], &out->g2_values_monomial[0],
                                                     &out->g1_values_lagrange_brp[0], &out->g2_values_monomial[1]),
                   "Trusted setup is not in Lagrange form");

        // FsFFTSettings::new(13) (blst/src/types/fft_settings.rs:30-58)
        const size_t W = 2 * N;
        out->roots_of_unity = leak_array<blst_fr>(W + 1);
        out->reverse_roots_of_unity = leak_array<blst_fr>(W + 1);
        out->brp_roots_of_unity = leak_array<blst_fr>(W);
        std::vector<ff::Fr> roots;
        kzgamd::expand_roots(roots, 13);
        dev->brp_roots.resize(W);
        for (size_t i = 0; i <= W; ++i) {
            memcpy(&out->roots_of_unity[i], &roots[i], 32);
            memcpy(&out->reverse_roots_of_unity[i], &roots[W - i], 32);
        }
        for (size_t i = 0; i < W; ++i) {
            dev->brp_roots[i] = roots[reverse_bits(i, 13)];
            memcpy(&out->brp_roots_of_unity[i], &dev->brp_roots[i], 32);
        }
        CK_HIP(hipMalloc(&dev->d_brp_roots, N * sizeof(ff::Fr)));
        CK_HIP(hipMemcpy(dev->d_brp_roots, dev->brp_roots.data(), N * sizeof(ff::Fr), hipMemcpyHostToDevice));
        // FK20 columns of the settings struct (FsKZGSettings::new, blst/src/types/kzg_settings.rs:84-101): for every
        // offset < 64 the size-128 G1 transform of [ s^(N - 64 - 1 - offset - 64 i) ]_{i < 63}, identity, 64 x identity;
        // x_ext_fft_columns[row][offset] = transform[row].  This library's own cell proofs do not use them (they are
        // fixed-base MSMs), but a consumer of the struct — the reference's FK20 / recovery code — does: one batch of
        // 64 transforms through fft_g1 on the GPU.  tables / wbits / scratch_size stay empty like the reference's
        // (blst/src/eip_4844.rs:140-142).
        {
            const size_t K2 = 2 * CELLS_PER_BLOB;
            std::vector<blst_p1> xin(CELL_SIZE * K2), xout(CELL_SIZE * K2);
            memset(xin.data(), 0, xin.size() * sizeof(blst_p1));
            for (size_t offset = 0; offset < CELL_SIZE; ++offset) {
                const size_t start = N - CELL_SIZE - 1 - offset;
                for (size_t i = 0; i + 1 < CELLS_PER_BLOB; ++i) xin[offset * K2 + i] = out->g1_values_monomial[start - i * CELL_SIZE];
            }
            dev->ntt = kzgamd::ntt_create(13, dev->opt);
            if (!dev->ntt) throw CkErr{C_KZG_ERROR, "kzgamd_ntt_new failed"};
            if (kzgamd_fft_g1_batch(dev->ntt, xout.data(), xin.data(), K2, CELL_SIZE, 0) != 0) throw CkErr{C_KZG_ERROR, "fft_g1"};
            out->x_ext_fft_columns = leak_array<blst_p1*>(K2);
            for (size_t row = 0; row < K2; ++row) {
                out->x_ext_fft_columns[row] = leak_array<blst_p1>(CELL_SIZE);
                for (size_t offset = 0; offset < CELL_SIZE; ++offset) out->x_ext_fft_columns[row][offset] = xout[offset * K2 + row];
            }
        }
        (void)hipFree(d_bytes);
        (void)hipFree(d_pts);
        (void)hipFree(d_bad);
        (void)hipFree(d_p1);
        std::lock_guard<std::mutex> lk(g_registry_mu);
        g_registry[out->g1_values_lagrange_brp] = dev;
    } catch (...) {
        if (d_bytes) (void)hipFree(d_bytes);
        if (d_pts) (void)hipFree(d_pts);
        if (d_bad) (void)hipFree(d_bad);
        if (d_p1) (void)hipFree(d_p1);
        delete dev;
        free_host_arrays(out);
        throw;
    }
}

// device pipeline: blobs (device) -> 48-byte commitments (device)
// A compressed result costs the GPU a field inversion in one lane (~0.15 ms of latency whatever the batch); for a
// handful of results the host-buffer entry points fetch Jacobian points instead and compress them on the host
// (~20 us each): a single blob_to_kzg_commitment call 0.71 -> 0.5 ms.
}  // namespace
namespace ckz {
void compress_on_host(uint8_t* out48, const blst_p1* jac, size_t n) { kzgamd::host_p1_compress_batch(out48, jac, n); }
}  // namespace ckz
namespace {

void commit_enqueue(KzgAmdSettings* dev, void* d_out, int* d_status, const void* d_blobs, u32* d_scalars, size_t n,
                    hipStream_t stream, int out_mode = kzgamd::OUT_COMPRESSED, const BlobPtrs* ptrs = nullptr) {
    CK_HIP(hipMemsetAsync(d_status, 0, n * sizeof(int), stream));
    if (ptrs)  // a lane batch: one (page-locked host) pointer per blob
        hipLaunchKernelGGL(k_blob_to_scalars_ptrs, dim3((unsigned)((n * N + 255) / 256)), dim3(256), 0, stream, d_scalars,
                           d_status, *ptrs, n);
    else
        hipLaunchKernelGGL(k_blob_to_scalars, dim3((unsigned)((n * N + 255) / 256)), dim3(256), 0, stream, d_scalars, d_status,
                           (const u32*)d_blobs, n);
    kzgamd::msm_lock(dev->msm);
    try {
        kzgamd::msm_enqueue(dev->msm, d_out, d_scalars, N, n, 0, stream, out_mode);
    } catch (...) {
        kzgamd::msm_unlock(dev->msm);
        throw;
    }
    kzgamd::msm_unlock(dev->msm);
}


// 1 / FIELD_ELEMENTS_PER_BLOB in Montgomery form (the barycentric formula's 1/N)
static ff::Fr n_inverse() {
    static const ff::Fr v = [] {
        ff::Fr nfr = ff::Fr::zero();
        nfr.v[0] = (u32)N;
        return ff::inverse_bgcd(ff::to_mont(nfr));
    }();
    return v;
}

// blobs + evaluation points (device) -> proofs (48 B) + y (canonical limbs), all on `stream`
void prove_enqueue(KzgAmdSettings* dev, size_t off, size_t n, hipStream_t stream, bool evaluate_only = false,
                   int out_mode = kzgamd::OUT_COMPRESSED) {
    // blobs [off, off + n) of the staging buffers
    u32* scal = dev->d_scalars + off * N * 8;
    u32* yv = dev->d_y + off * 8;
    int* stat = dev->d_status + off;
    const u32* bl = (const u32*)(dev->d_blobs + off * BYTES_PER_BLOB);
    const u32* zv = (const u32*)(dev->d_z + off * 8);
    CK_HIP(hipMemsetAsync(stat, 0, n * sizeof(int), stream));
    if (n <= QSPLIT_MAX) {
        // a few blobs: QS workgroups per blob, two phases (k_quotient alone is 0.4 ms of latency per call)
        if (!dev->d_qscratch) CK_HIP(hipMalloc(&dev->d_qscratch, QSPLIT_MAX * QSCR_BYTES));
        hipLaunchKernelGGL(k_quotient_init, dim3(1), dim3(64), 0, stream, dev->d_qscratch, n);
        hipLaunchKernelGGL(k_quotient_a, dim3((unsigned)(n * QS)), dim3(QT), 0, stream, dev->d_qscratch, stat, bl, zv,
                           (const ff::Fr*)dev->d_brp_roots);
        hipLaunchKernelGGL(k_quotient_b, dim3((unsigned)(n * QS)), dim3(QT), 0, stream, scal, yv, dev->d_qscratch, bl, zv,
                           (const ff::Fr*)dev->d_brp_roots, n_inverse());
    } else {
        hipLaunchKernelGGL(k_quotient, dim3((unsigned)n), dim3(QT), 0, stream, scal, yv, stat, bl, zv,
                           (const ff::Fr*)dev->d_brp_roots, n_inverse());
    }
    if (evaluate_only) return;  // y = p(z) is all the caller wants (the field work of batched verification)
    kzgamd::msm_lock(dev->msm);
    try {
        kzgamd::msm_enqueue(dev->msm, dev->d_out + off * (out_mode == kzgamd::OUT_JACOBIAN ? 144 : 48), scal, N, n, 0, stream,
                            out_mode);
    } catch (...) {
        kzgamd::msm_unlock(dev->msm);
        throw;
    }
    kzgamd::msm_unlock(dev->msm);
}

// compute_challenge_rust (kzg/src/eip_4844.rs:920-945) on already-validated inputs: the canonical
// re-serialisation of valid blob elements / a valid commitment equals the input bytes
void challenge_bytes(uint8_t out[32], const uint8_t* blob, const uint8_t commitment[48]) {
    kzgamd::Sha256 h;
    uint8_t head[32] = {0};
    memcpy(head, "FSBLOBVERIFY_V1_", 16);
    const uint64_t nfe = N;
    for (int i = 0; i < 8; ++i) head[24 + 7 - i] = (uint8_t)(nfe >> (8 * i));
    h.update(head, 32);
    h.update(blob, BYTES_PER_BLOB);
    h.update(commitment, 48);
    uint8_t digest[32];
    h.finish(digest);
    // hash_to_bls_field: from_bytes_unchecked = value mod r, re-serialised big-endian
    ff::Fr v;
    for (int i = 0; i < 8; ++i) {
        const uint8_t* q = digest + (7 - i) * 4;
        v.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    ff::Fr red = ff::from_mont(ff::mul(v, ff::Fr::r2()));  // v * R^2 / R / R = v mod r
    for (int i = 0; i < 8; ++i) {
        uint8_t* q = out + (7 - i) * 4;
        q[0] = (uint8_t)(red.v[i] >> 24);
        q[1] = (uint8_t)(red.v[i] >> 16);
        q[2] = (uint8_t)(red.v[i] >> 8);
        q[3] = (uint8_t)red.v[i];
    }
}

void fr_limbs_to_be32(uint8_t out[32], const u32 limbs[8]) {
    for (int i = 0; i < 8; ++i) {
        uint8_t* q = out + (7 - i) * 4;
        q[0] = (uint8_t)(limbs[i] >> 24);
        q[1] = (uint8_t)(limbs[i] >> 16);
        q[2] = (uint8_t)(limbs[i] >> 8);
        q[3] = (uint8_t)limbs[i];
    }
}

}  // namespace
namespace ckz {
bool host_blob_valid(const uint8_t* blob) {
    for (size_t i = 0; i < N; ++i) {
        const uint8_t* e = blob + 32 * i;
        // big-endian compare with r
        static const uint8_t R_BE[32] = {0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8,
                                         0x08, 0x09, 0xa1, 0xd8, 0x05, 0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe,
                                         0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
        if (memcmp(e, R_BE, 32) >= 0) return false;
    }
    return true;
}
}  // namespace ckz
namespace {

}  // namespace
namespace ckz {
// proofs for n (blob, z) pairs; z_src = explicit evaluation points or nullptr to derive them
// from the commitments (compute_blob_kzg_proof)
// proofs == nullptr: evaluation only (zs_out receives the derived challenges)
void prove_batch(KZGProof* proofs, Bytes32* ys, const Blob* blobs, const Bytes32* zs, const Bytes48* commitments, size_t n,
                 KzgAmdSettings* dev, Bytes32* zs_out, bool commitments_checked_elsewhere) {
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    dev->ensure(n);
    std::vector<Bytes32> zbuf;
    std::vector<int> cstat;
    const bool derive = zs == nullptr;
    // The blobs are in pageable memory: the copy call below returns when they are staged (~1.5 ms for 256 blobs).
    // When the challenges have to be derived, the hashing threads are started first and run beside it.
    if (!derive) CK_HIP(hipMemcpyAsync(dev->d_blobs, blobs, n * BYTES_PER_BLOB, hipMemcpyHostToDevice, dev->stream));
    // Commitment validity (decode + subgroup) only decides BadArgs at the end; nothing downstream depends
    // on it.  A few commitments: on the host while the GPU proves (a serial 381-bit chain is ~7x faster on
    // a CPU core than in one GPU lane); a batch: one lane each on a second stream.
    // (the device check is a 1.7 ms latency chain whatever the count; a host core takes ~0.2 ms per commitment with
    // 64-bit limbs, and the hashing pool does them in parallel while the GPU proves: up to 64 blobs the host wins)
    const bool host_check = derive && n <= dev->cfg_host_check_max && !commitments_checked_elsewhere;
    // tuning key device_sha=1: the Fiat-Shamir hashes of a host-buffer batch on the GPU too (k_challenge_sha256, one lane per
    // blob: 2050 serial compressions, ~8 ms however many blobs) and no host threads at all.  Measured against the host
    // pool: 256 blobs 22 k vs 50 k proofs/s, 1024 blobs 47 k vs 66 k, 4096 blobs 72 k vs 78 k — the kernel only pays when
    // other batches hide it (the device-resident pipeline), so the host pool stays the default.
    const bool device_sha = derive && n >= 2 * PROVE_CHUNK && dev->cfg_device_sha && !commitments_checked_elsewhere;
    if (derive) {
        cstat.assign(n, 0);
        if (!host_check && !commitments_checked_elsewhere) {
            CK_HIP(hipMemcpyAsync(dev->d_commit, commitments, n * 48, hipMemcpyHostToDevice, dev->stream2));
            if (device_sha) {
                if (!dev->ev_commit) CK_HIP(hipEventCreateWithFlags(&dev->ev_commit, hipEventDisableTiming));
                CK_HIP(hipEventRecord(dev->ev_commit, dev->stream2));
            }
            CK_HIP(hipMemsetAsync(dev->d_cstatus, 0, n * sizeof(int), dev->stream2));
            // up to 512 commitments: square root and membership test limb-parallel (0.3 + 0.25 ms instead of 1.7 ms of
            // single-lane chain; a wave per point is ~9x the instructions of a lane per point, which a larger batch,
            // whose MSM hides the chain anyway, should not pay)
            if (n <= WIDE_COMMIT_CHECK_MAX && dev->cfg_wide_check) {
                hipLaunchKernelGGL(k_decode_g1_wide, dim3((unsigned)((n + 3) / 4)), dim3(64), 0, dev->stream2, dev->d_cpts,
                                   dev->d_cstatus, (const unsigned char*)dev->d_commit, n);
                hipLaunchKernelGGL(k_affpts_in_g1_wide, dim3((unsigned)n), dim3(64), 0, dev->stream2, dev->d_cstatus,
                                   (const AffPt*)dev->d_cpts, n);
            } else {
                hipLaunchKernelGGL(k_check_commitments, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, dev->stream2,
                                   dev->d_cstatus, (const unsigned char*)dev->d_commit, n);
            }
            // its result is fetched at the very end: a device-to-host copy into pageable memory blocks the calling
            // thread until the kernel is done (1.7 ms for 256 commitments), and nothing below depends on it
        }
        zbuf.resize(n);
        zs = zbuf.data();
    }
    const bool host_compress = proofs && n <= HOST_COMPRESS_MAX;
    const int out_mode = host_compress ? kzgamd::OUT_JACOBIAN : kzgamd::OUT_COMPRESSED;
    unsigned nth = std::thread::hardware_concurrency();
    if (nth == 0) nth = 1;
    // the large-batch pipeline below runs a copier thread and this thread next to the hashing pool: leave them cores
    if (nth > 4 && nth <= 18 && n >= 2 * PROVE_CHUNK) nth -= 2;
    if (nth > 16) nth = 16;
    if (nth > n) nth = (unsigned)n;
    if (device_sha) {
        // a quarter of the call per chunk (64 .. 1024 blobs) on rotating streams: this thread stages chunk k + 1
        // (pageable memory: the copy call returns when the bytes are staged) while the GPU hashes and proves chunk k
        size_t chunk = ((n + 3) / 4 + PROVE_CHUNK - 1) / PROVE_CHUNK * PROVE_CHUNK;
        if (chunk > 1024) chunk = 1024;
        const size_t nchunks = (n + chunk - 1) / chunk;
        for (size_t k = 0; k < nchunks; ++k) {
            const size_t off = k * chunk, cn = off + chunk <= n ? chunk : n - off;
            hipStream_t cs = dev->pipe_stream(k);
            CK_HIP(hipMemcpyAsync(dev->d_blobs + off * BYTES_PER_BLOB, blobs + off, cn * BYTES_PER_BLOB, hipMemcpyHostToDevice, cs));
            if (k < (size_t)KzgAmdSettings::NPIPE) CK_HIP(hipStreamWaitEvent(cs, dev->ev_commit, 0));
            launch_challenge_sha256((u32*)(dev->d_z + off * 8), (const u32*)(dev->d_blobs + off * BYTES_PER_BLOB),
                                    (const u32*)(dev->d_commit + off * 48), cn, dev->cfg_sha_lanes, cs);
            prove_enqueue(dev, off, cn, cs, proofs == nullptr, out_mode);
        }
        dev->pipe_join();  // dev->stream now waits for every chunk
        CK_HIP(hipMemcpyAsync(zbuf.data(), dev->d_z, n * 32, hipMemcpyDeviceToHost, dev->stream));
    } else if (derive && nth > 1 && n >= 2 * PROVE_CHUNK) {
        // blobs per pipeline chunk (median ms per call, chunk 64 / 128 / 256: 256 blobs 6.63 / 6.25 / 6.41, 512 blobs
        // 11.0 / 9.1 / 9.7, 1024 blobs 19.0 / 16.5 / 16.2)
        // (this round, 256 blobs, ms per call by first / later chunk size: 128 / 128 5.0, 64 / 64 4.9, 32 / 64 5.2,
        //  32 / 96 4.75, 48 / 104 4.73 — multiples of 32 blobs fill whole rounds of two waves per SIMD in k_fbw_accum)
        size_t PCH = n >= 1024 ? 4 * PROVE_CHUNK : (n >= 512 ? 2 * PROVE_CHUNK : PROVE_CHUNK + PROVE_CHUNK / 2);
        if (const size_t v = dev->cfg_prove_chunk) {
            if (v >= 16 && v <= 4096) PCH = v;
        }
        // the first chunk is short: nothing runs on the GPU until its blobs are hashed and staged (trace of a 256-blob
        // call with equal chunks of 128: first kernel at 1.17 ms of 5.1)
        size_t first = dev->cfg_prove_first ? dev->cfg_prove_first : (PCH % 64 ? PCH / 3 : PCH / 2);
        if (first > PCH || first < 8) first = PCH;
        // Large batch: a pipeline of PCH-blob chunks on rotating streams.  The pool hashes the blobs in
        // index order, a copier thread stages chunk after chunk (pageable memory: each copy call blocks until the
        // bytes are staged), and this thread enqueues the kernels of a chunk as soon as its challenges and its
        // blobs are there: the GPU proves chunk k while the host is still hashing chunk k+1, and the low-occupancy
        // tails of neighbouring chunks overlap (one MSM workspace per stream).
        std::vector<size_t> coff{0};  // chunk k = blobs [coff[k], coff[k + 1])
        for (size_t at = first < n ? first : n; ; at = at + PCH < n ? at + PCH : n) {
            coff.push_back(at);
            if (at == n) break;
        }
        const size_t nchunks = coff.size() - 1;
        auto chunk_of = [&](size_t i) { return i < first ? (size_t)0 : 1 + (i - first) / PCH; };
        std::vector<char> blob_ok(n, 1);
        std::vector<std::atomic<unsigned>> hashed(nchunks);
        for (auto& h : hashed) h.store(0);
        std::atomic<size_t> copied{0};
        std::atomic<int> copy_err{0};
        if (!dev->pool) dev->pool.reset(new WorkerPool(16));
        std::vector<hipStream_t> cs(nchunks);
        for (size_t k = 0; k < nchunks; ++k) cs[k] = dev->pipe_stream(k);
        std::thread copier([&] {
            (void)hipSetDevice(dev->device);
            for (size_t k = 0; k < nchunks; ++k) {
                const size_t off = coff[k], cn = coff[k + 1] - off;
                if (hipMemcpyAsync(dev->d_blobs + off * BYTES_PER_BLOB, blobs + off, cn * BYTES_PER_BLOB, hipMemcpyHostToDevice,
                                   cs[k]) != hipSuccess)
                    copy_err.store(1);
                copied.store(k + 1, std::memory_order_release);
            }
        });
        std::thread hasher([&] {
            dev->pool->run(nth, [&, nth](unsigned w) {
                for (size_t i = w; i < n; i += nth) {
                    blob_ok[i] = host_blob_valid(blobs[i].bytes) ? 1 : 0;
                    if (blob_ok[i]) challenge_bytes(zbuf[i].bytes, blobs[i].bytes, commitments[i].bytes);
                    hashed[chunk_of(i)].fetch_add(1, std::memory_order_release);
                }
            });
        });
        struct Joiner {
            std::thread &a, &b;
            ~Joiner() {
                if (a.joinable()) a.join();
                if (b.joinable()) b.join();
            }
        } joiner{copier, hasher};
        bool all_ok = true;
        for (size_t k = 0; k < nchunks && all_ok; ++k) {
            const size_t off = coff[k], cn = coff[k + 1] - off;
            while (hashed[k].load(std::memory_order_acquire) < cn || copied.load(std::memory_order_acquire) <= k)
                std::this_thread::yield();
            for (size_t i = off; i < off + cn; ++i) all_ok = all_ok && blob_ok[i];
            if (!all_ok || copy_err.load()) break;
            CK_HIP(hipMemcpyAsync(dev->d_z + off * 8, zs + off, cn * 32, hipMemcpyHostToDevice, cs[k]));
            prove_enqueue(dev, off, cn, cs[k], proofs == nullptr, out_mode);
        }
        copier.join();
        hasher.join();
        dev->pipe_join();  // dev->stream now waits for every chunk
        if (!all_ok || copy_err.load()) {
            (void)hipStreamSynchronize(dev->stream2);
            (void)hipStreamSynchronize(dev->stream);
            if (copy_err.load()) throw CkErr{C_KZG_ERROR, "host-to-device copy failed"};
            throw CkErr{C_KZG_BADARGS, "Invalid scalar"};
        }
    } else {
        if (derive) {
            std::vector<char> blob_ok(n, 1);
            auto work = [&, nth](unsigned w) {
                for (size_t i = w; i < n; i += nth) {
                    blob_ok[i] = host_blob_valid(blobs[i].bytes) ? 1 : 0;
                    if (blob_ok[i]) challenge_bytes(zbuf[i].bytes, blobs[i].bytes, commitments[i].bytes);
                }
            };
            if (nth == 1) {
                CK_HIP(hipMemcpyAsync(dev->d_blobs, blobs, n * BYTES_PER_BLOB, hipMemcpyHostToDevice, dev->stream));
                work(0);
            } else {
                // hash on the pool while this thread stages the blobs (pageable memory: the copy call returns when
                // the bytes are staged)
                if (!dev->pool) dev->pool.reset(new WorkerPool(16));
                hipError_t ce = hipSuccess;
                std::thread copier([&] {
                    (void)hipSetDevice(dev->device);
                    ce = hipMemcpyAsync(dev->d_blobs, blobs, n * BYTES_PER_BLOB, hipMemcpyHostToDevice, dev->stream);
                });
                struct Joiner {  // an exception out of the pool must not leave a joinable thread behind (std::terminate)
                    std::thread& t;
                    ~Joiner() {
                        if (t.joinable()) t.join();
                    }
                } joiner{copier};
                dev->pool->run(nth, work);
                copier.join();
                CK_HIP(ce);
            }
            for (size_t i = 0; i < n; ++i)
                if (!blob_ok[i]) {
                    (void)hipStreamSynchronize(dev->stream2);
                    (void)hipStreamSynchronize(dev->stream);
                    throw CkErr{C_KZG_BADARGS, "Invalid scalar"};
                }
        }
        CK_HIP(hipMemcpyAsync(dev->d_z, zs, n * 32, hipMemcpyHostToDevice, dev->stream));
        prove_enqueue(dev, 0, n, dev->stream, proofs == nullptr, out_mode);
    }
    // commitment check on the host, while the GPU works (the copies below block until it is done)
    if (host_check) {
        auto check = [&](size_t i) {
            blst_p1 c;
            if (!kzgamd::host_p1_uncompress(&c, commitments[i].bytes) || !kzgamd::host_p1_in_g1(&c)) cstat[i] = 1;
        };
        if (n > 1 && nth > 1) {
            if (!dev->pool) dev->pool.reset(new WorkerPool(16));
            dev->pool->run(nth, [&, nth](unsigned w) {
                for (size_t i = w; i < n; i += nth) check(i);
            });
        } else {
            for (size_t i = 0; i < n; ++i) check(i);
        }
    }
    std::vector<int> status(n);
    std::vector<u32> ylimbs(n * 8);
    blst_p1 jac[HOST_COMPRESS_MAX];
    CK_HIP(hipMemcpyAsync(status.data(), dev->d_status, n * sizeof(int), hipMemcpyDeviceToHost, dev->stream));
    CK_HIP(hipMemcpyAsync(ylimbs.data(), dev->d_y, n * 32, hipMemcpyDeviceToHost, dev->stream));
    if (host_compress) CK_HIP(hipMemcpyAsync(jac, dev->d_out, n * 144, hipMemcpyDeviceToHost, dev->stream));
    else if (proofs) CK_HIP(hipMemcpyAsync(proofs, dev->d_out, n * 48, hipMemcpyDeviceToHost, dev->stream));
    CK_HIP(hipStreamSynchronize(dev->stream));
    if (zs_out) memcpy(zs_out, zs, n * 32);
    if (derive) {
        if (!host_check && !commitments_checked_elsewhere) {
            CK_HIP(hipMemcpyAsync(cstat.data(), dev->d_cstatus, n * sizeof(int), hipMemcpyDeviceToHost, dev->stream2));
            CK_HIP(hipStreamSynchronize(dev->stream2));
        }
        for (size_t i = 0; i < n; ++i) CK_REQUIRE(cstat[i] == 0, "Invalid commitment");
    }
    for (size_t i = 0; i < n; ++i) CK_REQUIRE(status[i] == 0, "Invalid scalar");
    if (host_compress) compress_on_host(proofs[0].bytes, jac, n);
    if (ys)
        for (size_t i = 0; i < n; ++i) fr_limbs_to_be32(ys[i].bytes, &ylimbs[8 * i]);
}
}  // namespace ckz
namespace {


}  // namespace

KzgAmdSettings* kzgamd::device_settings(const CKZGSettings* s) { return lookup(s); }

// ---------------------------------------------------------------- C ABI (B3)

extern "C" C_KZG_RET kzgamd_load_trusted_setup_ex(CKZGSettings* out, const uint8_t* g1_monomial_bytes, uint64_t num_g1_monomial_bytes,
                                                  const uint8_t* g1_lagrange_bytes, uint64_t num_g1_lagrange_bytes,
                                                  const uint8_t* g2_monomial_bytes, uint64_t num_g2_monomial_bytes,
                                                  uint64_t precompute, const KzgAmdConfig* cfg) {
    (void)precompute;
    if (!out) return C_KZG_BADARGS;
    zero_settings(out);
    if (!g1_monomial_bytes || !g1_lagrange_bytes || !g2_monomial_bytes) return C_KZG_BADARGS;
    return guarded([&] {
        load_impl(out, g1_monomial_bytes, num_g1_monomial_bytes, g1_lagrange_bytes, num_g1_lagrange_bytes, g2_monomial_bytes,
                  num_g2_monomial_bytes, cfg);
    });
}
extern "C" C_KZG_RET load_trusted_setup(CKZGSettings* out, const uint8_t* g1_monomial_bytes, uint64_t num_g1_monomial_bytes,
                                        const uint8_t* g1_lagrange_bytes, uint64_t num_g1_lagrange_bytes,
                                        const uint8_t* g2_monomial_bytes, uint64_t num_g2_monomial_bytes,
                                        uint64_t precompute) {
    return kzgamd_load_trusted_setup_ex(out, g1_monomial_bytes, num_g1_monomial_bytes, g1_lagrange_bytes, num_g1_lagrange_bytes,
                                        g2_monomial_bytes, num_g2_monomial_bytes, precompute, nullptr);
}

extern "C" C_KZG_RET load_trusted_setup_file(CKZGSettings* out, FILE* in) { return kzgamd_load_trusted_setup_file_ex(out, in, nullptr); }

extern "C" C_KZG_RET kzgamd_load_trusted_setup_file_ex(CKZGSettings* out, FILE* in, const KzgAmdConfig* cfg) {
    if (!out) return C_KZG_BADARGS;
    zero_settings(out);
    if (!in) return C_KZG_BADARGS;
    return guarded([&] {
        std::string buf(1024 * 1024, '\0');  // the reference reads at most 1 MiB (blst/src/eip_4844.rs:244-246)
        size_t len = fread(&buf[0], 1, buf.size(), in);
        buf.resize(len);
        std::vector<uint8_t> g1m, g1l, g2m;
        parse_setup_text(buf, g1m, g1l, g2m);
        load_impl(out, g1m.data(), g1m.size(), g1l.data(), g1l.size(), g2m.data(), g2m.size(), cfg);
    });
}

extern "C" void free_trusted_setup(CKZGSettings* s) {
    if (!s) return;
    KzgAmdSettings* dev = nullptr;
    if (s->g1_values_lagrange_brp) {
        std::lock_guard<std::mutex> lk(g_registry_mu);
        auto it = g_registry.find(s->g1_values_lagrange_brp);
        if (it != g_registry.end()) {
            dev = it->second;
            g_registry.erase(it);
        }
    }
    if (dev) {
        kzgamd::DeviceGuard on_device(dev->device);
        delete dev;
    }
    free_host_arrays(s);
}

namespace {
// See KzgAmdSettings::CoalesceQueue.  Req has `bool done`, `C_KZG_RET rc` and `void side_work()`; run(batch) serves
// every request of the batch (sets rc).  A caller returns as soon as its own request is served; leadership passes to
// whoever is waiting.  side_work() is the part of a request that needs no GPU (the host-side commitment check of a blob
// proof): a waiting caller does it before it sleeps, a leader between launching a batch and waiting for it
// (run's implementation calls it), and whoever has not got to it by the end does it then.
template <class Req, class Run>
C_KZG_RET coalesced_call(KzgAmdSettings::CoalesceQueue& q, Req& me, Run&& run) {
    const size_t gather_min = q.gather_min;
    const int gather_us = q.gather_us;
    // everything that can allocate happens before the request is visible to other callers: nothing below throws
    std::vector<Req*> batch;
    std::unique_lock<std::mutex> lk(q.mu, std::defer_lock);
    try {
        batch.reserve(KzgAmdSettings::LANE_MAX_BLOBS);
        lk.lock();
        q.pending.push_back(&me);
    } catch (...) {
        return C_KZG_MALLOC;
    }
    q.cv.notify_one();  // a leader gathering requests may have enough now
    bool idled = false;
    while (!me.done) {
        if (q.leaders < q.max_leaders && !q.pending.empty()) {
            ++q.leaders;
            while (!q.pending.empty() && !me.done) {
                // under load (other batches in flight) a short wait lets the callers that have just been served come
                // back with their next request: larger batches, fewer pipeline invocations
                if (q.leaders > 1 && q.pending.size() < gather_min) {
                    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(gather_us);
                    while (q.pending.size() < gather_min && !me.done &&
                           q.cv.wait_until(lk, until) != std::cv_status::timeout) {
                    }
                    if (me.done) break;
                    if (q.pending.empty()) continue;
                }
                batch.clear();
                while (!q.pending.empty() && batch.size() < KzgAmdSettings::LANE_MAX_BLOBS) {
                    batch.push_back(static_cast<Req*>(q.pending.front()));
                    q.pending.pop_front();
                }
                lk.unlock();
                // a failure of the batch as a whole (a HIP error, no memory) fails every request of it, with the same
                // mapping as every other entry point (guarded: BadArgs like the reference, or Malloc); run() has waited
                // for the lane's stream before it throws, so the callers' staged blobs are no longer being read
                const C_KZG_RET brc = guarded([&] { run(batch); });
                if (brc != C_KZG_OK)
                    for (Req* r : batch) r->rc = brc;
                lk.lock();
                for (Req* r : batch) r->done = true;
                q.cv.notify_all();
            }
            --q.leaders;
            q.cv.notify_all();
        } else if (!idled) {
            idled = true;
            lk.unlock();
            me.side_work();
            lk.lock();
        } else {
            q.cv.wait(lk);
        }
    }
    lk.unlock();
    me.side_work();
    return me.rc;
}

struct CommitReq {
    const Blob* blob;
    KZGCommitment* out;
    const unsigned char* staged = nullptr;  // the caller's page-locked copy of the blob (PinnedSlots), if it got a slot
    bool done = false;
    C_KZG_RET rc = C_KZG_ERROR;
    void side_work() {}
};

// releases a caller's slot on every way out of its entry point
struct SlotHold {
    KzgAmdSettings* dev;
    unsigned char* p;
    SlotHold(KzgAmdSettings* d, const void* blob) : dev(d), p(d->slots.acquire(d->device)) {
        if (p) memcpy(p, blob, BYTES_PER_BLOB);
    }
    ~SlotHold() { dev->slots.release(p); }
    SlotHold(const SlotHold&) = delete;
    SlotHold& operator=(const SlotHold&) = delete;
};

// up to LANE_MAX_BLOBS commitments on one lane, through its page-locked staging: every copy asynchronous, one wait at
// the end; a blob with an element >= r fails its own request only (per-blob status of k_blob_to_scalars)
void commit_lane_batch(KzgAmdSettings* dev, const std::vector<CommitReq*>& reqs) {
    const size_t n = reqs.size();
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    dev->ensure(KzgAmdSettings::LANE_MAX_BLOBS);
    dev->ensure_pinned();
    const bool host_compress = n <= HOST_COMPRESS_MAX;
    BlobPtrs ptrs;
    for (size_t i = 0; i < KzgAmdSettings::LANE_MAX_BLOBS; ++i) ptrs.p[i] = nullptr;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char* src = reqs[i]->staged;
        if (!src) {  // no slot was free: this thread stages the blob
            memcpy(dev->h_in + i * BYTES_PER_BLOB, reqs[i]->blob, BYTES_PER_BLOB);
            src = dev->h_in + i * BYTES_PER_BLOB;
        }
        ptrs.p[i] = reinterpret_cast<const u32*>(src);
    }
    int* hs = reinterpret_cast<int*>(dev->h_res);
    unsigned char* ho = dev->h_res + KzgAmdSettings::LANE_MAX_BLOBS * sizeof(int);
    auto enqueue_all = [&] {
        commit_enqueue(dev, dev->d_out, dev->d_status, nullptr, dev->d_scalars, n, dev->stream,
                       host_compress ? kzgamd::OUT_JACOBIAN : kzgamd::OUT_COMPRESSED, &ptrs);
        CK_HIP(hipMemcpyAsync(hs, dev->d_status, n * sizeof(int), hipMemcpyDeviceToHost, dev->stream));
        CK_HIP(hipMemcpyAsync(ho, dev->d_out, n * (host_compress ? 144 : 48), hipMemcpyDeviceToHost, dev->stream));
    };
    // (Replaying this sequence as one captured graph per batch size was measured: 19.4 k vs 18.9 k commitments/s at 16
    // threads, 3 991 vs 4 040 /s for one — nothing; the call-by-call form stays.  profiles/NOTES.md §9.)
    try {
        enqueue_all();
    } catch (...) {
        (void)hipStreamSynchronize(dev->stream);  // whatever was enqueued still reads the callers' slots
        throw;
    }
    CK_HIP(hipStreamSynchronize(dev->stream));
    for (size_t i = 0; i < n; ++i) {
        if (hs[i] != 0) {
            reqs[i]->rc = C_KZG_BADARGS;  // "Invalid scalar"
            continue;
        }
        if (!host_compress) memcpy(reqs[i]->out->bytes, ho + 48 * i, 48);
        reqs[i]->rc = C_KZG_OK;
    }
    if (host_compress) {
        // one inversion for the whole batch; a failed request's slot holds whatever the MSM made of its zeroed scalars
        uint8_t cb[KzgAmdSettings::LANE_MAX_BLOBS * 48];
        compress_on_host(cb, reinterpret_cast<const blst_p1*>(ho), n);
        for (size_t i = 0; i < n; ++i)
            if (reqs[i]->rc == C_KZG_OK) memcpy(reqs[i]->out->bytes, cb + 48 * i, 48);
    }
}

}  // namespace

extern "C" C_KZG_RET kzgamd_blob_to_kzg_commitment_batch(KZGCommitment* out, const Blob* blobs, size_t n,
                                                         const CKZGSettings* s) {
    if (!out || !blobs) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    if (n <= KzgAmdSettings::LANE_MAX_BLOBS)
        return guarded([&] {
            // a few blobs: an idle lane of the settings object (concurrent callers overlap on the GPU)
            std::vector<CommitReq> reqs(n);
            std::vector<CommitReq*> ptrs(n);
            for (size_t i = 0; i < n; ++i) {
                reqs[i].blob = blobs + i;
                reqs[i].out = out + i;
                ptrs[i] = &reqs[i];
            }
            LaneRef lane(dev, n);
            commit_lane_batch(lane.use, ptrs);
            for (size_t i = 0; i < n; ++i) CK_REQUIRE(reqs[i].rc == C_KZG_OK, "Invalid scalar");
        });
    return guarded([&, root = dev] {
        LaneRef lane(root, n);
        KzgAmdSettings* dev = lane.use;
        std::lock_guard<std::mutex> lk(dev->mu);
        kzgamd::DeviceGuard on_device(dev->device);
        CK_HIP(on_device.err);
        dev->ensure(n);
        const bool host_compress = n <= HOST_COMPRESS_MAX;
        if (n >= 2 * COMMIT_CHUNK) {
            // Large batch: chunks on rotating streams.  The blobs are in pageable memory, so a copy call returns when
            // its chunk is staged — while this thread stages chunk k + 1 the GPU commits to chunk k, and the tails of
            // neighbouring chunks overlap (one MSM workspace per stream).  PCIe and compute run concurrently instead
            // of back to back.
            // Chunk sizes: nothing runs until the first chunk is copied, so it is short (an eighth of the call, 32 .. 256
            // blobs); the others take a third of the rest each, at most 1024 blobs (where the MSM kernels run at their
            // full rate), in multiples of 32 blobs (whole rounds of two waves per SIMD in k_fbw_accum).  Small kernels
            // pay the per-launch fold and latency tails again, so there are few of them.
            // (256 blobs: one chunk 3.99 ms, 4 x 64 4.08, 64 + 192 see DESIGN.md §6)
            auto round32 = [](size_t v) { return (v + 31) / 32 * 32; };
            size_t first = round32(n / 8);
            first = first < 32 ? 32 : (first > 256 ? 256 : first);
            if (const size_t v = dev->cfg_commit_first) first = v;
            size_t chunk = round32((n - first + 2) / 3);
            chunk = chunk < COMMIT_CHUNK ? COMMIT_CHUNK : (chunk > 1024 ? 1024 : chunk);
            if (const size_t v = dev->cfg_commit_chunk) chunk = v;
            if (first > n) first = n;
            size_t off = 0;
            for (size_t k = 0; off < n; ++k) {
                const size_t cn = k == 0 ? first : (off + chunk <= n ? chunk : n - off);
                hipStream_t cs = dev->pipe_stream(k);
                CK_HIP(hipMemcpyAsync(dev->d_blobs + off * BYTES_PER_BLOB, blobs + off, cn * BYTES_PER_BLOB, hipMemcpyHostToDevice, cs));
                commit_enqueue(dev, dev->d_out + off * 48, dev->d_status + off, dev->d_blobs + off * BYTES_PER_BLOB,
                               dev->d_scalars + off * N * 8, cn, cs, kzgamd::OUT_COMPRESSED);
                off += cn;
            }
            dev->pipe_join();
        } else {
            CK_HIP(hipMemcpyAsync(dev->d_blobs, blobs, n * BYTES_PER_BLOB, hipMemcpyHostToDevice, dev->stream));
            commit_enqueue(dev, dev->d_out, dev->d_status, dev->d_blobs, dev->d_scalars, n, dev->stream,
                           host_compress ? kzgamd::OUT_JACOBIAN : kzgamd::OUT_COMPRESSED);
        }
        std::vector<int> status(n);
        blst_p1 jac[HOST_COMPRESS_MAX];
        CK_HIP(hipMemcpyAsync(status.data(), dev->d_status, n * sizeof(int), hipMemcpyDeviceToHost, dev->stream));
        if (host_compress) CK_HIP(hipMemcpyAsync(jac, dev->d_out, n * 144, hipMemcpyDeviceToHost, dev->stream));
        else CK_HIP(hipMemcpyAsync(out, dev->d_out, n * 48, hipMemcpyDeviceToHost, dev->stream));
        CK_HIP(hipStreamSynchronize(dev->stream));
        for (size_t i = 0; i < n; ++i) CK_REQUIRE(status[i] == 0, "Invalid scalar");
        if (host_compress) compress_on_host(out[0].bytes, jac, n);
    });
}

// blst/src/eip_4844.rs:163-175.  Concurrent callers on one settings object are merged into batches (coalesced_call).
extern "C" C_KZG_RET blob_to_kzg_commitment(KZGCommitment* out, const Blob* blob, const CKZGSettings* s) {
    if (!out || !blob) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    CommitReq me;
    me.blob = blob;
    me.out = out;
    SlotHold slot(dev, blob);
    me.staged = slot.p;
    return coalesced_call(dev->q_commit, me, [&](const std::vector<CommitReq*>& batch) {
        LaneRef lane(dev, batch.size());
        commit_lane_batch(lane.use, batch);
    });
}

extern "C" C_KZG_RET kzgamd_blob_to_kzg_commitment_device(void* d_out, void* d_status, void* d_scratch, const void* d_blobs,
                                                          size_t n, const CKZGSettings* s, void* stream) {
    KzgAmdSettings* dev = lookup(s);
    if (!dev || !d_out || !d_status || !d_scratch || !d_blobs) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    return guarded([&] {
        kzgamd::DeviceGuard on_device(dev->device);
        CK_HIP(on_device.err);
        commit_enqueue(dev, d_out, (int*)d_status, d_blobs, (u32*)d_scratch, n, (hipStream_t)stream);
    });
}

// Pre-allocates what kzgamd_blob_to_kzg_commitment_device needs for batches of up to n blobs on `stream` (the MSM
// workspace of that stream), so that the enqueue calls never allocate — hipMalloc synchronises the device.
extern "C" C_KZG_RET kzgamd_settings_reserve(const CKZGSettings* s, size_t n, void* stream) {
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    return guarded([&] {
        kzgamd::DeviceGuard on_device(dev->device);
        CK_HIP(on_device.err);
        kzgamd::msm_lock(dev->msm);
        try {
            kzgamd::msm_enqueue(dev->msm, nullptr, nullptr, N, n, 0, (hipStream_t)stream, kzgamd::OUT_COMPRESSED, true);
        } catch (...) {
            kzgamd::msm_unlock(dev->msm);
            throw;
        }
        kzgamd::msm_unlock(dev->msm);
    });
}

extern "C" int kzgamd_settings_device(const CKZGSettings* s) {
    KzgAmdSettings* dev = lookup(s);
    return dev ? dev->device : -1;
}


// Device-resident compute_blob_kzg_proof for n blobs, enqueued on `stream` without synchronising: challenge (SHA-256
// on the device), commitment validation, barycentric evaluation + quotient, fixed-base MSM, compression.
// d_scratch: n x KZGAMD_PROOF_SCRATCH_BYTES.  d_status[i] != 0: blob i has an element >= r or commitment i is not a
// valid G1 element (the reference returns BadArgs for the call; here the other proofs of the batch are still valid).
extern "C" C_KZG_RET kzgamd_compute_blob_kzg_proof_device(void* d_proofs, void* d_status, void* d_scratch, const void* d_blobs,
                                                          const void* d_commitments, size_t n, const CKZGSettings* s,
                                                          void* stream) {
    KzgAmdSettings* dev = lookup(s);
    if (!dev || !d_proofs || !d_status || !d_scratch || !d_blobs || !d_commitments) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    return guarded([&] {
        kzgamd::DeviceGuard on_device(dev->device);
        CK_HIP(on_device.err);
        hipStream_t st = (hipStream_t)stream;
        u32* scal = (u32*)d_scratch;
        u32* z = scal + n * N * 8;
        u32* y = z + n * 8;
        int* stat = (int*)d_status;
        CK_HIP(hipMemsetAsync(stat, 0, n * sizeof(int), st));
        launch_challenge_sha256(z, (const u32*)d_blobs, (const u32*)d_commitments, n, dev->cfg_sha_lanes, st);
        hipLaunchKernelGGL(k_check_commitments, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, stat,
                           (const unsigned char*)d_commitments, n);
        hipLaunchKernelGGL(k_quotient, dim3((unsigned)n), dim3(QT), 0, st, scal, y, stat, (const u32*)d_blobs, (const u32*)z,
                           (const ff::Fr*)dev->d_brp_roots, n_inverse());
        kzgamd::msm_lock(dev->msm);
        try {
            kzgamd::msm_enqueue(dev->msm, d_proofs, scal, N, n, 0, st, kzgamd::OUT_COMPRESSED);
        } catch (...) {
            kzgamd::msm_unlock(dev->msm);
            throw;
        }
        kzgamd::msm_unlock(dev->msm);
        CK_HIP(hipGetLastError());
    });
}

namespace {
struct ProofReq {
    const Blob* blob;
    const Bytes48* commitment;  // compute_blob_kzg_proof
    const Bytes32* z;           // compute_kzg_proof; compute_blob_kzg_proof: the derived challenge (zder)
    KZGProof* proof;
    Bytes32* y;
    const unsigned char* staged = nullptr;  // the caller's page-locked copy of the blob, if it got a slot
    Bytes32 zder;
    bool commitment_ok = true, checked = false;
    bool done = false;
    C_KZG_RET rc = C_KZG_ERROR;
    // validate_batched_input's half for the commitment (decode + subgroup, ~0.2 ms of one core): never on the GPU's
    // critical path
    void side_work() {
        if (checked || !commitment) return;
        checked = true;
        blst_p1 c;
        commitment_ok = kzgamd::host_p1_uncompress(&c, commitment->bytes) && kzgamd::host_p1_in_g1(&c);
    }
};

// A batch of merged single-proof calls on one lane.  Every caller has staged its blob (page-locked slot), validated it
// and derived its challenge on its own thread; here: gather the blobs on the device, quotient + MSM, results back
// through the lane's page-locked buffer, one wait.  Status is per request (k_quotient flags a blob or an evaluation
// point that is not canonical): an invalid request fails alone.
void proof_lane_batch(KzgAmdSettings* root, const std::vector<ProofReq*>& reqs, ProofReq* leader) {
    const size_t n = reqs.size();
    LaneRef lane(root, n);
    KzgAmdSettings* dev = lane.use;
    const C_KZG_RET rc = guarded([&] {
        std::lock_guard<std::mutex> lk(dev->mu);
        kzgamd::DeviceGuard on_device(dev->device);
        CK_HIP(on_device.err);
        dev->ensure(KzgAmdSettings::LANE_MAX_BLOBS);
        dev->ensure_pinned();
        constexpr size_t LB = KzgAmdSettings::LANE_MAX_BLOBS;
        // lane result buffer: status | Jacobian proofs | y | z (in)
        int* hs = reinterpret_cast<int*>(dev->h_res);
        unsigned char* ho = dev->h_res + LB * sizeof(int);
        u32* hy = reinterpret_cast<u32*>(ho + LB * 144);
        unsigned char* hz = reinterpret_cast<unsigned char*>(hy) + LB * 32;
        BlobPtrs ptrs;
        for (size_t i = 0; i < LB; ++i) ptrs.p[i] = nullptr;
        for (size_t i = 0; i < n; ++i) {
            const unsigned char* src = reqs[i]->staged;
            if (!src) {
                memcpy(dev->h_in + i * BYTES_PER_BLOB, reqs[i]->blob, BYTES_PER_BLOB);
                src = dev->h_in + i * BYTES_PER_BLOB;
            }
            ptrs.p[i] = reinterpret_cast<const u32*>(src);
            memcpy(hz + 32 * i, reqs[i]->z->bytes, 32);
        }
        try {
            hipLaunchKernelGGL(k_gather_blobs, dim3((unsigned)((n * (BYTES_PER_BLOB / 16) + 255) / 256)), dim3(256), 0, dev->stream,
                               reinterpret_cast<uint4*>(dev->d_blobs), ptrs, n);
            CK_HIP(hipMemcpyAsync(dev->d_z, hz, n * 32, hipMemcpyHostToDevice, dev->stream));
            prove_enqueue(dev, 0, n, dev->stream, false, kzgamd::OUT_JACOBIAN);
            CK_HIP(hipMemcpyAsync(hs, dev->d_status, n * sizeof(int), hipMemcpyDeviceToHost, dev->stream));
            CK_HIP(hipMemcpyAsync(hy, dev->d_y, n * 32, hipMemcpyDeviceToHost, dev->stream));
            CK_HIP(hipMemcpyAsync(ho, dev->d_out, n * 144, hipMemcpyDeviceToHost, dev->stream));
        } catch (...) {
            (void)hipStreamSynchronize(dev->stream);  // whatever was enqueued still reads the callers' slots
            throw;
        }
        if (leader) leader->side_work();  // while the GPU works
        CK_HIP(hipStreamSynchronize(dev->stream));
        uint8_t cb[LB * 48];
        compress_on_host(cb, reinterpret_cast<const blst_p1*>(ho), n);
        for (size_t i = 0; i < n; ++i) {
            if (hs[i] != 0) {
                reqs[i]->rc = C_KZG_BADARGS;  // "Invalid scalar"
                continue;
            }
            memcpy(reqs[i]->proof->bytes, cb + 48 * i, 48);
            if (reqs[i]->y) fr_limbs_to_be32(reqs[i]->y->bytes, hy + 8 * i);
            reqs[i]->rc = C_KZG_OK;
        }
    });
    if (rc != C_KZG_OK)
        for (ProofReq* r : reqs) r->rc = rc;
}
}  // namespace

extern "C" C_KZG_RET compute_kzg_proof(KZGProof* proof_out, Bytes32* y_out, const Blob* blob, const Bytes32* z_bytes,
                                       const CKZGSettings* s) {
    if (!proof_out || !y_out || !blob || !z_bytes) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    ProofReq me{blob, nullptr, z_bytes, proof_out, y_out};
    SlotHold slot(dev, blob);
    me.staged = slot.p;
    return coalesced_call(dev->q_proof, me, [&](const std::vector<ProofReq*>& batch) { proof_lane_batch(dev, batch, &me); });
}

extern "C" C_KZG_RET kzgamd_compute_blob_kzg_proof_batch(KZGProof* out, const Blob* blobs, const Bytes48* commitments,
                                                         size_t n, const CKZGSettings* s) {
    if (!out || !blobs || !commitments) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    return guarded([&] {
        LaneRef lane(dev, n);
        prove_batch(out, nullptr, blobs, nullptr, commitments, n, lane.use);
    });
}

// compute_challenges_and_evaluate_polynomial (kzg/src/eip_4844.rs:690-719): the per-blob field work of
// verify_blob_kzg_proof_batch — Fiat-Shamir challenge z_i (host SHA-256) and y_i = p_i(z_i) (barycentric, on the
// GPU).  The pairing side (verify_kzg_proof_batch, :380-435) stays with the caller.
extern "C" C_KZG_RET kzgamd_compute_challenges_and_evaluate_batch(Bytes32* zs_out, Bytes32* ys_out, const Blob* blobs,
                                                                  const Bytes48* commitments, size_t n,
                                                                  const CKZGSettings* s) {
    if (!zs_out || !ys_out || !blobs || !commitments) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    return guarded([&] {
        LaneRef lane(dev, n);
        prove_batch(nullptr, ys_out, blobs, nullptr, commitments, n, lane.use, zs_out);
    });
}

// blst/src/eip_4844.rs:498-517.  Concurrent callers on one settings object are merged into batches (coalesced_call).
extern "C" C_KZG_RET compute_blob_kzg_proof(KZGProof* out, const Blob* blob, const Bytes48* commitment_bytes,
                                            const CKZGSettings* s) {
    if (!out || !blob || !commitment_bytes) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    KZGProof proof;  // *out is written only when the call succeeds (an invalid commitment is found after the proof)
    ProofReq me{blob, commitment_bytes, nullptr, &proof, nullptr};
    // this thread's share, in parallel with the other callers': the page-locked copy, blob_to_polynomial's range check
    // and the Fiat-Shamir challenge (one SHA-256 over the blob)
    SlotHold slot(dev, blob);
    me.staged = slot.p;
    if (!host_blob_valid(blob->bytes)) return C_KZG_BADARGS;  // "Invalid scalar"
    challenge_bytes(me.zder.bytes, blob->bytes, commitment_bytes->bytes);
    me.z = &me.zder;
    const C_KZG_RET rc = coalesced_call(dev->q_blob_proof, me,
                                        [&](const std::vector<ProofReq*>& batch) { proof_lane_batch(dev, batch, &me); });
    if (rc == C_KZG_OK && !me.commitment_ok) return C_KZG_BADARGS;  // "Invalid commitment"
    if (rc == C_KZG_OK) *out = proof;
    return rc;
}

// The reference exports this helper with raw blst types (blst/src/eip_4844.rs:501-514): the commitment
// is a blst_p1, the result a Montgomery blst_fr; inputs are trusted (the reference unwraps).
extern "C" void compute_challenge(blst_fr* eval_challenge_out, const Blob* blob, const blst_p1* commitment) {
    uint8_t cbytes[48], zbe[32];
    kzgamd::host_p1_compress(cbytes, commitment);
    challenge_bytes(zbe, blob->bytes, cbytes);
    ff::Fr v;
    for (int i = 0; i < 8; ++i) {
        const uint8_t* q = zbe + (7 - i) * 4;
        v.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    v = ff::to_mont(v);
    memcpy(eval_challenge_out, &v, 32);
}

extern "C" C_KZG_RET bytes_to_kzg_commitment(blst_p1* out, const Bytes48* b) {   /* blst/src/eip_4844.rs:519-523 */
    if (!out || !b) return C_KZG_BADARGS;
    return kzgamd::host_p1_uncompress(out, b->bytes) ? C_KZG_OK : C_KZG_BADARGS;
}

extern "C" void bytes_from_bls_field(Bytes32* out, const blst_fr* in) {          /* blst/src/eip_4844.rs:528-530 */
    ff::Fr v;
    memcpy(&v, in, 32);
    v = ff::from_mont(v);
    fr_limbs_to_be32(out->bytes, v.v);
}

extern "C" int kzgamd_settings_table_info(const CKZGSettings* s, int which, int* window_bits, int* rows, int* wide_table) {
    KzgAmdSettings* dev = lookup(s);
    if (!dev || which < 0 || which > 2) return -1;
    kzgamd::MsmContext* h = which == 0 ? dev->msm : which == 1 ? dev->msm_monomial : dev->msm_xext;
    if (!h) {
        if (wide_table) *wide_table = 0;
        return 1;
    }
    size_t nb = 0, np = 0;
    int c = 0, r = 0;
    kzgamd_msm_info(h, &c, &r, &nb, &np);
    if (window_bits) *window_bits = c;
    if (rows) *rows = r;
    if (wide_table) *wide_table = kzgamd_msm_uses_wide_table(h);
    return 0;
}

extern "C" void* kzgamd_settings_msm_handle(const CKZGSettings* s) {
    KzgAmdSettings* dev = lookup(s);
    return dev ? (void*)dev->msm : nullptr;
}

