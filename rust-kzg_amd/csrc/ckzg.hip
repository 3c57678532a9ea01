// c-kzg-4844 surface (B3) for the proving path, MI355X-native.
// Replaces blst/src/eip_4844.rs:160-530 + the generic protocol code it calls
// (kzg/src/eip_4844.rs).  The CKZGSettings struct has the reference's layout
// (kzg/src/eth/c_bindings.rs:55-108); the device-resident state (fixed-base MSM table,
// roots) hangs off a registry keyed by the settings' g1_values_lagrange_brp pointer —
// the same trick the reference uses for its precomputation tables
// (PrecomputationTableManager, kzg/src/eip_4844.rs:64-146).
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "ckzg_internal.h"
#include "config.h"
#include "device_guard.h"
#include "ff.hip.h"
#include "fr29.hip.h"
#include "g1_io.hip.h"
#include "g1w.hip.h"
#include "host_g1.h"
#include "host_pairing.h"
#include "msm_internal.h"
#include "ntt_internal.h"
#include "sha256.h"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <thread>

using ff::u32;
using ff::u64;
using g1::AffPt;

namespace {

struct CkErr {
    C_KZG_RET rc;
    std::string what;
};
#define CK_HIP(x)                                                                         \
    do {                                                                                  \
        hipError_t _e = (x);                                                              \
        if (_e != hipSuccess) throw CkErr{C_KZG_ERROR, std::string(#x) + ": " + hipGetErrorString(_e)}; \
    } while (0)
#define CK_REQUIRE(cond, msg)                      \
    do {                                           \
        if (!(cond)) throw CkErr{C_KZG_BADARGS, msg}; \
    } while (0)

constexpr size_t N = FIELD_ELEMENTS_PER_BLOB;
constexpr size_t NUM_G2 = 65;
constexpr size_t CELL_SIZE = 64;                 // FIELD_ELEMENTS_PER_CELL
constexpr size_t CELLS_PER_BLOB = N / CELL_SIZE;  // 64; the extended blob has 128 cells

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

// ---- Fr helpers for the proving kernel (Montgomery, 8 x u32) ----
__device__ __forceinline__ ff::Fr fr_load_be(const u32* __restrict__ w8, bool* ok) {
    // 32 big-endian bytes -> canonical limbs; *ok = value < r
    ff::Fr a;
#pragma unroll
    for (int k = 0; k < 8; ++k) a.v[k] = __builtin_bswap32(w8[7 - k]);
    u64 borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        u64 d = (u64)a.v[k] - ff::FrParams::p(k) - borrow;
        borrow = (d >> 32) & 1;
    }
    *ok = borrow != 0;
    return a;
}
// Montgomery inverse by binary Euclid (ff.hip.h); 0 -> 0 like blst_fr_eucl_inverse
__device__ ff::Fr fr_inverse(const ff::Fr& a) { return ff::inverse_bgcd(a); }
// ff::mul on blst_fr values through the 29-bit multiplier of the NTT (fr29::mul_blst: the same result in about half
// the instructions); the quotient kernels below are a stream of such products
__device__ __forceinline__ ff::Fr fmul(const ff::Fr& a, const ff::Fr& b) { return fr29::mul_blst(a, b); }

// Host worker threads for the per-blob SHA-256 challenges of a batch, kept alive between calls: spawning 16
// threads costs ~0.4 ms, a tenth of a 256-blob proof call.
class WorkerPool {
  public:
    explicit WorkerPool(unsigned n) {
        for (unsigned w = 0; w < n; ++w) th_.emplace_back([this, w] { loop(w); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    unsigned size() const { return (unsigned)th_.size(); }
    // runs fn(w) for w = 0 .. active-1 on the pool and returns when all are done
    void run(unsigned active, const std::function<void(unsigned)>& fn) {
        std::unique_lock<std::mutex> lk(m_);
        job_ = &fn;
        active_ = active;
        pending_ = (unsigned)th_.size();
        ++gen_;
        cv_.notify_all();
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop(unsigned w) {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(unsigned)>* job;
            unsigned active;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                job = job_;
                active = active_;
            }
            if (w < active) (*job)(w);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)>* job_ = nullptr;
    unsigned active_ = 0, pending_ = 0, gen_ = 0;
    bool stop_ = false;
};

constexpr int QT = 512;            // threads per blob
// cell proofs by FK20 from this batch size.  Re-measured after the G1 stages were rewritten (round 4, tools/time_cells.py,
// one settings object per form): 1 / 2 / 3 / 4 / 8 / 16 blobs FK20 4.15 / 4.16 / 4.23 / 4.25 / 4.29 / 5.32 ms, direct form
// 1.91 / 3.16 / 4.52 / 5.75 / 10.88 / 20.65 ms (round 3's crossover was 16 blobs at 25 ms either way)
constexpr size_t FK20_MIN_BLOBS = 3;
constexpr size_t PROVE_CHUNK = 64;  // blobs per pipeline stage of a large compute_blob_kzg_proof batch
constexpr size_t COMMIT_CHUNK = 64;   // smallest pipeline stage of a blob_to_kzg_commitment batch (batches from twice this are pipelined)
constexpr size_t QSPLIT_MAX = 16;  // up to this many blobs (a lane batch) run the multi-workgroup variant (k_quotient_a/b)
constexpr int QE = (int)(N / QT);  // elements per thread (8), element index i = k*QT + t

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
            const u32 s0 = sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3);
            const u32 s1 = sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10);
            w[t & 15] += s0 + w[(t - 7) & 15] + s1;
        }
        const u32 S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25);
        const u32 ch = (e & f) ^ (~e & g);
        const u32 t1 = hh + S1 + ch + K[t] + w[t & 15];
        const u32 S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22);
        const u32 maj = (a & b) ^ (a & c) ^ (b & c);
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

// decode + membership test of np compressed points: up to WIDE_CHECK_MAX points the test runs one wave per point
constexpr size_t WIDE_CHECK_MAX = 4096;
constexpr size_t WIDE_COMMIT_CHECK_MAX = 512;  // ... the commitments of a proof batch: up to this many
// `decoded` (optional) is recorded when the slots are written — before the membership test in the two-kernel form, so a
// consumer that only needs the points (the MSM of a verification call, whose result is thrown away if a test fails) can
// start while the test still runs
void decode_check_enqueue(AffPt* d_pts, int* d_stat, const unsigned char* d_bytes, size_t np, hipStream_t st, bool wide,
                          hipEvent_t decoded = nullptr) {
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


// ---------------- EIP-7594 cells + cell proofs ----------------
__device__ __forceinline__ u32 brev32(u32 v, int bits) { return __builtin_bitreverse32(v) >> (32 - bits); }

// blob bytes -> Montgomery Fr in bit-reversed order (blob_to_polynomial + reverse_bit_order of
// poly_lagrange_to_monomial, kzg/src/das.rs:618-629); status = 1 when an element is >= r
__global__ void __launch_bounds__(256) k_blob_to_fr_brp(ff::Fr* __restrict__ out, int* __restrict__ status,
                                                        const u32* __restrict__ blobs, size_t nblobs) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * N) return;
    const size_t b = t / N, i = t % N;
    bool ok;
    ff::Fr v = fr_load_be(blobs + (b * N + brev32((u32)i, 12)) * 8, &ok);
    if (!ok) status[b] = 1;
    out[t] = ff::to_mont(v);
}

// monomial coefficients (4096) -> zero-extended 8192 (das.rs:260-261)
__global__ void __launch_bounds__(256) k_zero_extend(ff::Fr* __restrict__ ext, const ff::Fr* __restrict__ mono, size_t nblobs) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 2 * N) return;
    const size_t b = t / (2 * N), i = t % (2 * N);
    ext[t] = i < N ? mono[b * N + i] : ff::Fr::zero();
}

// evaluations on the 8192 domain -> cells: bit-reversed order, 32-byte big-endian (das.rs:267-275)
__global__ void __launch_bounds__(256) k_cells_out(u32* __restrict__ cells, const ff::Fr* __restrict__ ev, size_t nblobs) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 2 * N) return;
    const size_t b = t / (2 * N), f = t % (2 * N);
    ff::Fr v = ff::from_mont(ev[b * 2 * N + brev32((u32)f, 13)]);
#pragma unroll
    for (int k = 0; k < 8; ++k) cells[t * 8 + k] = __builtin_bswap32(v.v[7 - k]);
}

// Quotient coefficients of the 128 cell proofs: q_k = p div (X^64 - a_k), a_k = w_128^brp7(k).
// The reference reaches the same commitments through FK20 (Toeplitz FFTs + fft_g1, kzg/src/das.rs:660-696);
// with the 4096-point wide table a proof is simply one more fixed-base MSM, so the division recurrence
//   q_j = p_{j+64} + a_k * q_{j+64}
// is run per (cell, residue class mod 64) and the 128 scalar vectors go to the MSM engine.
__global__ void __launch_bounds__(64) k_cell_quotients(u32* __restrict__ q_out, const ff::Fr* __restrict__ mono,
                                                       const ff::Fr* __restrict__ roots8192, size_t nblobs) {
    const size_t b = blockIdx.x / 128, k = blockIdx.x % 128;
    const int r = threadIdx.x;  // residue class
    const ff::Fr a = roots8192[64 * brev32((u32)k, 7)];
    const ff::Fr* p = mono + b * N;
    u32* q = q_out + (b * 128 + k) * N * 8;
    ff::Fr acc = ff::Fr::zero();
    // j = r + 64*t ; top quotient index is N - 65
#pragma unroll 1
    for (int t = 63; t >= 0; --t) {
        const int j = r + 64 * t;
        ff::Fr v;
        if (t == 63) {
            v = ff::Fr::zero();  // q_j = 0 for j >= N - 64
        } else {
            acc = ff::add(p[j + 64], ff::mul(a, acc));
            v = acc;
        }
        ff::Fr c = ff::from_mont(v);
#pragma unroll
        for (int l = 0; l < 8; ++l) q[(size_t)j * 8 + l] = c.v[l];
    }
}

}  // namespace

// ---------------------------------------------------------------- settings object
struct KzgAmdSettings {
    int device = 0;
    kzgamd::MsmContext* msm = nullptr;  // prepared over g1_lagrange_brp
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // commitment validation runs beside the proving pipeline
    // EIP-7594 state, built on first use
    AffPt* d_monomial = nullptr;              // g1_values_monomial as table slots
    kzgamd::MsmContext* msm_monomial = nullptr;
    kzgamd::MsmContext* msm_xext = nullptr;  // FK20: the 128 columns of 64 points of x_ext_fft_columns, one wide table
    ff::Fr *d_fk_a = nullptr, *d_fk_b = nullptr;  // FK20: n x 64 x 128 Toeplitz vectors / their transforms
    g1::Xyzz *d_fk_h = nullptr, *d_fk_h2 = nullptr;  // FK20: n x 128 points, and the transform scratch
    size_t cap_fk = 0, cap_q = 0;
    void ensure_fk20(size_t nblobs) {
        if (nblobs <= cap_fk) return;
        release_fk20();
        CK_HIP(hipMalloc(&d_fk_a, nblobs * 8192 * sizeof(ff::Fr)));
        CK_HIP(hipMalloc(&d_fk_b, nblobs * 8192 * sizeof(ff::Fr)));
        CK_HIP(hipMalloc(&d_fk_h, nblobs * 128 * sizeof(g1::Xyzz)));
        CK_HIP(hipMalloc(&d_fk_h2, nblobs * 128 * sizeof(g1::Xyzz)));
        cap_fk = nblobs;
    }
    void release_fk20() {
        if (d_fk_a) (void)hipFree(d_fk_a);
        if (d_fk_b) (void)hipFree(d_fk_b);
        if (d_fk_h) (void)hipFree(d_fk_h);
        if (d_fk_h2) (void)hipFree(d_fk_h2);
        d_fk_a = d_fk_b = nullptr;
        d_fk_h = d_fk_h2 = nullptr;
        cap_fk = 0;
    }
    void ensure_q(size_t nblobs) {  // the 128 quotient vectors per blob of the direct cell-proof path (16 MB per blob)
        if (nblobs <= cap_q) return;
        if (d_q) (void)hipFree(d_q);
        d_q = nullptr;
        cap_q = 0;
        CK_HIP(hipMalloc(&d_q, nblobs * 128 * N * 32));
        cap_q = nblobs;
    }
    // Lanes: the reference's callers share one settings object between rayon workers (kzg/src/eip_4844.rs:781-805).
    // A host-buffer call of a few blobs takes the first idle lane — a settings object of its own for everything a call
    // mutates (streams, staging buffers, mutex) that BORROWS the tables and engine handles of its parent — so that up to
    // MAX_LANES + 1 small calls are in flight on the GPU at once (their kernels are a few hundred waves each) instead
    // of queueing on one mutex.  Lane objects are created on demand and live as long as the parent.
    static constexpr size_t LANE_MAX_BLOBS = 16;
    static constexpr int MAX_LANES = 15;
    // Coalescing of concurrent single-blob calls (one queue per entry point): callers push a request; up to
    // MAX_LEADERS of them at a time take everything queued (up to LANE_MAX_BLOBS requests) and run it as ONE batch on a
    // lane, so that the ~10 runtime operations of a pipeline invocation (each of them takes a device-wide lock inside
    // the HIP runtime: ~100 us of serialised host time per invocation) are paid per batch, not per call.
    struct CoalesceQueue {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<void*> pending;
        int leaders = 0;
        int max_leaders = 3, gather_us = 60;  // tuning keys leaders / gather_min / gather_us (apply_options)
        size_t gather_min = 6;
    };
    // the tuning keys of config.h this layer reads, copied once when the settings object is created (apply_options;
    // lanes take their parent's)
    kzgamd::Options opt;
    CoalesceQueue q_commit, q_blob_proof, q_proof;
    bool cfg_device_sha = false;
    size_t cfg_host_check_max = 64;  // see HOST_CHECK_MAX
    size_t cfg_prove_chunk = 0;
    bool cfg_wide_check = true;      // false: single-lane tests
    size_t cfg_prove_first = 0, cfg_commit_first = 0, cfg_commit_chunk = 0;
    int cfg_fk20 = -1;               // -1: by batch size
    void apply_options(const kzgamd::Options& o) {
        using namespace kzgamd;
        opt = o;
        for (CoalesceQueue* q : {&q_commit, &q_blob_proof, &q_proof}) {
            q->max_leaders = (int)o.t[T_LEADERS];
            q->gather_min = (size_t)o.t[T_GATHER_MIN];
            q->gather_us = (int)o.t[T_GATHER_US];
        }
        cfg_device_sha = o.t[T_DEVICE_SHA] != 0;
        cfg_host_check_max = (size_t)o.t[T_HOST_CHECK_MAX];
        cfg_prove_chunk = (size_t)o.t[T_PROVE_CHUNK];
        cfg_wide_check = o.t[T_WIDE_CHECK] != 0;
        cfg_prove_first = (size_t)o.t[T_PROVE_FIRST];
        cfg_commit_first = (size_t)o.t[T_COMMIT_FIRST];
        cfg_commit_chunk = (size_t)o.t[T_COMMIT_CHUNK];
        cfg_fk20 = (int)o.t[T_FK20];
    }
    bool is_lane = false;
    std::atomic<bool> busy{false};
    // page-locked staging for calls of up to LANE_MAX_BLOBS blobs: copies to and from it are truly asynchronous (a
    // copy from / to the caller's pageable memory goes through the runtime's own staging path, which serialises
    // concurrent callers)
    unsigned char* h_in = nullptr;   // LANE_MAX_BLOBS blobs
    unsigned char* h_res = nullptr;  // per blob: 144 B result + 32 B y + 4 B status + 4 B commitment status
    void ensure_pinned() {
        if (h_in) return;
        CK_HIP(hipHostMalloc((void**)&h_in, LANE_MAX_BLOBS * BYTES_PER_BLOB, hipHostMallocDefault));
        CK_HIP(hipHostMalloc((void**)&h_res, LANE_MAX_BLOBS * 256, hipHostMallocDefault));
    }
    // Page-locked blob slots for the callers of the coalesced entry points: a caller copies its blob into a slot on its
    // own thread (in parallel with the other callers) before it queues its request; the batch's kernels read the
    // slots in place.  The pool belongs to the root settings object; a caller that finds it empty leaves the copy
    // to the leader (the lane's own staging).
    struct PinnedSlots {
        static constexpr int NSLOTS = 48;
        std::mutex mu;
        unsigned char* base = nullptr;
        bool failed = false;
        std::vector<unsigned char*> free_;
        unsigned char* acquire(int device) {
            std::lock_guard<std::mutex> lk(mu);
            if (!base && !failed) {
                kzgamd::DeviceGuard on_device(device);
                if (on_device.err != hipSuccess ||
                    hipHostMalloc((void**)&base, (size_t)NSLOTS * BYTES_PER_BLOB, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
                    base = nullptr;
                    failed = true;
                    (void)hipGetLastError();
                } else {
                    for (int i = NSLOTS; i-- > 0;) free_.push_back(base + (size_t)i * BYTES_PER_BLOB);
                }
            }
            if (free_.empty()) return nullptr;
            unsigned char* p = free_.back();
            free_.pop_back();
            return p;
        }
        void release(unsigned char* p) {
            if (!p) return;
            std::lock_guard<std::mutex> lk(mu);
            free_.push_back(p);
        }
        ~PinnedSlots() {
            if (base) (void)hipHostFree(base);
        }
    } slots;
    std::mutex lanes_mu;
    std::vector<std::unique_ptr<KzgAmdSettings>> lanes;
    std::atomic<unsigned> lane_rr{0};
    void* ntt = nullptr;                      // kzgamd_ntt_new(13)
    ff::Fr* d_roots8192 = nullptr;            // roots_of_unity[0..=8192], Montgomery
    ff::Fr *d_fr_a = nullptr, *d_fr_b = nullptr, *d_fr_ext = nullptr;  // 4096, 4096, 8192 per blob
    u32* d_cells = nullptr;
    u32* d_q = nullptr;                       // 128 x 4096 x 8 per blob
    unsigned char* d_proofs = nullptr;
    size_t cap_cells = 0;
    int* d_cstatus = nullptr;
    AffPt* d_cpts = nullptr;  // decoded commitments of a proof batch of <= WIDE_COMMIT_CHECK_MAX blobs (wide check)
    std::mutex mu;
    // staging for the host-buffer entry points
    unsigned char* d_blobs = nullptr;
    u32* d_scalars = nullptr;
    int* d_status = nullptr;
    unsigned char* d_out = nullptr;
    u32* d_z = nullptr;              // n x 32 B big-endian evaluation points
    u32* d_y = nullptr;              // n x 8 u32 canonical y
    unsigned char* d_commit = nullptr;  // n x 48 B
    unsigned char* d_qscratch = nullptr;  // k_quotient_a/b scratch for up to QSPLIT_MAX blobs
    size_t cap_blobs = 0;
    std::unique_ptr<WorkerPool> pool;  // created by the first batched proof call
    // extra streams for the chunk pipeline of large proof batches (created on first use); chunk k runs on
    // pipe_stream(k), `stream` waits for all of them in pipe_join()
    static constexpr int NPIPE = 4;
    hipStream_t pipe[NPIPE] = {};
    hipEvent_t pipe_ev[NPIPE] = {};
    hipEvent_t ev_commit = nullptr;  // the commitments of a proof batch are on the device (recorded on stream2)
    hipEvent_t ev_cells = nullptr;   // the cells of a cells-and-proofs call are ready (recorded on stream; stream2 copies them out)
    // batched verification: staging for [proofs | commitments | G] and the variable-base handle over them, kept
    // between calls (a fresh handle per call cost 1.7 ms of stream / allocation / free round trips)
    std::mutex vmu;                 // one batched verification at a time per settings object (its staging buffers)
    std::vector<uint8_t> vstage;    // host staging of the 2n + 1 compressed points (must outlive the async copy)
    unsigned char* d_vbytes = nullptr;
    AffPt* d_vpts = nullptr;
    int* d_vstat = nullptr;
    hipEvent_t ev_decoded = nullptr;  // the points of a verification call are decoded (their membership test may still run)
    size_t vcap = 0;
    kzgamd::MsmContext* msm_verify = nullptr;
    void ensure_verify(size_t np) {
        if (np <= vcap) return;
        if (d_vbytes) (void)hipFree(d_vbytes);
        if (d_vpts) (void)hipFree(d_vpts);
        if (d_vstat) (void)hipFree(d_vstat);
        d_vbytes = nullptr;
        d_vpts = nullptr;
        d_vstat = nullptr;
        vcap = 0;
        const size_t cap = np < 257 ? 257 : np;
        CK_HIP(hipMalloc(&d_vbytes, cap * 48));
        CK_HIP(hipMalloc(&d_vpts, cap * sizeof(AffPt)));
        CK_HIP(hipMalloc(&d_vstat, cap * sizeof(int)));
        vcap = cap;
    }
    hipStream_t pipe_stream(size_t k) {
        const int j = (int)(k % NPIPE);
        if (!pipe[j]) {
            if (hipStreamCreateWithFlags(&pipe[j], hipStreamNonBlocking) != hipSuccess) {
                pipe[j] = nullptr;
                return stream;
            }
            (void)hipEventCreateWithFlags(&pipe_ev[j], hipEventDisableTiming);
        }
        return pipe[j];
    }
    void pipe_join() {
        for (int j = 0; j < NPIPE; ++j)
            if (pipe[j] && pipe_ev[j]) {
                (void)hipEventRecord(pipe_ev[j], pipe[j]);
                (void)hipStreamWaitEvent(stream, pipe_ev[j], 0);
            }
    }
    // EIP-7594 cell verification / recovery state, built on first use
    std::vector<uint8_t> mono64_bytes;  // g1_values_monomial[0..64) compressed (the interpolation-polynomial commitment)
    AffPt* d_mono64 = nullptr;          // ... decoded and subgroup-checked once, as MSM slots
    ff::Fr* d_rec[4] = {nullptr, nullptr, nullptr, nullptr};  // recovery: four vectors of 8192 field elements
    u32* d_rec_in = nullptr;         // up to 128 cells as canonical limbs
    u32* d_rec_idx = nullptr;        // their cell indices
    ff::Fr* d_pow7 = nullptr;        // 7^i and 7^-i, i < 8192 (coset shifts, das.rs:463-491)
    ff::Fr* d_pow7inv = nullptr;
    bool fk20_unavailable = false;   // the FK20 table could not be built (no HBM left): batches use the direct form
    // verify_cell_kzg_proof_batch: the cells (canonical limbs), their columns and the powers of r, for k_vcell_agg
    u32* d_vc_cells = nullptr;
    u32* d_vc_cols = nullptr;
    ff::Fr* d_vc_pw = nullptr;
    size_t cap_vc = 0;
    void ensure_vcells(size_t n) {
        if (n <= cap_vc) return;
        if (d_vc_cells) (void)hipFree(d_vc_cells);
        if (d_vc_cols) (void)hipFree(d_vc_cols);
        if (d_vc_pw) (void)hipFree(d_vc_pw);
        d_vc_cells = d_vc_cols = nullptr;
        d_vc_pw = nullptr;
        cap_vc = 0;
        const size_t cap = n < 128 ? 128 : n;
        CK_HIP(hipMalloc(&d_vc_cells, cap * CELL_SIZE * 32));
        CK_HIP(hipMalloc(&d_vc_cols, (cap + 2 * CELLS_PER_BLOB + 1) * sizeof(u32)));
        CK_HIP(hipMalloc(&d_vc_pw, cap * sizeof(ff::Fr)));
        cap_vc = cap;
    }
    void ensure_recover() {
        if (d_rec[0]) return;
        for (int k = 0; k < 4; ++k) CK_HIP(hipMalloc(&d_rec[k], 2 * N * sizeof(ff::Fr)));
        CK_HIP(hipMalloc(&d_rec_in, 2 * N * 32));
        CK_HIP(hipMalloc(&d_rec_idx, 128 * sizeof(u32)));
        CK_HIP(hipMalloc(&d_pow7, 2 * N * sizeof(ff::Fr)));
        CK_HIP(hipMalloc(&d_pow7inv, 2 * N * sizeof(ff::Fr)));
        std::vector<ff::Fr> p(2 * N), q(2 * N);
        ff::Fr seven = ff::Fr::zero();
        seven.v[0] = 7;
        seven = ff::to_mont(seven);
        const ff::Fr inv7 = ff::inverse_bgcd(seven);
        p[0] = q[0] = ff::Fr::one();
        for (size_t i = 1; i < 2 * N; ++i) {
            p[i] = ff::mul(p[i - 1], seven);
            q[i] = ff::mul(q[i - 1], inv7);
        }
        CK_HIP(hipMemcpy(d_pow7, p.data(), p.size() * sizeof(ff::Fr), hipMemcpyHostToDevice));
        CK_HIP(hipMemcpy(d_pow7inv, q.data(), q.size() * sizeof(ff::Fr), hipMemcpyHostToDevice));
    }
    std::vector<kzgamd::pairing::G2Jac> g2_monomial;  // [tau^i]G2, i < 65 (host; the pairing checks use [1])
    std::vector<ff::Fr> brp_roots;  // brp_roots_of_unity[0..8192) (host copy, Montgomery)
    ff::Fr* d_brp_roots = nullptr;  // first 4096 = the blob evaluation domain
    ~KzgAmdSettings() {
        lanes.clear();  // before the handles they borrow go away
        if (h_in) (void)hipHostFree(h_in);
        if (h_res) (void)hipHostFree(h_res);
        if (is_lane) {
            msm = nullptr;
            msm_monomial = msm_xext = nullptr;
            d_monomial = nullptr;
            d_brp_roots = nullptr;
            ntt = nullptr;
            d_roots8192 = nullptr;
        }
        if (d_z) (void)hipFree(d_z);
        if (d_y) (void)hipFree(d_y);
        if (d_commit) (void)hipFree(d_commit);
        if (d_qscratch) (void)hipFree(d_qscratch);
        if (ev_commit) (void)hipEventDestroy(ev_commit);
        if (ev_cells) (void)hipEventDestroy(ev_cells);
        if (msm_verify) kzgamd::msm_destroy(msm_verify);
        if (ev_decoded) (void)hipEventDestroy(ev_decoded);
        if (d_mono64) (void)hipFree(d_mono64);
        if (d_vbytes) (void)hipFree(d_vbytes);
        if (d_vpts) (void)hipFree(d_vpts);
        if (d_vstat) (void)hipFree(d_vstat);
        for (int j = 0; j < NPIPE; ++j) {
            if (pipe_ev[j]) (void)hipEventDestroy(pipe_ev[j]);
            if (pipe[j]) (void)hipStreamDestroy(pipe[j]);
        }
        if (d_brp_roots) (void)hipFree(d_brp_roots);
        for (int k = 0; k < 4; ++k)
            if (d_rec[k]) (void)hipFree(d_rec[k]);
        if (d_vc_cells) (void)hipFree(d_vc_cells);
        if (d_vc_cols) (void)hipFree(d_vc_cols);
        if (d_vc_pw) (void)hipFree(d_vc_pw);
        if (d_rec_in) (void)hipFree(d_rec_in);
        if (d_rec_idx) (void)hipFree(d_rec_idx);
        if (d_pow7) (void)hipFree(d_pow7);
        if (d_pow7inv) (void)hipFree(d_pow7inv);
        if (d_monomial) (void)hipFree(d_monomial);
        if (msm_monomial) kzgamd::msm_destroy(msm_monomial);
        if (msm_xext) kzgamd::msm_destroy(msm_xext);
        release_fk20();
        if (ntt) kzgamd_ntt_free(ntt);
        if (d_roots8192) (void)hipFree(d_roots8192);
        release_cells();
        if (d_cstatus) (void)hipFree(d_cstatus);
        if (d_cpts) (void)hipFree(d_cpts);
        if (stream2 && stream2 != stream) (void)hipStreamDestroy(stream2);
        if (msm) kzgamd::msm_destroy(msm);
        if (d_blobs) (void)hipFree(d_blobs);
        if (d_scalars) (void)hipFree(d_scalars);
        if (d_status) (void)hipFree(d_status);
        if (d_out) (void)hipFree(d_out);
        if (stream) (void)hipStreamDestroy(stream);
    }
    void release_cells() {
        if (d_fr_a) (void)hipFree(d_fr_a);
        if (d_fr_b) (void)hipFree(d_fr_b);
        if (d_fr_ext) (void)hipFree(d_fr_ext);
        if (d_cells) (void)hipFree(d_cells);
        if (d_q) (void)hipFree(d_q);
        if (d_proofs) (void)hipFree(d_proofs);
        d_fr_a = d_fr_b = d_fr_ext = nullptr;
        d_cells = d_q = nullptr;
        d_proofs = nullptr;
        cap_cells = 0;
        cap_q = 0;
    }
    void ensure_cells(size_t nblobs) {
        if (nblobs <= cap_cells) return;
        release_cells();
        CK_HIP(hipMalloc(&d_fr_a, nblobs * N * 32));
        CK_HIP(hipMalloc(&d_fr_b, nblobs * N * 32));
        CK_HIP(hipMalloc(&d_fr_ext, nblobs * 2 * N * 32));
        CK_HIP(hipMalloc(&d_cells, nblobs * 2 * N * 32));
        CK_HIP(hipMalloc(&d_proofs, nblobs * 128 * 48));
        cap_cells = nblobs;
    }
    void ensure(size_t nblobs) {
        if (nblobs <= cap_blobs) return;
        if (d_blobs) (void)hipFree(d_blobs);
        if (d_scalars) (void)hipFree(d_scalars);
        if (d_status) (void)hipFree(d_status);
        if (d_out) (void)hipFree(d_out);
        if (d_z) (void)hipFree(d_z);
        if (d_y) (void)hipFree(d_y);
        if (d_commit) (void)hipFree(d_commit);
        if (d_cstatus) (void)hipFree(d_cstatus);
        d_cstatus = nullptr;
        if (d_cpts) (void)hipFree(d_cpts);
        d_cpts = nullptr;
        d_blobs = nullptr;
        d_scalars = nullptr;
        d_status = nullptr;
        d_out = nullptr;
        d_z = nullptr;
        d_y = nullptr;
        d_commit = nullptr;
        cap_blobs = 0;
        CK_HIP(hipMalloc(&d_blobs, nblobs * BYTES_PER_BLOB));
        CK_HIP(hipMalloc(&d_scalars, nblobs * BYTES_PER_BLOB));
        CK_HIP(hipMalloc(&d_status, nblobs * sizeof(int)));
        CK_HIP(hipMalloc(&d_out, nblobs * 144));  // 48-byte compressed results, or Jacobian for the small-batch path
        CK_HIP(hipMalloc(&d_z, nblobs * 32));
        CK_HIP(hipMalloc(&d_y, nblobs * 32));
        CK_HIP(hipMalloc(&d_commit, nblobs * 48));
        CK_HIP(hipMalloc(&d_cstatus, nblobs * sizeof(int)));
        CK_HIP(hipMalloc(&d_cpts, WIDE_COMMIT_CHECK_MAX * sizeof(AffPt)));
        cap_blobs = nblobs;
    }
};

namespace {

std::mutex g_registry_mu;
std::map<const void*, KzgAmdSettings*> g_registry;

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

// The settings object a small host-buffer call runs on: the parent if idle, else an idle lane, else a new lane, else
// (all MAX_LANES busy) one of them in turn — its mutex queues the call.  Released by the destructor.
struct LaneRef {
    KzgAmdSettings* use = nullptr;
    bool flagged = false;
    LaneRef(KzgAmdSettings* dev, size_t nblobs) {
        use = dev;
        if (nblobs > KzgAmdSettings::LANE_MAX_BLOBS) {
            // large batches: the parent's own pipeline, one at a time (its mutex); marked busy so that small calls go
            // to the lanes meanwhile
            flagged = !dev->busy.exchange(true);
            return;
        }
        // small calls run on lanes (their streams have MSM workspaces of their own: no cross-stream events per enqueue);
        // the parent stays free for large batches
        bool expect = false;
        std::lock_guard<std::mutex> lk(dev->lanes_mu);
        for (auto& ln : dev->lanes) {
            expect = false;
            if (ln->busy.compare_exchange_strong(expect, true)) {
                use = ln.get();
                flagged = true;
                return;
            }
        }
        if ((int)dev->lanes.size() < KzgAmdSettings::MAX_LANES) {
            use = make_lane(dev);  // created busy
            flagged = true;
            return;
        }
        use = dev->lanes[dev->lane_rr.fetch_add(1) % dev->lanes.size()].get();
    }
    ~LaneRef() {
        if (flagged) use->busy.store(false);
    }
    LaneRef(const LaneRef&) = delete;
    LaneRef& operator=(const LaneRef&) = delete;
};

size_t reverse_bits(size_t v, unsigned bits) {
    size_t r = 0;
    for (unsigned b = 0; b < bits; ++b)
        if (v & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
    return r;
}

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
// batches up to this size leave the device as Jacobian points and are compressed on the host with one inversion
// (host_p1_compress_batch): k_final's one-lane inversion is 0.25 ms of latency, worth paying only when a block of 64
// points shares it
constexpr size_t HOST_COMPRESS_MAX = 16;
constexpr size_t HOST_CHECK_MAX = 64;  // commitments of a proof batch validated on the host's cores up to this many

void compress_on_host(uint8_t* out48, const blst_p1* jac, size_t n) { kzgamd::host_p1_compress_batch(out48, jac, n); }

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

// proofs for n (blob, z) pairs; z_src = explicit evaluation points or nullptr to derive them
// from the commitments (compute_blob_kzg_proof)
// proofs == nullptr: evaluation only (zs_out receives the derived challenges)
void prove_batch(KZGProof* proofs, Bytes32* ys, const Blob* blobs, const Bytes32* zs, const Bytes48* commitments, size_t n,
                 KzgAmdSettings* dev, Bytes32* zs_out = nullptr, bool commitments_checked_elsewhere = false) {
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
    // KZGAMD_DEVICE_SHA=1: the Fiat-Shamir hashes of a host-buffer batch on the GPU too (k_challenge_sha256, one lane per
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
            hipLaunchKernelGGL(k_challenge_sha256, dim3((unsigned)((cn + 63) / 64)), dim3(64), 0, cs, (u32*)(dev->d_z + off * 8),
                               (const u32*)(dev->d_blobs + off * BYTES_PER_BLOB),
                               (const u32*)(dev->d_commit + off * 48), cn);
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


// ---------------- FK20 cell proofs (compute_fk20_proofs, kzg/src/das.rs:630-696) for batches ----------------
// toeplitz_coeffs_stride for every (blob, offset i < 64): a 128-vector with p[4095 - i] at 0 and p[4095 - i - 64 j] at
// 128 - j, j = 1 .. 62 (the circulant embedding of the Toeplitz matrix of every 64th coefficient)
__global__ void __launch_bounds__(256) k_fk20_toeplitz(ff::Fr* __restrict__ out, const ff::Fr* __restrict__ mono, size_t nblobs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 64 * 128) return;
    const size_t b = t >> 13;
    const u32 i = (u32)(t >> 7) & 63, idx = (u32)t & 127;
    const ff::Fr* p = mono + b * N;
    ff::Fr v = ff::Fr::zero();
    if (idx == 0) v = p[N - 1 - i];
    else if (idx >= 66) v = p[N - 1 - i - 64 * (128 - idx)];
    out[t] = v;
}
// coeffs[blob][j][i] = transform_i[j] / 128: the scalars of column j next to each other (the MSM's layout), with the
// 1/128 of the inverse G1 transform that follows folded in (a field multiplication here instead of a scalar
// multiplication per point there)
__global__ void __launch_bounds__(256) k_fk20_transpose(ff::Fr* __restrict__ out, const ff::Fr* __restrict__ in, size_t nblobs,
                                                        ff::Fr inv128) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 128 * 64) return;
    const size_t b = t >> 13;
    const u32 j = (u32)(t >> 6) & 127, i = (u32)t & 63;
    out[t] = ff::mul(in[(b * 64 + i) * 128 + j], inv128);
}
// h[64 .. 128) = identity (das.rs:688-691)
__global__ void __launch_bounds__(256) k_fk20_zero_upper(g1::Xyzz* __restrict__ h, size_t nblobs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 64) return;
    g1::set_inf(h[(t >> 6) * 128 + 64 + (t & 63)]);
}
// reverse_bit_order of the 128 proofs of a blob (das.rs:288)
__global__ void __launch_bounds__(256) k_fk20_brp(g1::Xyzz* __restrict__ out, const g1::Xyzz* __restrict__ in, size_t nblobs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 128) return;
    out[t] = in[(t & ~(size_t)127) | (__builtin_bitreverse32((u32)t & 127u) >> 25)];
}

// the fixed-base handle over the 128 x 64 points of x_ext_fft_columns (column-major: base j * 64 + i)
void fk20_prepare(KzgAmdSettings* dev, const CKZGSettings* cs) {
    if (dev->msm_xext) return;
    const size_t K2 = 2 * CELLS_PER_BLOB, total = K2 * CELL_SIZE;
    std::vector<ff::Fp> aff(2 * total);  // blst_p1_affine: x, y
    std::vector<ff::Fp> pre(total);
    // Montgomery's trick over the Z coordinates (the columns hold no point at infinity for a valid setup; a zero Z
    // is skipped and its point written as (0, 0), blst's affine infinity)
    ff::Fp run = ff::Fp::one();
    for (size_t k = 0; k < total; ++k) {
        const ff::Fp* P = reinterpret_cast<const ff::Fp*>(&cs->x_ext_fft_columns[k / CELL_SIZE][k % CELL_SIZE]);
        pre[k] = run;
        if (!P[2].is_zero()) run = hfp::mul(run, P[2]);
    }
    ff::Fp inv = ff::inverse_bgcd(run);
    for (size_t k = total; k-- > 0;) {
        const ff::Fp* P = reinterpret_cast<const ff::Fp*>(&cs->x_ext_fft_columns[k / CELL_SIZE][k % CELL_SIZE]);
        if (P[2].is_zero()) {
            aff[2 * k] = aff[2 * k + 1] = ff::Fp::zero();
            continue;
        }
        const ff::Fp zi = hfp::mul(inv, pre[k]), zi2 = hfp::sqr(zi);
        inv = hfp::mul(inv, P[2]);
        aff[2 * k] = hfp::mul(P[0], zi2);
        aff[2 * k + 1] = hfp::mul(P[1], hfp::mul(zi2, zi));
    }
    dev->msm_xext = kzgamd::msm_create(aff.data(), total, false, true, false, kzgamd::G1_TRUSTED, &dev->opt);
}

// The 128 cell proofs of n polynomials whose 4096 monomial coefficients are in dev->d_fr_b, compressed into
// dev->d_proofs (compute_fk20_proofs + reverse_bit_order, kzg/src/das.rs:280-288, 630-696): enqueue only.
// The caller has prepared the handle its `fk20` choice needs and the buffers (ensure_fk20 / ensure_q).
void enqueue_cell_proofs(KzgAmdSettings* dev, size_t n, hipStream_t st, bool fk20) {
    if (fk20) {
        const size_t nv = n * 64 * 128;
        hipLaunchKernelGGL(k_fk20_toeplitz, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, dev->d_fk_a,
                           (const ff::Fr*)dev->d_fr_b, n);
        if (kzgamd_ntt_fr_device(dev->ntt, dev->d_fk_b, dev->d_fk_a, 128, 64 * n, 0, st) != 0) throw CkErr{C_KZG_ERROR, "ntt"};
        ff::Fr k128 = ff::Fr::zero();
        k128.v[0] = 128;
        const ff::Fr inv128 = ff::inverse_bgcd(ff::to_mont(k128));  // Montgomery form of 1/128
        hipLaunchKernelGGL(k_fk20_transpose, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, dev->d_fk_a,
                           (const ff::Fr*)dev->d_fk_b, n, inv128);
        // h_ext_fft[blob][j] = sum_i coeffs[j][i] * x_ext_fft_columns[j][i]: 128 n MSMs of 64 points, column j of the table
        kzgamd::msm_lock(dev->msm_xext);
        try {
            kzgamd::msm_enqueue(dev->msm_xext, dev->d_fk_h, dev->d_fk_a, CELL_SIZE, n * 128, 1, st, kzgamd::OUT_XYZZ, false, 128);
        } catch (...) {
            kzgamd::msm_unlock(dev->msm_xext);
            throw;
        }
        kzgamd::msm_unlock(dev->msm_xext);
        // h = ifft_g1(h_ext_fft), upper half cleared, proofs = fft_g1(h), bit-reversed, compressed
        g1::Xyzz* h = (g1::Xyzz*)kzgamd::fftg1_device((NttCtx*)dev->ntt, dev->d_fk_h, dev->d_fk_h2, 128, n, 1, st, false);
        if (!h) throw CkErr{C_KZG_ERROR, "fft_g1"};
        g1::Xyzz* other = h == dev->d_fk_h ? dev->d_fk_h2 : dev->d_fk_h;
        hipLaunchKernelGGL(k_fk20_zero_upper, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, h, n);
        g1::Xyzz* pr = (g1::Xyzz*)kzgamd::fftg1_device((NttCtx*)dev->ntt, h, other, 128, n, 0, st);
        if (!pr) throw CkErr{C_KZG_ERROR, "fft_g1"};
        g1::Xyzz* fin = pr == h ? other : h;
        hipLaunchKernelGGL(k_fk20_brp, dim3((unsigned)((n * 128 + 255) / 256)), dim3(256), 0, st, fin, (const g1::Xyzz*)pr, n);
        kzgamd::g1_compress_xyzz(dev->d_proofs, fin, n * 128, st);
    } else {
        hipLaunchKernelGGL(k_cell_quotients, dim3((unsigned)(n * 128)), dim3(64), 0, st, dev->d_q, (const ff::Fr*)dev->d_fr_b,
                           (const ff::Fr*)dev->d_roots8192, n);
        kzgamd::msm_lock(dev->msm_monomial);
        try {
            kzgamd::msm_enqueue(dev->msm_monomial, dev->d_proofs, dev->d_q, N, n * 128, 0, st, kzgamd::OUT_COMPRESSED);
        } catch (...) {
            kzgamd::msm_unlock(dev->msm_monomial);
            throw;
        }
        kzgamd::msm_unlock(dev->msm_monomial);
    }
}

// compute_cells_and_kzg_proofs (kzg/src/das.rs:244-292) for n blobs; cells / proofs may be null (not both)
void cells_and_proofs(uint8_t* cells, KZGProof* proofs, const Blob* blobs, size_t n, const CKZGSettings* cs,
                      KzgAmdSettings* dev) {
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    if (!dev->d_roots8192) {
        CK_HIP(hipMalloc(&dev->d_roots8192, (2 * N + 1) * sizeof(ff::Fr)));
        CK_HIP(hipMemcpy(dev->d_roots8192, cs->roots_of_unity, (2 * N + 1) * sizeof(ff::Fr), hipMemcpyHostToDevice));
    }
    // Cell proofs: FK20 for batches (the reference's algorithm: 64 transforms of 128 scalars, 128 MSMs of 64 points
    // over x_ext_fft_columns, two G1 transforms of 128 points — ~25x fewer point additions than 128 MSMs of 4096, but
    // the G1 transforms are 14 serial stages of a 128-bit scalar multiplication each: tens of ms of latency whatever
    // the batch).  A few blobs: the direct form, one more fixed-base MSM per cell over the monomial table.
    // KZGAMD_FK20 = 0 / 1 forces one or the other.
    bool fk20 = proofs && n >= FK20_MIN_BLOBS;
    if (dev->cfg_fk20 >= 0) fk20 = proofs && dev->cfg_fk20 != 0;
    if (dev->fk20_unavailable) fk20 = false;
    if (proofs && fk20) {
        // no HBM left for the FK20 table (creation throws, or succeeds without a wide table): the direct form computes
        // the same proofs; the useless handle is dropped and the choice remembered
        try {
            fk20_prepare(dev, cs);
        } catch (...) {
            dev->fk20_unavailable = true;
        }
        if (!dev->fk20_unavailable && !kzgamd::msm_has_wide_table(dev->msm_xext)) dev->fk20_unavailable = true;
        if (dev->fk20_unavailable) {
            if (dev->msm_xext) kzgamd::msm_destroy(dev->msm_xext);
            dev->msm_xext = nullptr;
            fk20 = false;
        }
    }
    if (proofs && !fk20 && !dev->msm_monomial) dev->msm_monomial = kzgamd::msm_create(dev->d_monomial, N, true, true, true, kzgamd::G1_TRUSTED, &dev->opt);
    dev->ensure(n);
    dev->ensure_cells(n);
    if (proofs && fk20) dev->ensure_fk20(n);
    if (proofs && !fk20) dev->ensure_q(n);
    hipStream_t st = dev->stream;
    CK_HIP(hipMemcpyAsync(dev->d_blobs, blobs, n * BYTES_PER_BLOB, hipMemcpyHostToDevice, st));
    CK_HIP(hipMemsetAsync(dev->d_status, 0, n * sizeof(int), st));
    hipLaunchKernelGGL(k_blob_to_fr_brp, dim3((unsigned)((n * N + 255) / 256)), dim3(256), 0, st, dev->d_fr_a, dev->d_status,
                       (const u32*)dev->d_blobs, n);
    // poly_lagrange_to_monomial: inverse NTT of the bit-reversed evaluations
    if (kzgamd_ntt_fr_device(dev->ntt, dev->d_fr_b, dev->d_fr_a, N, n, 1, st) != 0) throw CkErr{C_KZG_ERROR, "ntt"};
    if (cells) {
        hipLaunchKernelGGL(k_zero_extend, dim3((unsigned)((n * 2 * N + 255) / 256)), dim3(256), 0, st, dev->d_fr_ext,
                           (const ff::Fr*)dev->d_fr_b, n);
        // d_fr_a is free again only for n*N elements; the 8192-point result needs its own buffer: reuse d_cells
        // as scratch for the transform output, then convert in place through d_fr_ext
        ff::Fr* ev = reinterpret_cast<ff::Fr*>(dev->d_cells);
        if (kzgamd_ntt_fr_device(dev->ntt, ev, dev->d_fr_ext, 2 * N, n, 0, st) != 0) throw CkErr{C_KZG_ERROR, "ntt"};
        hipLaunchKernelGGL(k_cells_out, dim3((unsigned)((n * 2 * N + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<u32*>(dev->d_fr_ext), (const ff::Fr*)ev, n);
        // fetched below, after the proof kernels are enqueued: a copy into pageable memory blocks this thread
    }
    // The cells (256 KiB per blob) go back on the second stream while the proof kernels run on the first: the copy
    // engine is idle during FK20 (256 blobs: 67 MB, ~2.5 ms that used to follow the proofs on the same stream).
    const bool side_copy = cells && proofs && dev->stream2 && dev->stream2 != st;
    if (side_copy) {
        if (!dev->ev_cells) CK_HIP(hipEventCreateWithFlags(&dev->ev_cells, hipEventDisableTiming));
        CK_HIP(hipEventRecord(dev->ev_cells, st));
    }
    std::vector<int> status(n);
    try {
        if (proofs) enqueue_cell_proofs(dev, n, st, fk20);
        if (side_copy) {
            CK_HIP(hipStreamWaitEvent(dev->stream2, dev->ev_cells, 0));
            CK_HIP(hipMemcpyAsync(cells, dev->d_fr_ext, n * 2 * N * 32, hipMemcpyDeviceToHost, dev->stream2));
        } else if (cells) {
            CK_HIP(hipMemcpyAsync(cells, dev->d_fr_ext, n * 2 * N * 32, hipMemcpyDeviceToHost, st));
        }
        if (proofs) CK_HIP(hipMemcpyAsync(proofs, dev->d_proofs, n * 128 * 48, hipMemcpyDeviceToHost, st));
        CK_HIP(hipMemcpyAsync(status.data(), dev->d_status, n * sizeof(int), hipMemcpyDeviceToHost, st));
    } catch (...) {
        // a copy into the caller's `cells` / `proofs` (or into `status`) may be in flight on either stream: nothing of it
        // may outlive this call
        (void)hipStreamSynchronize(st);
        if (side_copy) (void)hipStreamSynchronize(dev->stream2);
        throw;
    }
    const hipError_t e1 = hipStreamSynchronize(st);
    if (side_copy) CK_HIP(hipStreamSynchronize(dev->stream2));  // also on the way out of a failure: `cells` is the caller's
    CK_HIP(e1);
    for (size_t i = 0; i < n; ++i) CK_REQUIRE(status[i] == 0, "Invalid scalar");
}


// ---------------- EIP-7594 recovery (kzg/src/das.rs:101-243, 566-657) ----------------
// provided cells (canonical little-endian limbs, already checked < r on the host) -> the 8192 evaluations in
// bit-reversed order, Montgomery form, missing positions and the reference's "null" sentinel as zero
// (recover_cells: `if cells_brp[i].is_null() { zero }`, das.rs:611-617; Fr::null() = from_u64_arr([u64::MAX; 4]),
// blst/src/types/fr.rs:36-38 — a provided element equal to it is dropped by the reference too).  drop_null is false
// when all 128 cells are given: the reference then skips recover_cells (das.rs:172-181) and hands the values as they
// are to poly_lagrange_to_monomial (:186-188), the sentinel value included.
__global__ void __launch_bounds__(256) k_rec_scatter(ff::Fr* __restrict__ ev_brp, const u32* __restrict__ limbs,
                                                     const u32* __restrict__ cell_idx, size_t ncells, bool drop_null) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncells * CELL_SIZE) return;
    const u32 c = cell_idx[t >> 6], j = (u32)t & 63;
    ff::Fr v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v.v[k] = limbs[t * 8 + k];
    v = ff::to_mont(v);
    ff::Fr nul;
#pragma unroll
    for (int k = 0; k < 8; ++k) nul.v[k] = 0xffffffffu;
    nul = ff::to_mont(nul);  // from_u64_arr reduces: (2^256 - 1) mod r in Montgomery form
    if (drop_null && v == nul) v = ff::Fr::zero();
    ev_brp[brev32(c * (u32)CELL_SIZE + j, 13)] = v;
}
__global__ void __launch_bounds__(256) k_fr_mul(ff::Fr* __restrict__ out, const ff::Fr* __restrict__ a, const ff::Fr* __restrict__ b,
                                                size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = fr29::mul_blst(a[t], b[t]);
}
// 1 / x per element (batch_inverse of the vanishing polynomial over the coset, das.rs:628-630: never zero there)
__global__ void __launch_bounds__(64) k_fr_inverse(ff::Fr* __restrict__ data, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) data[t] = ff::inverse_bgcd(data[t]);
}

template <class F>
C_KZG_RET guarded(F&& f) {
    try {
        f();
        return C_KZG_OK;
    } catch (const CkErr& e) {
        static const bool debug = getenv("KZGAMD_DEBUG") != nullptr;
        if (debug) fprintf(stderr, "kzg_mi355x: %s\n", e.what.c_str());
        return e.rc == C_KZG_MALLOC ? C_KZG_MALLOC : C_KZG_BADARGS;  // the reference maps every failure to BadArgs
    } catch (const std::bad_alloc&) {
        return C_KZG_MALLOC;
    } catch (...) {
        return C_KZG_BADARGS;
    }
}

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
    // threads, 3 991 vs 4 040 /s for one — nothing; the call-by-call form stays.  DESIGN.md §9.)
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
        hipLaunchKernelGGL(k_challenge_sha256, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, z, (const u32*)d_blobs,
                           (const u32*)d_commitments, n);
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

namespace {

bool fr_from_be32_checked(ff::Fr& out, const uint8_t* in) {  // FsFr::from_bytes: canonical limbs, false if >= r
    for (int i = 0; i < 8; ++i) {
        const uint8_t* q = in + (7 - i) * 4;
        out.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    u64 borrow = 0;
    for (int k = 0; k < 8; ++k) {
        u64 d = (u64)out.v[k] - ff::FrParams::p(k) - borrow;
        borrow = (d >> 32) & 1;
    }
    return borrow != 0;
}

// verify_kzg_proof_batch (kzg/src/eip_4844.rs:380-435) up to the pairing.  Host: the Fiat-Shamir scalar r
// (compute_r_powers, :328-378) and the three scalar vectors; GPU: decoding + subgroup checks of the 2n points
// (validate_batched_input, :721-734) and the linear combinations — as ONE two-row MSM over [proofs | commitments | G]:
//     row 0:  r^i           0      0                 -> proof_lincomb
//     row 1:  r^i z_i       r^i    -sum r^i y_i      -> rhs  ( = sum r^i (C_i - [y_i]G) + sum r^i z_i proof_i )
// Batched verification, G1 half, in two steps so that the decode + subgroup check of the 2n points (a 1.7 ms latency
// chain on its own stream) runs under whatever the caller does in between — the challenges and evaluations of
// verify_blob_kzg_proof_batch.  The caller holds dev->vmu from begin to finish.
void verify_g1_begin(const Bytes48* commitments, const Bytes48* proofs, size_t n, KzgAmdSettings* dev) {
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    const size_t np = 2 * n + 1;
    dev->ensure_verify(np);
    // device: [proofs | commitments | generator], decoded and checked
    dev->vstage.resize(np * 48);
    memcpy(dev->vstage.data(), proofs, n * 48);
    memcpy(dev->vstage.data() + n * 48, commitments, n * 48);
    static const uint8_t G1_GENERATOR_COMPRESSED[48] = {
        0x97, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f,
        0xc3, 0x68, 0x8c, 0x4f, 0x97, 0x74, 0xb9, 0x05, 0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58,
        0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a, 0xdb, 0x22, 0xc6, 0xbb};
    memcpy(dev->vstage.data() + 2 * n * 48, G1_GENERATOR_COMPRESSED, 48);
    hipStream_t st = dev->stream2;
    CK_HIP(hipMemcpyAsync(dev->d_vbytes, dev->vstage.data(), dev->vstage.size(), hipMemcpyHostToDevice, st));
    CK_HIP(hipMemsetAsync(dev->d_vstat, 0, np * sizeof(int), st));
    CK_HIP(hipMemsetAsync(dev->d_vpts, 0, np * sizeof(AffPt), st));
    if (!dev->ev_decoded) CK_HIP(hipEventCreateWithFlags(&dev->ev_decoded, hipEventDisableTiming));
    decode_check_enqueue(dev->d_vpts, dev->d_vstat, (const unsigned char*)dev->d_vbytes, np, st, dev->cfg_wide_check, dev->ev_decoded);
    CK_HIP(hipGetLastError());
}

void verify_g1_finish(blst_p1* proof_lincomb, blst_p1* rhs, const Bytes48* commitments, const Bytes32* zs, const Bytes32* ys,
                      const Bytes48* proofs, size_t n, KzgAmdSettings* dev) {
    std::vector<ff::Fr> z(n), y(n);
    bool scalars_ok = true;
    for (size_t i = 0; i < n; ++i) {
        scalars_ok = scalars_ok && fr_from_be32_checked(z[i], zs[i].bytes) && fr_from_be32_checked(y[i], ys[i].bytes);
        z[i] = ff::to_mont(z[i]);
        y[i] = ff::to_mont(y[i]);
    }
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    const size_t np = 2 * n + 1;
    hipStream_t st = dev->stream2;
    // host, meanwhile: r = hash_to_bls_field(sha256(domain | 4096 | n | (C_i | z_i | y_i | proof_i)...)), powers of r
    std::vector<ff::Fr> sc(2 * np, ff::Fr::zero());
    {
        kzgamd::Sha256 h;
        uint8_t head[32] = {0};
        memcpy(head, "RCKZGBATCH___V1_", 16);
        const uint64_t nfe = N, nn = n;
        for (int i = 0; i < 8; ++i) {
            head[16 + 7 - i] = (uint8_t)(nfe >> (8 * i));
            head[24 + 7 - i] = (uint8_t)(nn >> (8 * i));
        }
        h.update(head, 32);
        for (size_t i = 0; i < n; ++i) {
            h.update(commitments[i].bytes, 48);
            h.update(zs[i].bytes, 32);
            h.update(ys[i].bytes, 32);
            h.update(proofs[i].bytes, 48);
        }
        uint8_t digest[32];
        h.finish(digest);
        ff::Fr v;
        for (int i = 0; i < 8; ++i) {
            const uint8_t* q = digest + (7 - i) * 4;
            v.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
        }
        const ff::Fr r = ff::mul(v, ff::Fr::r2());  // Montgomery form of (v mod r)
        ff::Fr pw = ff::Fr::one(), sy = ff::Fr::zero();
        for (size_t i = 0; i < n; ++i) {
            sc[i] = pw;                          // row 0: proofs
            sc[np + i] = ff::mul(pw, z[i]);      // row 1: proofs
            sc[np + n + i] = pw;                 // row 1: commitments
            sy = ff::add(sy, ff::mul(pw, y[i]));
            pw = ff::mul(pw, r);
        }
        sc[np + 2 * n] = ff::neg(sy);            // row 1: generator
    }
    // The MSM starts as soon as the points are decoded, next to their membership test (0.25 ms on stream2): if a point
    // fails it — or is no encoding at all: its slot stays zero — the sums below are garbage that nobody reads.
    blst_p1 out[2];
    if (scalars_ok) {
        try {
            CK_HIP(hipEventSynchronize(dev->ev_decoded));
            if (!dev->msm_verify) dev->msm_verify = kzgamd::msm_create(dev->d_vpts, np, true, false, true, kzgamd::G1_TRUSTED, &dev->opt);
            else kzgamd::msm_reset_points(dev->msm_verify, dev->d_vpts, np);
            kzgamd::msm_run_host(dev->msm_verify, out, sc.data(), np, 2);
        } catch (...) {
            (void)hipStreamSynchronize(st);  // nothing of this call stays in flight
            throw;
        }
    }
    std::vector<int> stat(np);
    CK_HIP(hipMemcpyAsync(stat.data(), dev->d_vstat, np * sizeof(int), hipMemcpyDeviceToHost, st));
    CK_HIP(hipStreamSynchronize(st));
    CK_REQUIRE(scalars_ok, "Invalid scalar");
    for (size_t i = 0; i < np; ++i) CK_REQUIRE(stat[i] != 1, "Invalid G1 encoding");
    for (size_t i = 0; i < n; ++i) CK_REQUIRE(stat[i] == 0, "Invalid proof");
    for (size_t i = n; i < 2 * n; ++i) CK_REQUIRE(stat[i] == 0, "Invalid commitment");
    *proof_lincomb = out[0];
    *rhs = out[1];
}

void verify_batch_g1(blst_p1* proof_lincomb, blst_p1* rhs, const Bytes48* commitments, const Bytes32* zs, const Bytes32* ys,
                     const Bytes48* proofs, size_t n, KzgAmdSettings* dev) {
    std::lock_guard<std::mutex> vlk(dev->vmu);
    // (the scalars are validated before anything is launched, as before)
    for (size_t i = 0; i < n; ++i) {
        ff::Fr t;
        CK_REQUIRE(fr_from_be32_checked(t, zs[i].bytes) && fr_from_be32_checked(t, ys[i].bytes), "Invalid scalar");
    }
    verify_g1_begin(commitments, proofs, n, dev);
    verify_g1_finish(proof_lincomb, rhs, commitments, zs, ys, proofs, n, dev);
}

}  // namespace

extern "C" C_KZG_RET kzgamd_verify_kzg_proof_batch_g1(blst_p1* proof_lincomb_out, blst_p1* rhs_out, const Bytes48* commitments,
                                                      const Bytes32* zs, const Bytes32* ys, const Bytes48* proofs, size_t n,
                                                      const CKZGSettings* s) {
    if (!proof_lincomb_out || !rhs_out) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) {
        memset(proof_lincomb_out, 0, sizeof *proof_lincomb_out);
        memset(rhs_out, 0, sizeof *rhs_out);
        return C_KZG_OK;
    }
    if (!commitments || !zs || !ys || !proofs) return C_KZG_BADARGS;
    return guarded([&] { verify_batch_g1(proof_lincomb_out, rhs_out, commitments, zs, ys, proofs, n, dev); });
}

// verify_blob_kzg_proof_batch (kzg/src/eip_4844.rs:736-832) up to the pairing: challenges + evaluations on the GPU
// (:690-719), then the G1 half above.  The caller finishes with  e(proof_lincomb, [tau]G2) == e(rhs, G2).
extern "C" C_KZG_RET kzgamd_verify_blob_kzg_proof_batch_g1(blst_p1* proof_lincomb_out, blst_p1* rhs_out, const Blob* blobs,
                                                           const Bytes48* commitments, const Bytes48* proofs, size_t n,
                                                           const CKZGSettings* s) {
    if (!proof_lincomb_out || !rhs_out) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) {
        memset(proof_lincomb_out, 0, sizeof *proof_lincomb_out);
        memset(rhs_out, 0, sizeof *rhs_out);
        return C_KZG_OK;
    }
    if (!blobs || !commitments || !proofs) return C_KZG_BADARGS;
    return guarded([&] {
        std::vector<Bytes32> zs(n), ys(n);
        std::lock_guard<std::mutex> vlk(dev->vmu);
        verify_g1_begin(commitments, proofs, n, dev);  // decode + subgroup check run under the evaluations
        try {
            prove_batch(nullptr, ys.data(), blobs, nullptr, commitments, n, dev, zs.data(), true);
        } catch (...) {
            (void)hipStreamSynchronize(dev->stream2);
            throw;
        }
        verify_g1_finish(proof_lincomb_out, rhs_out, commitments, zs.data(), ys.data(), proofs, n, dev);
    });
}

namespace {

// check_proof_single (blst/src/types/kzg_settings.rs:178-196) on decoded, validated inputs.  The reference tests
//     e(C - [y]G, G2) == e(proof, [tau]G2 - [z]G2);
// with the [z] moved to the G1 side (bilinearity; the proof is a checked r-torsion point) the same statement is
//     e(C - [y]G + [z]proof, G2) == e(proof, [tau]G2),
// which pairs with the two fixed G2 points of the setup only: their line tables are cached (host_pairing.h), and the
// G2 scalar multiplication becomes a G1 one.  One pairing-product check on the host (the reference keeps the pairing
// on the CPU too).
bool check_proof_single(const blst_p1& commitment, const blst_p1& proof, const ff::Fr& z_plain, const ff::Fr& y_plain,
                        KzgAmdSettings* dev) {
    using namespace kzgamd::pairing;
    kzgamd::HostJac g;
    {
        const uint64_t GX[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull,
                                0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
        const uint64_t GY[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull,
                                0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
        for (int k = 0; k < 6; ++k) {
            g.x.v[2 * k] = (u32)GX[k];
            g.x.v[2 * k + 1] = (u32)(GX[k] >> 32);
            g.y.v[2 * k] = (u32)GY[k];
            g.y.v[2 * k + 1] = (u32)(GY[k] >> 32);
        }
        g.z = ff::Fp::one();
    }
    g.y = hfp::neg(g.y);  // -G
    kzgamd::HostJac pi;
    memcpy(&pi, &proof, sizeof pi);
    // [z]proof - [y]G: one joint double-and-add over the 255 bits
    kzgamd::HostJac acc;
    acc.x = acc.y = acc.z = ff::Fp::zero();
    for (int bit = 254; bit >= 0; --bit) {
        acc = kzgamd::host_jac_dbl(acc);
        if ((y_plain.v[bit >> 5] >> (bit & 31)) & 1) acc = kzgamd::host_jac_add(acc, g);
        if ((z_plain.v[bit >> 5] >> (bit & 31)) & 1) acc = kzgamd::host_jac_add(acc, pi);
    }
    kzgamd::HostJac c;
    memcpy(&c, &commitment, sizeof c);
    const kzgamd::HostJac lhs = kzgamd::host_jac_add(c, acc);
    blst_p1 a1;
    memcpy(&a1, &lhs, sizeof a1);
    const G2Jac g2gen = g2_generator();
    blst_p2 a2, b2;
    memcpy(&a2, &g2gen, sizeof a2);
    memcpy(&b2, &dev->g2_monomial[1], sizeof b2);
    return pairings_verify(&a1, &a2, &proof, &b2);
}

// FsG1::from_bytes + the is_inf / is_valid test of verify_kzg_proof_rust (kzg/src/eip_4844.rs:603-608)
void decode_valid_g1(blst_p1& out, const uint8_t* bytes, const char* what) {
    CK_REQUIRE(kzgamd::host_p1_uncompress(&out, bytes), std::string("Invalid ") + what);
    CK_REQUIRE(kzgamd::host_p1_in_g1(&out), std::string("Invalid ") + what);
}

}  // namespace

// blst/src/eip_4844.rs:383-405 -> verify_kzg_proof_raw (kzg/src/eip_4844.rs:613-637)
extern "C" C_KZG_RET verify_kzg_proof(bool* ok, const Bytes48* commitment_bytes, const Bytes32* z_bytes, const Bytes32* y_bytes,
                                      const Bytes48* proof_bytes, const CKZGSettings* s) {
    if (!ok || !commitment_bytes || !z_bytes || !y_bytes || !proof_bytes) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    return guarded([&] {
        blst_p1 c, pr;
        ff::Fr z, y;
        CK_REQUIRE(kzgamd::host_p1_uncompress(&c, commitment_bytes->bytes), "Invalid commitment");
        CK_REQUIRE(fr_from_be32_checked(z, z_bytes->bytes), "Invalid scalar");
        CK_REQUIRE(fr_from_be32_checked(y, y_bytes->bytes), "Invalid scalar");
        CK_REQUIRE(kzgamd::host_p1_uncompress(&pr, proof_bytes->bytes), "Invalid proof");
        CK_REQUIRE(kzgamd::host_p1_in_g1(&c), "Invalid commitment");
        CK_REQUIRE(kzgamd::host_p1_in_g1(&pr), "Invalid proof");
        *ok = check_proof_single(c, pr, z, y, dev);
    });
}

// blst/src/eip_4844.rs:410-430 -> verify_blob_kzg_proof_raw (kzg/src/eip_4844.rs:667-688): challenge and evaluation
// on the GPU, one pairing check on the host
extern "C" C_KZG_RET verify_blob_kzg_proof(bool* ok, const Blob* blob, const Bytes48* commitment_bytes,
                                           const Bytes48* proof_bytes, const CKZGSettings* s) {
    if (!ok || !blob || !commitment_bytes || !proof_bytes) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    return guarded([&] {
        blst_p1 c, pr;
        CK_REQUIRE(host_blob_valid(blob->bytes), "Invalid scalar");           // bytes_to_blob
        // the two points are decoded and subgroup-checked (0.3 ms of one core each) on two helper threads while this
        // one hashes the challenge and the GPU evaluates the polynomial
        bool c_ok = false, pr_ok = false;
        std::thread tc, tp;
        struct Joiner {  // declared before the threads start: an exception (also out of the second thread's creation)
            std::thread &a, &b;  // must not leave a joinable thread behind
            ~Joiner() {
                if (a.joinable()) a.join();
                if (b.joinable()) b.join();
            }
        } joiner{tc, tp};
        tc = std::thread([&] {
            c_ok = kzgamd::host_p1_uncompress(&c, commitment_bytes->bytes) && kzgamd::host_p1_in_g1(&c);  // infinity passes
        });
        tp = std::thread([&] {
            pr_ok = kzgamd::host_p1_uncompress(&pr, proof_bytes->bytes) && kzgamd::host_p1_in_g1(&pr);
        });
        Bytes32 zb, yb;
        {
            LaneRef lane(dev, 1);
            prove_batch(nullptr, &yb, blob, nullptr, commitment_bytes, 1, lane.use, &zb, true);
        }
        tc.join();
        tp.join();
        CK_REQUIRE(c_ok, "Invalid commitment");
        CK_REQUIRE(pr_ok, "Invalid proof");
        ff::Fr z, y;
        CK_REQUIRE(fr_from_be32_checked(z, zb.bytes) && fr_from_be32_checked(y, yb.bytes), "Invalid scalar");
        *ok = check_proof_single(c, pr, z, y, dev);
    });
}

// blst/src/eip_4844.rs:435-471 -> verify_blob_kzg_proof_batch_raw (kzg/src/eip_4844.rs:736-866): n == 0 is true,
// n == 1 the single verification, otherwise challenges, evaluations and the three linear combinations on the GPU
// and ONE pairing check e(sum r^i proof_i, [tau]G2) == e(rhs, G2) on the host
extern "C" C_KZG_RET verify_blob_kzg_proof_batch(bool* ok, const Blob* blobs, const Bytes48* commitments_bytes,
                                                 const Bytes48* proofs_bytes, size_t n, const CKZGSettings* s) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) {
        *ok = true;
        return C_KZG_OK;
    }
    if (!blobs || !commitments_bytes || !proofs_bytes) return C_KZG_BADARGS;
    if (n == 1) return verify_blob_kzg_proof(ok, blobs, commitments_bytes, proofs_bytes, s);
    return guarded([&] {
        blst_p1 pl, rhs;
        std::vector<Bytes32> zs(n), ys(n);
        {
            std::lock_guard<std::mutex> vlk(dev->vmu);
            verify_g1_begin(commitments_bytes, proofs_bytes, n, dev);  // decode + subgroup check run under the evaluations
            try {
                prove_batch(nullptr, ys.data(), blobs, nullptr, commitments_bytes, n, dev, zs.data(), true);
            } catch (...) {
                (void)hipStreamSynchronize(dev->stream2);  // the decode kernel reads dev->vstage's device copy: drain it
                throw;
            }
            verify_g1_finish(&pl, &rhs, commitments_bytes, zs.data(), ys.data(), proofs_bytes, n, dev);
        }
        blst_p2 g2gen, g2tau;
        const kzgamd::pairing::G2Jac gen = kzgamd::pairing::g2_generator();
        memcpy(&g2gen, &gen, sizeof g2gen);
        memcpy(&g2tau, &dev->g2_monomial[1], sizeof g2tau);
        *ok = kzgamd::pairing::pairings_verify(&pl, &g2tau, &rhs, &g2gen);
    });
}


// ================================================================ EIP-7594: cell verification and recovery
namespace {

constexpr size_t CELLS_PER_EXT_BLOB = 2 * CELLS_PER_BLOB;  // 128
constexpr size_t BYTES_PER_CELL = CELL_SIZE * 32;

inline u32 rbl7(u32 i) {  // CELL_INDICES_RBL (das.rs:87-96): reverse_bits_limited(128, i)
    u32 r = 0;
    for (int b = 0; b < 7; ++b)
        if (i & (1u << b)) r |= 1u << (6 - b);
    return r;
}

inline void put_u64_be(uint8_t* p, uint64_t v) {
    for (int i = 0; i < 8; ++i) p[7 - i] = (uint8_t)(v >> (8 * i));
}

// hash_to_bls_field (kzg/src/eip_4844.rs:916-918): 32 big-endian bytes reduced mod r, Montgomery form
inline ff::Fr hash_to_fr(const uint8_t digest[32]) {
    ff::Fr v;
    for (int i = 0; i < 8; ++i) {
        const uint8_t* q = digest + (7 - i) * 4;
        v.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    return ff::mul(v, ff::Fr::r2());
}

// compute_verify_cell_kzg_proof_batch_challenge (kzg/src/das.rs:391-452) on the caller's bytes: FsG1::to_bytes /
// FsFr::to_bytes of a decoded, valid input are the input bytes themselves
ff::Fr cell_batch_challenge(const Bytes48* commitments, size_t ncommit, const uint64_t* commitment_indices,
                            const uint64_t* cell_indices, const Cell* cells, const Bytes48* proofs, size_t ncells) {
    kzgamd::Sha256 h;
    uint8_t head[48];
    memcpy(head, "RCKZGCBATCH__V1_", 16);
    put_u64_be(head + 16, N);
    put_u64_be(head + 24, CELL_SIZE);
    put_u64_be(head + 32, ncommit);
    put_u64_be(head + 40, ncells);
    h.update(head, 48);
    for (size_t i = 0; i < ncommit; ++i) h.update(commitments[i].bytes, 48);
    for (size_t i = 0; i < ncells; ++i) {
        uint8_t ix[16];
        put_u64_be(ix, commitment_indices[i]);
        put_u64_be(ix + 8, cell_indices[i]);
        h.update(ix, 16);
        h.update(cells[i].bytes, BYTES_PER_CELL);
        h.update(proofs[i].bytes, 48);
    }
    uint8_t digest[32];
    h.finish(digest);
    return hash_to_fr(digest);
}

// cells -> field elements (FsFr::from_bytes per element, c_bindings.rs:225-233): canonical limbs, false if any >= r
bool cells_to_limbs(std::vector<ff::Fr>& out, const Cell* cells, size_t ncells) {
    out.resize(ncells * CELL_SIZE);
    bool ok = true;
    for (size_t i = 0; i < ncells; ++i)
        for (size_t j = 0; j < CELL_SIZE; ++j) ok = fr_from_be32_checked(out[i * CELL_SIZE + j], cells[i].bytes + 32 * j) && ok;
    return ok;
}

// decode `np` compressed G1 points on the GPU (stream2): AffPt slots in dev->d_vpts, per-point status in dev->d_vstat
// (0 ok, 1 not an encoding of a curve point, 2 on the curve but outside G1).  Caller holds dev->vmu.
// `tail` (optional): `ntail` slots decoded and checked by an earlier call, appended behind the np decoded ones (status 0)
void decode_points_begin(KzgAmdSettings* dev, const std::vector<uint8_t>& bytes, size_t np, const AffPt* tail = nullptr,
                         size_t ntail = 0) {
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    dev->ensure_verify(np + ntail);
    dev->vstage = bytes;
    hipStream_t st = dev->stream2;
    CK_HIP(hipMemcpyAsync(dev->d_vbytes, dev->vstage.data(), np * 48, hipMemcpyHostToDevice, st));
    CK_HIP(hipMemsetAsync(dev->d_vstat, 0, (np + ntail) * sizeof(int), st));
    CK_HIP(hipMemsetAsync(dev->d_vpts, 0, np * sizeof(AffPt), st));
    if (ntail) CK_HIP(hipMemcpyAsync(dev->d_vpts + np, tail, ntail * sizeof(AffPt), hipMemcpyDeviceToDevice, st));
    if (!dev->ev_decoded) CK_HIP(hipEventCreateWithFlags(&dev->ev_decoded, hipEventDisableTiming));
    decode_check_enqueue(dev->d_vpts, dev->d_vstat, (const unsigned char*)dev->d_vbytes, np, st, dev->cfg_wide_check, dev->ev_decoded);
    CK_HIP(hipGetLastError());
}
std::vector<int> decode_points_status(KzgAmdSettings* dev, size_t np) {
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    std::vector<int> stat(np);
    CK_HIP(hipMemcpyAsync(stat.data(), dev->d_vstat, np * sizeof(int), hipMemcpyDeviceToHost, dev->stream2));
    CK_HIP(hipStreamSynchronize(dev->stream2));
    return stat;
}

// The aggregated interpolation polynomial of verify_cell_kzg_proof_batch
// (compute_commitment_to_aggregated_interpolation_poly, kzg/src/das.rs:778-835) on the GPU — on the host its ~50 000
// field multiplications were three quarters of a 128-cell call.
// k_vcell_agg: agg[col][brp6(f)] = sum over the cells i of column col of r^i * cell_i[f]  (r^i Montgomery, the cell
// elements canonical: the products and sums stay canonical; columns nobody asked about stay zero);
// then 128 inverse transforms of 64 values (ntt.hip);
// k_vcell_interp: interp[k] = sum_col v[col][k] * h_col^-k,  h_col^-k = roots_of_unity[(8192 - rbl7(col)) k mod 8192].
// cols = [start of column 0 .. 128 in `order` (129 words) | order: the cells' indices grouped by column, ascending inside]
__global__ void __launch_bounds__(256) k_vcell_agg(ff::Fr* __restrict__ agg, const u32* __restrict__ cells,
                                                   const u32* __restrict__ cols, const ff::Fr* __restrict__ pw) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= CELLS_PER_EXT_BLOB * CELL_SIZE) return;
    const u32 col = t >> 6, f = t & 63u;
    ff::Fr acc = ff::Fr::zero();
    const u32* order = cols + CELLS_PER_EXT_BLOB + 1;
    for (u32 j = cols[col]; j < cols[col + 1]; ++j) {
        const size_t i = order[j];
        ff::Fr c;
#pragma unroll
        for (int k = 0; k < 8; ++k) c.v[k] = cells[(i * CELL_SIZE + f) * 8 + k];
        acc = ff::add(acc, fmul(pw[i], c));
    }
    agg[col * CELL_SIZE + (__builtin_bitreverse32(f) >> 26)] = acc;
}
__global__ void __launch_bounds__(128) k_vcell_interp(ff::Fr* __restrict__ out, const ff::Fr* __restrict__ v,
                                                      const ff::Fr* __restrict__ roots8192) {
    __shared__ ff::Fr sh[CELLS_PER_EXT_BLOB];
    const u32 k = blockIdx.x, col = threadIdx.x;
    const u32 rbl = __builtin_bitreverse32(col) >> 25;  // CELL_INDICES_RBL (das.rs:87-96)
    const u32 idx = ((2u * (u32)N - rbl) * k) & (2u * (u32)N - 1u);
    sh[col] = fmul(roots8192[idx], v[col * CELL_SIZE + k]);  // Montgomery x canonical -> canonical
    __syncthreads();
    for (u32 off = CELLS_PER_EXT_BLOB / 2; off > 0; off >>= 1) {
        if (col < off) sh[col] = ff::add(sh[col], sh[col + off]);
        __syncthreads();
    }
    if (col == 0) out[k] = sh[0];
}

// verify_cell_kzg_proof_batch (kzg/src/das.rs:294-389).  Host: parsing, the Fiat-Shamir scalar, the powers of r.
// GPU: the aggregated interpolation polynomial (k_vcell_agg, <= 128 inverse transforms of 64 values, k_vcell_interp;
// compute_commitment_to_aggregated_interpolation_poly, :778-835), decoding + subgroup checks of proofs and commitments, and every linear
// combination as ONE two-row MSM over [proofs | unique commitments | g1_monomial[0..64)]:
//     row 0:  r^i             0          0        -> proof_lincomb
//     row 1:  r^i h_k(i)^64   weight_j   -I_k     -> sum_j w_j C_j - [I(s)] + sum_i r^i h^64 proof_i
// then one pairing check e(row 1, G2) == e(row 0, [s^64]G2) on the host.
void verify_cells(bool* ok, const Bytes48* commitments_bytes, const uint64_t* cell_indices, const Cell* cells,
                  const Bytes48* proofs_bytes, size_t n, const CKZGSettings* cs, KzgAmdSettings* dev) {
    for (size_t i = 0; i < n; ++i) CK_REQUIRE(cell_indices[i] < CELLS_PER_EXT_BLOB, "Invalid cell index");
    // deduplicate_with_indices (das.rs:57-76): first occurrences, in order
    std::vector<Bytes48> uniq;
    std::vector<uint64_t> cidx(n);
    for (size_t i = 0; i < n; ++i) {
        size_t j = 0;
        while (j < uniq.size() && memcmp(uniq[j].bytes, commitments_bytes[i].bytes, 48) != 0) ++j;
        if (j == uniq.size()) uniq.push_back(commitments_bytes[i]);
        cidx[i] = j;
    }
    const size_t m = uniq.size(), np = n + m + CELL_SIZE;
    std::lock_guard<std::mutex> vlk(dev->vmu);
    if (dev->mono64_bytes.empty()) {
        dev->mono64_bytes.resize(CELL_SIZE * 48);
        compress_on_host(dev->mono64_bytes.data(), cs->g1_values_monomial, CELL_SIZE);
    }
    // the 64 setup points are decoded and tested by the first call of a settings object and kept as slots (d_mono64)
    const bool have_mono = dev->d_mono64 != nullptr;
    const size_t ndec = have_mono ? n + m : np;
    std::vector<uint8_t> stage(ndec * 48);
    memcpy(stage.data(), proofs_bytes, n * 48);
    memcpy(stage.data() + n * 48, uniq.data(), m * 48);
    if (!have_mono) memcpy(stage.data() + (n + m) * 48, dev->mono64_bytes.data(), CELL_SIZE * 48);
    decode_points_begin(dev, stage, ndec, dev->d_mono64, have_mono ? CELL_SIZE : 0);
    // host, meanwhile (the decode + subgroup tests are 0.75 ms of GPU latency): the cells' field elements, ...
    std::vector<ff::Fr> cf;
    if (!cells_to_limbs(cf, cells, n)) {
        (void)decode_points_status(dev, np);  // nothing of this call stays in flight
        throw CkErr{C_KZG_BADARGS, "Invalid scalar"};
    }
    const ff::Fr* roots = reinterpret_cast<const ff::Fr*>(cs->roots_of_unity);
    const ff::Fr r = cell_batch_challenge(uniq.data(), m, cidx.data(), cell_indices, cells, proofs_bytes, n);
    std::vector<ff::Fr> sc(2 * np, ff::Fr::zero());
    std::vector<ff::Fr> pws(n);
    std::vector<u32> cols32(CELLS_PER_EXT_BLOB + 1 + n, 0u);  // column starts, then the cells grouped by column (k_vcell_agg)
    for (size_t i = 0; i < n; ++i) ++cols32[(size_t)cell_indices[i] + 1];
    for (size_t c = 0; c < CELLS_PER_EXT_BLOB; ++c) cols32[c + 1] += cols32[c];
    {
        std::vector<u32> cursor(cols32.begin(), cols32.begin() + CELLS_PER_EXT_BLOB);
        for (size_t i = 0; i < n; ++i) cols32[CELLS_PER_EXT_BLOB + 1 + cursor[(size_t)cell_indices[i]]++] = (u32)i;
    }
    ff::Fr pw = ff::Fr::one();
    for (size_t i = 0; i < n; ++i) {
        const size_t col = (size_t)cell_indices[i];
        sc[i] = pw;                                                                     // row 0: proofs
        sc[np + i] = ff::mul(pw, roots[rbl7((u32)col) * CELL_SIZE]);                    // row 1: r^i * h_k^64 (:837-884)
        sc[np + n + cidx[i]] = ff::add(sc[np + n + cidx[i]], pw);                       // row 1: commitment weights (:698-743)
        pws[i] = pw;
        pw = ff::mul(pw, r);
    }
    // the aggregated interpolation polynomial (:778-835) on the GPU: k_vcell_agg, 128 inverse transforms of 64, k_vcell_interp
    std::vector<ff::Fr> interp(CELL_SIZE);
    {
        std::lock_guard<std::mutex> lk(dev->mu);
        kzgamd::DeviceGuard on_device(dev->device);
        CK_HIP(on_device.err);
        dev->ensure_recover();
        dev->ensure_vcells(n);
        if (!dev->d_roots8192) {
            CK_HIP(hipMalloc(&dev->d_roots8192, (2 * N + 1) * sizeof(ff::Fr)));
            CK_HIP(hipMemcpy(dev->d_roots8192, cs->roots_of_unity, (2 * N + 1) * sizeof(ff::Fr), hipMemcpyHostToDevice));
        }
        hipStream_t st = dev->stream;
        CK_HIP(hipMemcpyAsync(dev->d_vc_cells, cf.data(), n * CELL_SIZE * 32, hipMemcpyHostToDevice, st));
        CK_HIP(hipMemcpyAsync(dev->d_vc_cols, cols32.data(), cols32.size() * sizeof(u32), hipMemcpyHostToDevice, st));
        CK_HIP(hipMemcpyAsync(dev->d_vc_pw, pws.data(), n * sizeof(ff::Fr), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_vcell_agg, dim3((unsigned)(CELLS_PER_EXT_BLOB * CELL_SIZE / 256)), dim3(256), 0, st, dev->d_rec[0],
                           (const u32*)dev->d_vc_cells, (const u32*)dev->d_vc_cols, (const ff::Fr*)dev->d_vc_pw);
        if (kzgamd_ntt_fr_device(dev->ntt, dev->d_rec[1], dev->d_rec[0], CELL_SIZE, CELLS_PER_EXT_BLOB, 1, st) != 0)
            throw CkErr{C_KZG_ERROR, "ntt"};
        hipLaunchKernelGGL(k_vcell_interp, dim3((unsigned)CELL_SIZE), dim3((unsigned)CELLS_PER_EXT_BLOB), 0, st, dev->d_rec[2],
                           (const ff::Fr*)dev->d_rec[1], (const ff::Fr*)dev->d_roots8192);
        CK_HIP(hipMemcpyAsync(interp.data(), dev->d_rec[2], CELL_SIZE * sizeof(ff::Fr), hipMemcpyDeviceToHost, st));
        CK_HIP(hipStreamSynchronize(st));
    }
    for (size_t k = 0; k < CELL_SIZE; ++k) interp[k] = ff::to_mont(interp[k]);  // the kernels work on canonical values
    for (size_t k = 0; k < CELL_SIZE; ++k) sc[np + n + m + k] = ff::neg(interp[k]);
    // the MSM next to the membership test of its points, as in verify_g1_finish
    blst_p1 out[2];
    try {
        std::lock_guard<std::mutex> lk(dev->mu);
        kzgamd::DeviceGuard on_device(dev->device);
        CK_HIP(on_device.err);
        CK_HIP(hipEventSynchronize(dev->ev_decoded));
        if (!dev->msm_verify) dev->msm_verify = kzgamd::msm_create(dev->d_vpts, np, true, false, true, kzgamd::G1_TRUSTED, &dev->opt);
        else kzgamd::msm_reset_points(dev->msm_verify, dev->d_vpts, np);
        kzgamd::msm_run_host(dev->msm_verify, out, sc.data(), np, 2);
    } catch (...) {
        (void)decode_points_status(dev, np);  // nothing of this call stays in flight
        throw;
    }
    const std::vector<int> stat = decode_points_status(dev, np);
    for (size_t i = 0; i < np; ++i) CK_REQUIRE(stat[i] != 1, "Invalid G1 encoding");
    for (size_t i = 0; i < n; ++i) CK_REQUIRE(stat[i] == 0, "Proof is not valid");
    for (size_t i = n; i < n + m; ++i) CK_REQUIRE(stat[i] == 0, "Commitment is not valid");
    if (!have_mono) {
        bool mono_ok = true;
        for (size_t i = n + m; i < np; ++i) mono_ok = mono_ok && stat[i] == 0;
        if (mono_ok) {
            std::lock_guard<std::mutex> lk(dev->mu);
            kzgamd::DeviceGuard on_device(dev->device);
            CK_HIP(on_device.err);
            AffPt* keep = nullptr;
            CK_HIP(hipMalloc(&keep, CELL_SIZE * sizeof(AffPt)));
            if (hipMemcpy(keep, dev->d_vpts + n + m, CELL_SIZE * sizeof(AffPt), hipMemcpyDeviceToDevice) == hipSuccess) dev->d_mono64 = keep;
            else (void)hipFree(keep);
        }
    }
    blst_p2 g2gen, g2s64;
    const kzgamd::pairing::G2Jac gen = kzgamd::pairing::g2_generator();
    memcpy(&g2gen, &gen, sizeof g2gen);
    memcpy(&g2s64, &dev->g2_monomial[CELL_SIZE], sizeof g2s64);
    *ok = kzgamd::pairing::pairings_verify(&out[1], &g2gen, &out[0], &g2s64);
}

// compute_vanishing_polynomial_from_roots (das.rs:493-518)
std::vector<ff::Fr> vanishing_from_roots(const std::vector<ff::Fr>& rts) {
    std::vector<ff::Fr> poly;
    poly.push_back(ff::neg(rts[0]));
    for (size_t i = 1; i < rts.size(); ++i) {
        const ff::Fr nr = ff::neg(rts[i]);
        poly.push_back(ff::add(nr, poly[i - 1]));
        for (size_t j = i - 1; j >= 1; --j) poly[j] = ff::add(ff::mul(poly[j], nr), poly[j - 1]);
        poly[0] = ff::mul(poly[0], nr);
    }
    poly.push_back(ff::Fr::one());
    return poly;
}

// recover_cells_and_kzg_proofs (kzg/src/das.rs:101-205; recover_cells :566-657): the five 8192-point transforms, the
// pointwise products, the coset shifts and the inversions on the GPU; the vanishing polynomial of the <= 64 missing
// cells (65 coefficients) on the host.
void recover_cells(Cell* recovered_cells, KZGProof* recovered_proofs, const uint64_t* cell_indices, const Cell* cells,
                   size_t ncells, const CKZGSettings* cs, KzgAmdSettings* dev) {
    std::vector<ff::Fr> cf;
    CK_REQUIRE(cells_to_limbs(cf, cells, ncells), "Invalid scalar");
    CK_REQUIRE(ncells <= CELLS_PER_EXT_BLOB, "Cell length cannot be larger than CELLS_PER_EXT_BLOB");
    CK_REQUIRE(ncells >= CELLS_PER_EXT_BLOB / 2, "Impossible to recover");
    std::vector<char> have(CELLS_PER_EXT_BLOB, 0);
    std::vector<u32> idx32(ncells);
    for (size_t i = 0; i < ncells; ++i) {
        CK_REQUIRE(cell_indices[i] < CELLS_PER_EXT_BLOB, "Invalid cell index");
        if (i + 1 < ncells) CK_REQUIRE(cell_indices[i + 1] > cell_indices[i], "Indices must be in strictly ascending order");
        have[cell_indices[i]] = 1;
        idx32[i] = (u32)cell_indices[i];
    }
    const ff::Fr* roots = reinterpret_cast<const ff::Fr*>(cs->roots_of_unity);
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    dev->ensure(1);
    dev->ensure_cells(1);
    dev->ensure_recover();
    hipStream_t st = dev->stream;
    const size_t E = 2 * N;
    ff::Fr *A = dev->d_rec[0], *B = dev->d_rec[1], *C = dev->d_rec[2], *D = dev->d_rec[3];
    auto ntt = [&](ff::Fr* out, const ff::Fr* in, int inverse) {
        if (kzgamd_ntt_fr_device(dev->ntt, out, in, E, 1, inverse, st) != 0) throw CkErr{C_KZG_ERROR, "ntt"};
    };
    auto mul = [&](ff::Fr* out, const ff::Fr* a, const ff::Fr* b) {
        hipLaunchKernelGGL(k_fr_mul, dim3((unsigned)(E / 256)), dim3(256), 0, st, out, a, b, E);
    };
    // the provided evaluations in bit-reversed order, missing ones zero
    CK_HIP(hipMemsetAsync(A, 0, E * sizeof(ff::Fr), st));
    CK_HIP(hipMemcpyAsync(dev->d_rec_in, cf.data(), ncells * CELL_SIZE * 32, hipMemcpyHostToDevice, st));
    CK_HIP(hipMemcpyAsync(dev->d_rec_idx, idx32.data(), ncells * sizeof(u32), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rec_scatter, dim3((unsigned)((ncells * CELL_SIZE + 255) / 256)), dim3(256), 0, st, A,
                       (const u32*)dev->d_rec_in, (const u32*)dev->d_rec_idx, ncells, ncells != CELLS_PER_EXT_BLOB);
    std::vector<ff::Fr> vanishing;  // must outlive the copy below
    if (ncells != CELLS_PER_EXT_BLOB) {
        // vanishing_polynomial_for_missing_cells (:520-551): roots w^(64 * brp7(i)) for the missing cells i, short
        // polynomial stretched by 64
        std::vector<ff::Fr> rts;
        for (u32 i = 0; i < CELLS_PER_EXT_BLOB; ++i)
            if (!have[i]) rts.push_back(roots[(size_t)rbl7(i) * CELL_SIZE]);
        const std::vector<ff::Fr> shortp = vanishing_from_roots(rts);
        vanishing.assign(E, ff::Fr::zero());
        for (size_t i = 0; i < shortp.size(); ++i) vanishing[i * CELL_SIZE] = shortp[i];
        CK_HIP(hipMemcpyAsync(B, vanishing.data(), E * sizeof(ff::Fr), hipMemcpyHostToDevice, st));
        ntt(C, B, 0);                      // vanishing_poly_eval
        mul(A, A, C);                      // extended_evaluation_times_zero
        ntt(D, A, 1);                      // ..._coeffs
        mul(D, D, dev->d_pow7);            // coset_fft: shift_poly by 7, then the transform
        ntt(A, D, 0);                      // extended_evaluations_over_coset
        mul(B, B, dev->d_pow7);
        ntt(C, B, 0);                      // vanishing_poly_over_coset
        hipLaunchKernelGGL(k_fr_inverse, dim3((unsigned)(E / 64)), dim3(64), 0, st, C, E);
        mul(A, A, C);
        ntt(D, A, 1);                      // coset_ifft: the transform, then shift_poly by 1/7
        mul(D, D, dev->d_pow7inv);         // reconstructed_poly_coeff
        ntt(A, D, 0);                      // its 8192 evaluations, natural order
        hipLaunchKernelGGL(k_cells_out, dim3((unsigned)(E / 256)), dim3(256), 0, st, reinterpret_cast<u32*>(dev->d_fr_ext),
                           (const ff::Fr*)A, (size_t)1);
        CK_HIP(hipMemcpyAsync(recovered_cells, dev->d_fr_ext, E * 32, hipMemcpyDeviceToHost, st));
    } else {
        memcpy(recovered_cells, cells, E * 32);
        if (recovered_proofs) ntt(D, A, 1);  // poly_lagrange_to_monomial of the given cells (:186-188)
    }
    if (recovered_proofs) {
        // compute_fk20_proofs reads the first 4096 coefficients (:190-200, toeplitz_coeffs_stride :659-688)
        if (!dev->d_roots8192) {
            CK_HIP(hipMalloc(&dev->d_roots8192, (2 * N + 1) * sizeof(ff::Fr)));
            CK_HIP(hipMemcpy(dev->d_roots8192, cs->roots_of_unity, (2 * N + 1) * sizeof(ff::Fr), hipMemcpyHostToDevice));
        }
        if (!dev->msm_monomial) dev->msm_monomial = kzgamd::msm_create(dev->d_monomial, N, true, true, true, kzgamd::G1_TRUSTED, &dev->opt);
        dev->ensure_q(1);
        CK_HIP(hipMemcpyAsync(dev->d_fr_b, D, N * sizeof(ff::Fr), hipMemcpyDeviceToDevice, st));
        enqueue_cell_proofs(dev, 1, st, false);
        CK_HIP(hipMemcpyAsync(recovered_proofs, dev->d_proofs, CELLS_PER_EXT_BLOB * 48, hipMemcpyDeviceToHost, st));
    }
    CK_HIP(hipStreamSynchronize(st));
}

}  // namespace

// c_bindings.rs:290-355 -> DAS::verify_cell_kzg_proof_batch (kzg/src/das.rs:294-389)
extern "C" C_KZG_RET verify_cell_kzg_proof_batch(bool* ok, const Bytes48* commitments_bytes, const uint64_t* cell_indices,
                                                 const Cell* cells, const Bytes48* proofs_bytes, uint64_t num_cells,
                                                 const CKZGSettings* s) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (num_cells == 0) {
        *ok = true;
        return C_KZG_OK;
    }
    if (!commitments_bytes || !cell_indices || !cells || !proofs_bytes) return C_KZG_BADARGS;
    return guarded([&] { verify_cells(ok, commitments_bytes, cell_indices, cells, proofs_bytes, (size_t)num_cells, s, dev); });
}

// c_bindings.rs:202-289 -> DAS::recover_cells_and_kzg_proofs (kzg/src/das.rs:101-205); recovered_proofs may be NULL
extern "C" C_KZG_RET recover_cells_and_kzg_proofs(Cell* recovered_cells, KZGProof* recovered_proofs, const uint64_t* cell_indices,
                                                  const Cell* cells, uint64_t num_cells, const CKZGSettings* s) {
    if (!recovered_cells) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (num_cells && (!cell_indices || !cells)) return C_KZG_BADARGS;
    return guarded([&] { recover_cells(recovered_cells, recovered_proofs, cell_indices, cells, (size_t)num_cells, s, dev); });
}

// blst/src/eip_7594.rs:35-97: the Fiat-Shamir scalar of a cell batch (no settings: the inputs are only parsed —
// FsG1::from_bytes accepts any curve point, blst/src/types/g1.rs:65-87 — and hashed)
extern "C" C_KZG_RET compute_verify_cell_kzg_proof_batch_challenge(blst_fr* challenge_out, const Bytes48* commitment_bytes,
                                                                   uint64_t num_commitments, const uint64_t* commitment_indices,
                                                                   const uint64_t* cell_indices, const Cell* cells,
                                                                   const Bytes48* proofs_bytes, uint64_t num_cells) {
    if (!challenge_out) return C_KZG_BADARGS;
    memset(challenge_out, 0, sizeof *challenge_out);
    if ((num_commitments && !commitment_bytes) || (num_cells && (!commitment_indices || !cell_indices || !cells || !proofs_bytes)))
        return C_KZG_BADARGS;
    return guarded([&] {
        for (size_t i = 0; i < num_commitments; ++i) {
            blst_p1 t;
            CK_REQUIRE(kzgamd::host_p1_uncompress(&t, commitment_bytes[i].bytes), "Invalid commitment");
        }
        std::vector<ff::Fr> cf;
        CK_REQUIRE(cells_to_limbs(cf, cells, (size_t)num_cells), "Invalid scalar");
        for (size_t i = 0; i < num_cells; ++i) {
            blst_p1 t;
            CK_REQUIRE(kzgamd::host_p1_uncompress(&t, proofs_bytes[i].bytes), "Invalid proof");
        }
        const ff::Fr r = cell_batch_challenge(commitment_bytes, (size_t)num_commitments, commitment_indices, cell_indices, cells,
                                              proofs_bytes, (size_t)num_cells);
        memcpy(challenge_out, &r, sizeof r);
    });
}

// ---- host-only helpers over blst_p2 / the pairing (no GPU needed): what a binding test-suite or a caller that
// wants to finish kzgamd_verify_*_g1 itself uses.  pairings_verify = blst/src/kzg_proofs.rs:73-100.
extern "C" int kzgamd_pairings_verify(const blst_p1* a1, const blst_p2* a2, const blst_p1* b1, const blst_p2* b2) {
    if (!a1 || !a2 || !b1 || !b2) return -1;
    return kzgamd::pairing::pairings_verify(a1, a2, b1, b2) ? 1 : 0;
}
extern "C" int kzgamd_p2_uncompress(blst_p2* out, const uint8_t in[96]) {
    kzgamd::pairing::G2Jac p;
    if (!out || !in || !kzgamd::pairing::g2_uncompress(p, in)) return 1;
    memcpy(out, &p, sizeof p);
    return 0;
}
extern "C" void kzgamd_p2_compress(uint8_t out[96], const blst_p2* in) {
    kzgamd::pairing::G2Jac p;
    memcpy(&p, in, sizeof p);
    kzgamd::pairing::g2_compress(out, p);
}
extern "C" void kzgamd_p2_generator(blst_p2* out) {
    const kzgamd::pairing::G2Jac g = kzgamd::pairing::g2_generator();
    memcpy(out, &g, sizeof g);
}
extern "C" void kzgamd_p2_mult(blst_p2* out, const blst_p2* in, const blst_fr* scalar_mont) {
    kzgamd::pairing::G2Jac p;
    memcpy(&p, in, sizeof p);
    ff::Fr k;
    memcpy(&k, scalar_mont, 32);
    k = ff::from_mont(k);
    const kzgamd::pairing::G2Jac r = kzgamd::pairing::g2_mul(p, k.v);
    memcpy(out, &r, sizeof r);
}
extern "C" void kzgamd_p2_add(blst_p2* out, const blst_p2* a, const blst_p2* b) {
    kzgamd::pairing::G2Jac x, y;
    memcpy(&x, a, sizeof x);
    memcpy(&y, b, sizeof y);
    const kzgamd::pairing::G2Jac r = kzgamd::pairing::g2_add(x, y);
    memcpy(out, &r, sizeof r);
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

// kzg/src/eth/c_bindings.rs:356-372 (EIP-7594).  cells or proofs may be NULL, not both (das.rs:250-252).
extern "C" C_KZG_RET compute_cells_and_kzg_proofs(Cell* cells, KZGProof* proofs, const Blob* blob, const CKZGSettings* s) {
    if (!blob || (!cells && !proofs)) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    return guarded([&] { cells_and_proofs(cells ? cells->bytes : nullptr, proofs, blob, 1, s, dev); });
}

extern "C" C_KZG_RET kzgamd_compute_cells_and_kzg_proofs_batch(Cell* cells, KZGProof* proofs, const Blob* blobs, size_t n,
                                                               const CKZGSettings* s) {
    if (!blobs || (!cells && !proofs)) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    return guarded([&] { cells_and_proofs(cells ? cells->bytes : nullptr, proofs, blobs, n, s, dev); });
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
