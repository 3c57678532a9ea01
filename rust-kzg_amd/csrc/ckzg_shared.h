// Shared internals of the c-kzg layer (ckzg.hip: settings object, lanes, EIP-4844 proving; ckzg_verify.hip: the verify_*
// entry points and the host pairing exports; ckzg_7594.hip: cells, FK20, recovery, cell verification).  Types and
// templates every one of them needs, and the functions one of them defines for the others (namespace ckz).
#pragma once
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


// ---- errors (thrown across the three translation units, caught by guarded())
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


// ---- Fr helpers for the proving kernel (Montgomery, 8 x u32) ----
static __device__ __forceinline__ ff::Fr fr_load_be(const u32* __restrict__ w8, bool* ok) {
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
static __device__ ff::Fr fr_inverse(const ff::Fr& a) { return ff::inverse_bgcd(a); }
// ff::mul on blst_fr values through the 29-bit multiplier of the NTT (fr29::mul_blst: the same result in about half
// the instructions); the quotient kernels below are a stream of such products
static __device__ __forceinline__ ff::Fr fmul(const ff::Fr& a, const ff::Fr& b) { return fr29::mul_blst(a, b); }


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

// decode + membership test of np compressed points: up to WIDE_CHECK_MAX points the test runs one wave per point
constexpr size_t WIDE_CHECK_MAX = 4096;
constexpr size_t WIDE_COMMIT_CHECK_MAX = 512;  // ... the commitments of a proof batch: up to this many

// see ckzg.hip
// batches up to this size leave the device as Jacobian points and are compressed on the host with one inversion
// (host_p1_compress_batch): k_final's one-lane inversion is 0.25 ms of latency, worth paying only when a block of 64
// points shares it
constexpr size_t HOST_COMPRESS_MAX = 16;
constexpr size_t HOST_CHECK_MAX = 64;  // commitments of a proof batch validated on the host's cores up to this many


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
    int cfg_sha_lanes = 0;  // lanes per blob of the device Fiat-Shamir hash (tuning key sha_lanes; 0 = by batch size)
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
        cfg_sha_lanes = (int)o.t[T_SHA_LANES];
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

namespace ckz {
// ---- defined in ckzg.hip
KzgAmdSettings* lookup(const CKZGSettings* s);
KzgAmdSettings* make_lane(KzgAmdSettings* parent);
// decode + membership test of np compressed points (k_decode_g1_wide / k_affpts_in_g1_wide or the single-lane kernel)
void decode_check_enqueue(g1::AffPt* d_pts, int* d_stat, const unsigned char* d_bytes, size_t np, hipStream_t st, bool wide,
                          hipEvent_t decoded = nullptr);
void compress_on_host(uint8_t* out48, const blst_p1* jac, size_t n);
bool host_blob_valid(const uint8_t* blob);
void prove_batch(KZGProof* proofs, Bytes32* ys, const Blob* blobs, const Bytes32* zs, const Bytes48* commitments, size_t n,
                 KzgAmdSettings* dev, Bytes32* zs_out = nullptr, bool commitments_checked_elsewhere = false);
// ---- defined in ckzg_verify.hip
void verify_g1_begin(const Bytes48* commitments, const Bytes48* proofs, size_t n, KzgAmdSettings* dev);
void verify_g1_finish(blst_p1* proof_lincomb, blst_p1* rhs, const Bytes48* commitments, const Bytes32* zs, const Bytes32* ys, const Bytes48* proofs, size_t n, KzgAmdSettings* dev);
}  // namespace ckz
using namespace ckz;

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

inline size_t reverse_bits(size_t v, unsigned bits) {
    size_t r = 0;
    for (unsigned b = 0; b < bits; ++b)
        if (v & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
    return r;
}

inline bool fr_from_be32_checked(ff::Fr& out, const uint8_t* in) {  // FsFr::from_bytes: canonical limbs, false if >= r
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
