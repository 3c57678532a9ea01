// Shared between ntt.hip (Fr transforms) and fftg1.hip (G1-valued transforms): the context object
// behind the opaque handle of kzgamd_ntt_new().
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#include "ff.hip.h"

struct NttErr {
    hipError_t e;
};
#define NTT_TRY(x)                              \
    do {                                        \
        hipError_t _e = (x);                    \
        if (_e != hipSuccess) throw NttErr{_e}; \
    } while (0)

// device copy of one tile plan (ntt_plan.h): the per-thread element table of every round + per-round constants
struct NttPlanDev {
    void* d_tab = nullptr;
    int nrounds = 0;
    unsigned rd[12][6];  // element bit, its LDS position bit, pos, M, barrier_after, flags (fused DAS plans)
};

struct NttCtx {
    using Fr = ff::Fr;
    std::map<int, NttPlanDev> plans;  // key = kind * 16 + T (kind 3: fused DAS plans); filled by kzgamd_ntt_new, read-only afterwards
    int device = 0;
    unsigned scale = 0;
    size_t W = 0;
    void* d_roots = nullptr;  // W + 1 twiddles in the 2^261 domain as 9 x 29-bit limbs (Fr transforms), natural order
    void *d_tw_fwd = nullptr, *d_tw_inv = nullptr;  // the same, stage-major (forward roots / inverse roots)
    std::vector<Fr> roots;  // host copy, blst Montgomery form
    hipStream_t stream = nullptr;
    std::mutex mu;
    Fr *d_a = nullptr, *d_b = nullptr;
    size_t cap = 0;
    // G1-valued transforms (fftg1.hip): the roots as GLV-split scalars, staging and work buffers
    void* d_kroots = nullptr;  // (W + 1) x RootSplit
    void *d_p1 = nullptr, *d_pts = nullptr, *d_tab = nullptr;
    size_t cap_g1 = 0, cap_tab = 0;
    // G1 stages (fftg1.hip) by their number of half-butterflies: up to g1_wide_max a wave each (limb-parallel), up to
    // g1_quad_max four lanes each, up to g1_pair_max two lanes each, one lane each above
    // (tuning keys g1_wide_max / g1_quad_max / g1_pair_max, read at creation; 0 disables a form)
    size_t g1_wide_max = 4096, g1_quad_max = 16384, g1_pair_max = 32768;

    // Combining of concurrent host-buffer calls (ntt_fr, das_fft_extension) on ONE handle — the reference shares its
    // FFTSettings by reference between rayon workers.  As for the prepared MSM handle (msm.hip): a caller copies its input
    // into a page-locked slot on its own thread and queues a request; one caller at a time per lane takes everything
    // queued of the same kind and length (up to COMB_MAX requests) and runs it as ONE batched launch: a kernel gathers the
    // inputs from the slots over PCIe, the batched transform runs, a kernel writes the results back into the slots; every
    // caller copies its result out of its slot.  A 4096-point transform is 4 us of GPU under ~60 us of copies, launches
    // and a synchronisation: per batch instead of per call.  Lists longer than COMB_NMAX take the plain path.
    struct HostCall {
        void* out;
        const void* in;
        size_t n;
        int kind;  // 0 forward, 1 inverse, 2 DAS extension
        unsigned char* slot = nullptr;
        bool done = false;
        int rc = 0;
    };
    static constexpr size_t COMB_MAX = 32, COMB_NMAX = 8192;
    static constexpr int COMB_SLOTS = 40, COMB_LANES = 2;
    struct CombLane {
        hipStream_t st = nullptr;
        Fr *d_in = nullptr, *d_out = nullptr, *d_tmp = nullptr;  // COMB_MAX x COMB_NMAX elements each
        bool busy = false;
    };
    struct Combine {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<HostCall*> pending;
        int leaders = 0;
        CombLane lanes[COMB_LANES];
        // page-locked, mapped staging slots, allocated in chunks as callers need them (4, 4, 8, ...): see msm.hip's pool
        std::vector<unsigned char*> slot_chunks;
        int slots_allocated = 0;
        size_t slot_bytes = 0;
        bool pinned_failed = false;
        std::vector<unsigned char*> free_slots;
    } comb;
    bool combine = true;  // tuning key combine

    ~NttCtx() {
        for (auto* c : comb.slot_chunks) (void)hipHostFree(c);
        for (auto& l : comb.lanes) {
            if (l.d_in) (void)hipFree(l.d_in);
            if (l.d_out) (void)hipFree(l.d_out);
            if (l.d_tmp) (void)hipFree(l.d_tmp);
            if (l.st) (void)hipStreamDestroy(l.st);
        }
        if (d_roots) (void)hipFree(d_roots);
        if (d_tw_fwd) (void)hipFree(d_tw_fwd);
        if (d_tw_inv) (void)hipFree(d_tw_inv);
        if (d_a) (void)hipFree(d_a);
        if (d_b) (void)hipFree(d_b);
        if (d_kroots) (void)hipFree(d_kroots);
        if (d_p1) (void)hipFree(d_p1);
        if (d_pts) (void)hipFree(d_pts);
        if (d_tab) (void)hipFree(d_tab);
        if (stream) (void)hipStreamDestroy(stream);
        for (auto& kv : plans)
            if (kv.second.d_tab) (void)hipFree(kv.second.d_tab);
    }
    void ensure(size_t n) {
        if (n <= cap) return;
        if (d_a) (void)hipFree(d_a);
        if (d_b) (void)hipFree(d_b);
        d_a = d_b = nullptr;
        cap = 0;
        NTT_TRY(hipMalloc(&d_a, n * sizeof(Fr)));
        NTT_TRY(hipMalloc(&d_b, n * sizeof(Fr)));
        cap = n;
    }
};

namespace kzgamd {
struct Options;
// ntt.hip: what kzgamd_ntt_new_ex calls once the configuration is resolved (config.h); NULL on failure
void* ntt_create(unsigned scale, const Options& opt);
// fftg1.hip: G1 transforms of device-resident g1::Xyzz data, see there
void* fftg1_device(NttCtx* ctx, void* data_xyzz, void* scratch_xyzz, size_t n, size_t nbatch, int inverse, hipStream_t st,
                   bool scale_inverse = true);
}  // namespace kzgamd
