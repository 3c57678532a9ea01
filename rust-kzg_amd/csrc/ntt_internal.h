// Shared between ntt.hip (Fr transforms) and fftg1.hip (G1-valued transforms): the context object
// behind the opaque handle of kzgamd_ntt_new().
#pragma once
#include <hip/hip_runtime.h>
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

    ~NttCtx() {
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
