// Timing and bit-identity check of the limb-parallel chain arithmetic (fpw.hip.h / g1w.hip.h): one wave runs K dependent
// doublings (g1w::dbl_k) and K dependent additions (g1w::dadd of a fixed second operand), HIP-event time per operation
// and a checksum of the results.  Built twice — as is and with -DKZGAMD_WMUL_DIGIT_AHEAD (fpw.hip.h: the quotient digit taken off the first multiply-add) — the
// two must print the same checksums:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wmul_bench.hip -o tools/wmul_bench [-DKZGAMD_WMUL_DIGIT_AHEAD]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../rust-kzg_amd/csrc/g1w.hip.h"
using g1::Xyzz;
using ff::u32;

__global__ void __launch_bounds__(64) k_dbl_chain(const Xyzz* __restrict__ in, Xyzz* __restrict__ out, int k) {
    const int lane = threadIdx.x;
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt acc = g1w::load(in + blockIdx.x, lane);
    g1w::dbl_k(acc, k, lc, lane);
    g1w::store(out + blockIdx.x, acc, lc, lane);
}
__global__ void __launch_bounds__(64) k_add_chain(const Xyzz* __restrict__ in, Xyzz* __restrict__ out, int k) {
    __shared__ u32 sh[16];
    const int lane = threadIdx.x;
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt acc = g1w::load(in + blockIdx.x, lane);
    const g1w::WPt b = g1w::load(in + gridDim.x + blockIdx.x, lane);
    for (int i = 0; i < k; ++i) g1w::dadd(acc, b, lc, sh, lane);
    g1w::store(out + blockIdx.x, acc, lc, lane);
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2000, W = argc > 2 ? atoi(argv[2]) : 1;
    Xyzz *d_in, *d_out;
    const size_t n = 2 * (size_t)W;
    Xyzz* h = (Xyzz*)malloc(n * sizeof(Xyzz));
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < n * sizeof(Xyzz) / 4; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int limb = (int)(i % 14);
        ((u32*)h)[i] = (u32)s & (limb == 13 ? 0xffffu : 0xfffffffu);
    }
    hipMalloc(&d_in, n * sizeof(Xyzz));
    hipMalloc(&d_out, n * sizeof(Xyzz));
    hipMemcpy(d_in, h, n * sizeof(Xyzz), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int which = 0; which < 2; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            if (which == 0) hipLaunchKernelGGL(k_dbl_chain, dim3(W), dim3(64), 0, 0, d_in, d_out, K);
            else hipLaunchKernelGGL(k_add_chain, dim3(W), dim3(64), 0, 0, d_in, d_out, K);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        hipMemcpy(h, d_out, W * sizeof(Xyzz), hipMemcpyDeviceToHost);
        uint64_t sum = 0;
        for (size_t i = 0; i < W * sizeof(Xyzz) / 4; ++i) sum = sum * 1099511628211ull + ((u32*)h)[i];
        printf("%s x %d on %d wave(s): %.3f us per operation, checksum %016llx\n", which == 0 ? "doubling" : "addition", K, W, best * 1e3 / K,
               (unsigned long long)sum);
    }
    return 0;
}
