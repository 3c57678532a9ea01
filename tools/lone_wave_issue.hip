// How fast does ONE wave issue VALU instructions?  Streams of dependent / independent v_add_u32, v_mad_u64_u32 and DPP
// moves in a lone wave (1 workgroup of 64 lanes) and in W waves per SIMD; cycles per instruction from s_memrealtime-free
// wall clock (HIP events, nominal 2.4 GHz).   hipcc --offload-arch=gfx950 -O3 tools/lone_wave_issue.hip -o tools/lone_wave_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32;
typedef unsigned long long u64;
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(64) k(u32* out, int iters) {
    u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    u64 m0 = a0, m1 = a1, m2 = a2, m3 = a3;
    const u32 c = out[0] | 3u;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 64 dependent adds
            REP8(REP8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(c));))
        } else if (MODE == 1) {  // 64 adds on 8 independent chains
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 2) {  // 64 dependent mad64 (one s_nop between, as the hazard wants)
            REP8(REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0\n s_nop 0" : "+v"(m0) : "v"(c) : "vcc");))
        } else if (MODE == 3) {  // 64 mad64 on 4 independent chains
            REP8(REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %4, %4, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3"
                                   : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(c) : "vcc");))
        } else if (MODE == 6) {  // 64 mad64 on 2 interleaved chains: no hazard, no s_nop
            REP8(REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %2, %0\n v_mad_u64_u32 %1, vcc, %2, %2, %1" : "+v"(m0), "+v"(m1) : "v"(c) : "vcc");))
        } else if (MODE == 7) {  // 2 interleaved chains, with two plain instructions per four multiply-adds
            REP8(REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n"
                                   "v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(m0), "+v"(m1), "+v"(a1), "+v"(a2) : "v"(c) : "vcc");))
        } else if (MODE == 4) {  // 64 dependent DPP moves (row_shl:1), two wait states each
            REP8(REP8(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0));))
        } else if (MODE == 5) {  // dependent: add -> dpp -> add -> dpp ...
            REP8(REP8(asm volatile("v_add_u32 %0, %0, %1\n s_nop 1\n v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0) : "v"(c));))
        }
    }
    out[threadIdx.x + 64 * blockIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (u32)(m0 + m1 + m2 + m3);
}
int main() {
    u32* d;
    hipMalloc(&d, 1 << 22);
    hipMemset(d, 0, 1 << 22);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 20000;
    const char* names[8] = {"v_add_u32 dependent", "v_add_u32 8 chains", "v_mad_u64_u32 dependent (+s_nop 0)", "v_mad_u64_u32 4 chains", "v_mov_dpp dependent (+s_nop 1)", "add -> dpp chain", "v_mad_u64_u32 2 interleaved chains", "2 chains x 2 mads + 2 adds"};
    const int per[8] = {64, 64, 64, 256, 64, 128, 128, 384};
    for (int waves : {1, 1024, 2048, 3072, 4096}) {
        for (int mode = 0; mode < 8; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, d, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, d, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, d, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(waves), dim3(64), 0, 0, d, iters); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(waves), dim3(64), 0, 0, d, iters); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(waves), dim3(64), 0, 0, d, iters); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(waves), dim3(64), 0, 0, d, iters); break;
                    case 7: hipLaunchKernelGGL(k<7>, dim3(waves), dim3(64), 0, 0, d, iters); break;
                }
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("%4d wave(s): %-36s %.2f cycles per VALU instruction per wave (2.4 GHz nominal)\n", waves, names[mode], best * 1e-3 * 2.4e9 / ((double)iters * per[mode]));
        }
    }
    return 0;
}
