// Micro-benchmarks that decide how the field multiplier is written:
//  (1) issue rate of the integer instructions a wide multiply can be built from
//  (2) throughput + self-consistency of the candidate Montgomery multipliers
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ffbench.hip -o tools/ffbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../rust-kzg_amd/csrc/ff.hip.h"
#include "../rust-kzg_amd/csrc/ff28.hip.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

using namespace ff;

// ---------------------------------------------------------------- instruction rates
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) instr_kernel(unsigned* out, int iters, unsigned seed, unsigned long long* cyc) {
    unsigned a = seed * (threadIdx.x + 1) | 1, b = seed ^ (blockIdx.x * 77 + 5);
    unsigned long long acc[8];
    unsigned x[8];
    double d[8];
    for (int k = 0; k < 8; ++k) {
        acc[k] = a * (k + 3);
        x[k] = b * (k + 7);
        d[k] = (double)(a + k);
    }
    double da = (double)a * 1e-9, db = (double)b * 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if (OP == 1) {
#define X(k) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(x[k]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 2) {
#define X(k) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(x[k]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 3) {
#define X(k) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 4) {
#define X(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "v"(da), "v"(db));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 5) {
#define X(k) asm volatile("v_add_co_u32 %0, vcc, %1, %0\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc" : "+v"(x[k]), "+v"(a) : "v"(b) : "vcc");
            REP8(X)
#undef X
        } else if (OP == 6) {
#define X(k) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[k]));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 7) {
#define X(k) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 8) {
#define X(k) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(x[k]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 9) {
#define X(k) asm volatile("v_alignbit_b32 %0, %1, %0, 28" : "+v"(x[k]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 10) {  // dependent chain of mad_u64_u32 on one accumulator
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if (OP == 11) {  // 64-bit add via add_co/addc pair on 8 accumulators
#define X(k) asm volatile("v_add_co_u32 %0, vcc, %1, %0\n\tv_addc_co_u32 %2, vcc, %3, %2, vcc" : "+v"(x[k]), "+v"(a) : "v"(b), "v"(seed) : "vcc");
            REP8(X)
#undef X
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned r = a;
    for (int k = 0; k < 8; ++k) r ^= (unsigned)acc[k] ^ (unsigned)(acc[k] >> 32) ^ x[k] ^ (unsigned)d[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 7 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
static void run_instr(const char* name, int per_iter) {
    unsigned* out;
    const int blocks = 256 * 8, threads = 256, iters = 40000;
    CK(hipMalloc(&out, blocks * threads * 4));
    unsigned long long* cyc;
    CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    instr_kernel<OP><<<blocks, threads>>>(out, 10, 12345, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    instr_kernel<OP><<<blocks, threads>>>(out, iters, 12345, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double winstr = (double)blocks * (threads / 64) * iters * per_iter;  // wave-instructions
    double per_simd_per_s = winstr / (256.0 * 4) / (ms * 1e-3);
    // cycles per wave-instruction per SIMD at the reported clock
    unsigned long long hc;
    CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    // 8 waves resident per SIMD (2048 blocks x 4 waves over 1024 SIMDs, all co-resident)
    printf("%-28s %8.3f ms  => %5.2f cyc/wave-instr @2.4GHz wall;  s_memtime: %.2f ticks/instr/wave (8 waves/SIMD => %.2f per issue), eff clock %.2f GHz\n",
           name, ms, 2.4e9 / per_simd_per_s, (double)hc / ((double)iters * per_iter), (double)hc / ((double)iters * per_iter) / 8.0,
           (double)hc / (ms * 1e-3) * 1e-9);
    CK(hipFree(out));
}

// ---------------------------------------------------------------- multiplier throughput
template <int V>
__global__ void __launch_bounds__(256) mul_kernel(const Fp* in, Fp* out, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[2 * tid], y = in[2 * tid + 1];
    if (V == 0) {
        for (int i = 0; i < iters; ++i) {
            x = mul(x, y);
            y = mul(y, x);
        }
        out[2 * tid] = x;
        out[2 * tid + 1] = y;
    } else if (V == 1) {
        ff28::Fp28 a = ff28::from_sat(x), b = ff28::from_sat(y);
        for (int i = 0; i < iters; ++i) {
            a = ff28::mul(a, b);
            b = ff28::mul(b, a);
        }
        out[2 * tid] = ff28::to_sat(a);
        out[2 * tid + 1] = ff28::to_sat(b);
    } else if (V == 2) {  // add/sub mix on the saturated form
        for (int i = 0; i < iters; ++i) {
            x = add(x, y);
            y = sub(y, x);
        }
        out[2 * tid] = x;
        out[2 * tid + 1] = y;
    }
}

static void host_ref(const Fp* in, Fp* out, int n, int iters, int V) {
    for (int t = 0; t < n; ++t) {
        Fp x = in[2 * t], y = in[2 * t + 1];
        if (V == 1) {
            ff28::Fp28 a = ff28::from_sat(x), b = ff28::from_sat(y);
            for (int i = 0; i < iters; ++i) {
                a = ff28::mul(a, b);
                b = ff28::mul(b, a);
            }
            out[2 * t] = ff28::to_sat(a);
            out[2 * t + 1] = ff28::to_sat(b);
            continue;
        }
        for (int i = 0; i < iters; ++i) {
            if (V == 2) {
                x = add(x, y);
                y = sub(y, x);
            } else {
                x = mul(x, y);
                y = mul(y, x);
            }
        }
        out[2 * t] = x;
        out[2 * t + 1] = y;
    }
}

template <int V>
static void run_mul(const char* name) {
    const int blocks = 256 * 8, threads = 256, n = blocks * threads, iters = 200;
    std::vector<Fp> h(2 * n), ho(2 * n), ref(2 * 64);
    unsigned long long s = 88172645463325252ull;
    for (auto& f : h) {
        for (int i = 0; i < 12; ++i) {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            f.v[i] = (unsigned)s;
        }
        f.v[11] &= 0x0fffffff;  // < p
    }
    Fp *din, *dout;
    CK(hipMalloc(&din, 2 * n * sizeof(Fp)));
    CK(hipMalloc(&dout, 2 * n * sizeof(Fp)));
    CK(hipMemcpy(din, h.data(), 2 * n * sizeof(Fp), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    mul_kernel<V><<<blocks, threads>>>(din, dout, 4);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    mul_kernel<V><<<blocks, threads>>>(din, dout, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(ho.data(), dout, 2 * n * sizeof(Fp), hipMemcpyDeviceToHost));
    host_ref(h.data(), ref.data(), 64, iters, V);
    int bad = 0;
    for (int i = 0; i < 128; ++i)
        if (ref[i] != ho[i]) ++bad;
    double ops = (double)n * iters * 2;
    printf("%-28s %8.3f ms  %8.2f G op/s   mismatches(vs host, 128 samples)=%d\n", name, ms, ops / (ms * 1e-3) * 1e-9,
           bad);
    CK(hipFree(din));
    CK(hipFree(dout));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    run_instr<0>("v_mad_u64_u32 (8 indep)", 16);
    run_instr<10>("v_mad_u64_u32 (1 chain)", 16);
    run_instr<1>("v_mul_lo_u32", 16);
    run_instr<2>("v_mul_hi_u32", 16);
    run_instr<3>("v_mad_u32_u24", 16);
    run_instr<8>("v_mul_hi_u32_u24", 16);
    run_instr<4>("v_fma_f64", 16);
    run_instr<5>("v_add_co+v_addc pair", 16);
    run_instr<11>("v_add_co+v_addc pair (b)", 16);
    run_instr<6>("v_lshrrev_b64", 16);
    run_instr<7>("v_add3_u32", 16);
    run_instr<9>("v_alignbit_b32", 16);
    run_mul<0>("fp mul 12x32 CIOS");
    run_mul<1>("fp mul 14x28 comba");
    run_mul<2>("fp add+sub 12x32");
    return 0;
}
