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
        } else if (OP == 12) {
#define X(k) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[k]) : "v"(da));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 13) {
#define X(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(acc[(k + 1) & 7]));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 14) {
            float fa = (float)a, fb = 1.0f + (float)b * 1e-9f;
#define X(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[k]) : "v"(fa), "v"(fb));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 15) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[k]) : "v"(acc[(k + 3) & 7]));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 16) {
#define X(k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[k]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if (OP == 18) {
#define X(k) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if (OP == 17) {
            // the double-precision-FMA limb product ("DPFP", Emmart et al.): a 52 x 52-bit product as
            //   hi = fma_rz(a, b, 2^104); t = (2^104 + 2^52) - hi; lo = fma_rz(a, b, t)
            // whose mantissas are the two 52-bit halves, accumulated as integers: 5 instructions per product.
            // (round-toward-zero is the wave's DP rounding mode here; see dpfp_check for exactness)
            const double C1 = 0x1.0p104, C2 = 0x1.0p104 + 0x1.0p52;
#define X(k)                                                                                                  \
    {                                                                                                         \
        double hi_, t_, lo_;                                                                                  \
        asm volatile("v_fma_f64 %0, %5, %6, %7\n\tv_add_f64 %1, %8, -%0\n\tv_fma_f64 %2, %5, %6, %1\n\t"     \
                     "v_lshl_add_u64 %3, %0, 0, %3\n\tv_lshl_add_u64 %4, %2, 0, %4"                            \
                     : "=&v"(hi_), "=&v"(t_), "=&v"(lo_), "+v"(acc[k]), "+v"(acc[(k + 4) & 7])                \
                     : "v"(d[k]), "v"(d[(k + 1) & 7]), "v"(C1), "v"(C2));                                     \
    }
            REP8(X)
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

// exactness of the DPFP product: one 52 x 52-bit product per thread against the host's 128-bit integer product
__global__ void dpfp_check(const unsigned long long* a, const unsigned long long* b, unsigned long long* hi, unsigned long long* lo) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // MODE[3:2] (double / half rounding) = 3: toward zero
    const double x = (double)a[t], y = (double)b[t];
    const double C1 = 0x1.0p104, C2 = 0x1.0p104 + 0x1.0p52;
    double h, tt, l;
    // (the mode change is part of the same asm block: s_setreg needs wait states before a VALU instruction sees it,
    // which the compiler cannot insert around opaque asm)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\ts_nop 7\n\t"
                 "v_fma_f64 %0, %3, %4, %5\n\tv_add_f64 %1, %6, -%0\n\tv_fma_f64 %2, %3, %4, %1\n\t"
                 "s_nop 7\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\ts_nop 7"
                 : "=&v"(h), "=&v"(tt), "=&v"(l)
                 : "v"(x), "v"(y), "v"(C1), "v"(C2));
    hi[t] = (unsigned long long)__double_as_longlong(h) & 0xfffffffffffffull;
    lo[t] = (unsigned long long)__double_as_longlong(l) & 0xfffffffffffffull;
}
static void run_dpfp_check() {
    const int n = 1 << 16;
    std::vector<unsigned long long> a(n), b(n), hi(n), lo(n);
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = s & 0xfffffffffffffull;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; b[i] = s & 0xfffffffffffffull;
    }
    a[0] = b[0] = 0xfffffffffffffull;
    a[1] = 0; b[2] = 1;
    unsigned long long *da, *db, *dh, *dl;
    CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dh, n * 8)); CK(hipMalloc(&dl, n * 8));
    CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    dpfp_check<<<n / 256, 256>>>(da, db, dh, dl);
    CK(hipMemcpy(hi.data(), dh, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(lo.data(), dl, n * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned __int128 p = (unsigned __int128)a[i] * b[i];
        if (hi[i] != (unsigned long long)(p >> 52) || lo[i] != ((unsigned long long)p & 0xfffffffffffffull)) ++bad;
    }
    printf("DPFP 52x52 product (2 x v_fma_f64 + v_add_f64, round toward zero): %d products, %d mismatches vs 128-bit integers\n", n, bad);
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dh)); CK(hipFree(dl));
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
__device__ __host__ inline void chain_step(unsigned long long& acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(acc));
#endif
}
// ff28::mul on ONE accumulator chain: the empty asm after every multiply-add keeps the compiler from re-associating
// a column into several chains (each merge is a 64-bit addition, 4.4 cycles)
__device__ __host__ inline ff28::Fp28 mul28_signed(const ff28::Fp28& a, const ff28::Fp28& b) {
    using namespace ff28;
    u32 m[L];
    Fp28 r;
    unsigned long long acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) { acc += (unsigned long long)a.v[i] * b.v[k - i]; chain_step(acc); }
#pragma unroll
        for (int i = 0; i < k; ++i) { acc += (unsigned long long)m[i] * p28(k - i); chain_step(acc); }
        m[k] = ((u32)acc * P0INV) & MASK;
        acc += (unsigned long long)m[k] * p28(0);
        chain_step(acc);
        acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) { acc += (unsigned long long)a.v[i] * b.v[k - i]; chain_step(acc); }
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) { acc += (unsigned long long)m[i] * p28(k - i); chain_step(acc); }
        r.v[k - L] = (u32)acc & MASK;
        acc >>= 28;
    }
    r.v[L - 1] = (u32)acc;
    return r;
}

// One Karatsuba level on the a*b half of the Montgomery product (round 4): a = a0 + a1 * 2^196 (7 + 7 limbs),
//   a*b = z0 + (zs - z0 - z2) * 2^196 + z2 * 2^392,  z0 = a0*b0, z2 = a1*b1, zs = (a0 + a1)(b0 + b1)
// 3 x 49 = 147 multiply-adds instead of 196 — but the three partial products come as 3 x 13 separate 64-bit column
// sums that have to be added into / subtracted from the running column accumulator (a 64-bit addition is two
// instructions, 8.8 issue cycles against 5.6 for the multiply-add it saves), and the middle term needs the column sums
// of z0 and z2 a second time.  y[j] = zs[j] - z0[j] - z2[j] is formed on the zs chain (its accumulator starts at
// -(z0[j] + z2[j])), so the bill is 13 (z0 + z2) + 13 negations + 39 merges.  The quotient-digit half (m * p) is untouched.
__device__ __host__ inline ff28::Fp28 mul28_karatsuba(const ff28::Fp28& a, const ff28::Fp28& b) {
    using namespace ff28;
    typedef unsigned long long u64_;
    constexpr int H = 7;
    u32 sa[H], sb[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
        sa[i] = a.v[i] + a.v[H + i];  // < 2^29: a column of 7 products stays below 2^61
        sb[i] = b.v[i] + b.v[H + i];
    }
    u64_ z0[2 * H - 1], z2[2 * H - 1], y[2 * H - 1];
#pragma unroll
    for (int j = 0; j < 2 * H - 1; ++j) {
        u64_ c0 = 0, c2 = 0;
#pragma unroll
        for (int i = 0; i < H; ++i)
            if (j - i >= 0 && j - i < H) {
                c0 += (u64_)a.v[i] * b.v[j - i];
                c2 += (u64_)a.v[H + i] * b.v[H + j - i];
            }
        z0[j] = c0;
        z2[j] = c2;
        u64_ cs = 0 - (c0 + c2);
#pragma unroll
        for (int i = 0; i < H; ++i)
            if (j - i >= 0 && j - i < H) cs += (u64_)sa[i] * sb[j - i];
        y[j] = cs;
    }
    u32 m[L];
    Fp28 r;
    u64_ acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
        if (k < 2 * H - 1) acc += z0[k];
        if (k >= H && k - H < 2 * H - 1) acc += y[k - H];
        if (k >= 2 * H && k - 2 * H < 2 * H - 1) acc += z2[k - 2 * H];
        if (k < L) {
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (u64_)m[i] * p28(k - i);
            m[k] = ((u32)acc * P0INV) & MASK;
            acc += (u64_)m[k] * p28(0);
        } else {
#pragma unroll
            for (int i = k - L + 1; i < L; ++i) acc += (u64_)m[i] * p28(k - i);
            r.v[k - L] = (u32)acc & MASK;
        }
        acc >>= 28;
    }
    r.v[L - 1] = (u32)acc;
    return r;
}

// The same product with as much instruction-level parallelism as the data flow has: the 27 column sums of a*b on 27
// independent chains, then the Montgomery reduction operand-wise — once m[k] is known its 13 products m[k]*p[j] go to 13
// different columns at once, and the serial path is only  column k complete -> m[k] -> m[k]*p[1] into column k+1 -> carry.
// (The product-scanning form keeps two chains per column.)  For kernels that run ONE wave per SIMD, where nothing else
// fills the issue slots; costs 27 64-bit accumulators.
__device__ __host__ inline ff28::Fp28 mul28_ilp(const ff28::Fp28& a, const ff28::Fp28& b) {
    using namespace ff28;
    typedef unsigned long long u64_;
    u64_ col[2 * L - 1];
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
        u64_ s = 0;
#pragma unroll
        for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); ++i) s += (u64_)a.v[i] * b.v[k - i];
        col[k] = s;
    }
    Fp28 r;
    u64_ carry = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const u64_ t = col[k] + carry;
        const u32 m = ((u32)t * P0INV) & MASK;
#pragma unroll
        for (int j = 1; j < L; ++j) col[k + j] += (u64_)m * p28(j);
        carry = (t + (u64_)m * p28(0)) >> 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
        const u64_ t = col[k] + carry;
        r.v[k - L] = (u32)t & MASK;
        carry = t >> 28;
    }
    r.v[L - 1] = (u32)carry;
    return r;
}

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
    } else if (V == 3) {
        ff28::Fp28 a = ff28::from_sat(x), b = ff28::from_sat(y);
        for (int i = 0; i < iters; ++i) {
            a = mul28_signed(a, b);
            b = mul28_signed(b, a);
        }
        out[2 * tid] = ff28::to_sat(a);
        out[2 * tid + 1] = ff28::to_sat(b);
    } else if (V == 4) {
        ff28::Fp28 a = ff28::from_sat(x), b = ff28::from_sat(y);
        for (int i = 0; i < iters; ++i) {
            a = mul28_karatsuba(a, b);
            b = mul28_karatsuba(b, a);
        }
        out[2 * tid] = ff28::to_sat(a);
        out[2 * tid + 1] = ff28::to_sat(b);
    } else if (V == 5) {
        ff28::Fp28 a = ff28::from_sat(x), b = ff28::from_sat(y);
        for (int i = 0; i < iters; ++i) {
            a = mul28_ilp(a, b);
            b = mul28_ilp(b, a);
        }
        out[2 * tid] = ff28::to_sat(a);
        out[2 * tid + 1] = ff28::to_sat(b);
    }
}

static void host_ref(const Fp* in, Fp* out, int n, int iters, int V) {
    for (int t = 0; t < n; ++t) {
        Fp x = in[2 * t], y = in[2 * t + 1];
        if (V == 1 || V == 3 || V == 4 || V == 5) {
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

// blocks = 2048: eight waves per SIMD (throughput); 256 / 512: ONE / TWO waves per SIMD — the latency of a dependent chain
template <int V>
static void run_mul(const char* name, int blocks = 256 * 8) {
    const int threads = 256, n = blocks * threads, iters = 200;
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
    printf("%-28s %4d blocks %8.3f ms  %8.2f G op/s  %7.1f ns per dependent product  mismatches(vs host, 128 samples)=%d\n", name,
           blocks, ms, ops / (ms * 1e-3) * 1e-9, ms * 1e6 / (iters * 2), bad);
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
    run_instr<16>("v_add_u32", 16);
    run_instr<14>("v_fma_f32", 16);
    run_instr<15>("v_pk_fma_f32", 16);
    run_instr<12>("v_add_f64", 16);
    run_instr<13>("v_lshl_add_u64", 16);
    run_instr<18>("v_mad_i64_i32", 16);
    run_instr<17>("DPFP product (5 instr) x8", 40);
    run_dpfp_check();
    printf("bit-products per instruction: v_mad_u64_u32 on 28-bit limbs 784; DPFP 52 x 52 in 5 instructions 541\n");
    run_mul<0>("fp mul 12x32 CIOS");
    run_mul<1>("fp mul 14x28 comba");
    run_mul<3>("fp mul 14x28, one chain");
    run_mul<4>("fp mul 14x28, Karatsuba 7+7");
    run_mul<1>("fp mul 14x28 comba (again)");
    run_mul<4>("fp mul 14x28, Karatsuba (again)");
    run_mul<3>("fp mul 14x28, one chain (again)");
    run_mul<2>("fp add+sub 12x32");
    run_mul<5>("fp mul 14x28, ILP form");
    for (int rep = 0; rep < 2; ++rep)
        for (int blocks : {256, 512}) {
            run_mul<1>("fp mul 14x28 comba", blocks);
            run_mul<5>("fp mul 14x28, ILP form", blocks);
        }
    return 0;
}
