// Host check of ff::inverse_plain_fast (the batched binary GCD) against ff::inverse_plain_bgcd (the bit-by-bit one) on
// edge values (small, powers of two and their predecessors, p - small) and random values of several shapes, for Fr and
// Fp; counts the calls that fell back to the bit-by-bit algorithm (expected: none).
// Build + run: g++ -O2 -std=c++17 -I rust-kzg_amd/csrc tools/inv_check.cpp -o /tmp/inv_check && /tmp/inv_check 100000
#include <cstdio>
#include <cstdlib>
#include <random>
#include <chrono>
static long g_fallbacks = 0;
#define FF_INV_COUNT_FALLBACK g_fallbacks
#include "ff.hip.h"
template <class P> int run(const char* name, int iters) {
    typedef ff::Field<P> F;
    std::mt19937_64 rng(7);
    int bad = 0;
    auto check = [&](const F& y) {
        F a = ff::inverse_plain_bgcd(y), b = ff::inverse_plain_fast(y);
        for (int i = 0; i < P::N; ++i) if (a.v[i] != b.v[i]) { if (bad < 5) { printf("%s mismatch:", name); for (int k = P::N-1; k >= 0; --k) printf(" %08x", y.v[k]); printf("\n"); } ++bad; return; }
    };
    auto lt_p = [](const F& y) { uint64_t b = 0; for (int i = 0; i < P::N; ++i) { uint64_t d = (uint64_t)y.v[i] - P::p(i) - b; b = (d >> 32) & 1; } return b != 0; };
    // edge values
    for (uint32_t v = 0; v < 70; ++v) { F y = F::zero(); y.v[0] = v; check(y); }
    for (int bit = 0; bit < 32 * P::N; ++bit) {
        F y = F::zero(); y.v[bit >> 5] = 1u << (bit & 31); if (lt_p(y)) check(y);
        F z = y; for (int i = 0; i < (bit >> 5); ++i) z.v[i] = 0xffffffffu; z.v[bit >> 5] = (1u << (bit & 31)) - 1; if (lt_p(z) && !z.is_zero()) check(z);
    }
    for (uint32_t d = 1; d < 70; ++d) {  // p - d
        F y; uint64_t b = d; for (int i = 0; i < P::N; ++i) { uint64_t e = (uint64_t)P::p(i) - b; y.v[i] = (uint32_t)e; b = (e >> 32) & 1; } check(y);
    }
    for (int it = 0; it < iters; ++it) {
        F y; for (int i = 0; i < P::N; ++i) y.v[i] = (uint32_t)rng();
        int mode = it % 7;
        if (mode == 1) for (int i = P::N / 2; i < P::N; ++i) y.v[i] = 0;
        if (mode == 2) for (int i = 0; i < P::N / 2; ++i) y.v[i] = 0;
        if (mode == 3) for (int i = 1; i < P::N; ++i) y.v[i] = 0;
        if (mode == 4) { y.v[0] &= ~0xffffu; }
        y.v[P::N - 1] &= (mode == 5 ? 0x0fffffffu : 0xffffffffu);
        while (!lt_p(y)) y.v[P::N - 1] >>= 1;
        if (y.is_zero()) continue;
        check(y);
    }
    // timing
    F y; for (int i = 0; i < P::N; ++i) y.v[i] = (uint32_t)rng(); while (!lt_p(y)) y.v[P::N-1] >>= 1;
    auto t0 = std::chrono::steady_clock::now(); uint32_t s = 0;
    for (int i = 0; i < 20000; ++i) { F r = ff::inverse_plain_bgcd(y); s += r.v[0]; y.v[0] += 2; }
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20000; ++i) { F r = ff::inverse_plain_fast(y); s += r.v[0]; y.v[0] += 2; }
    auto t2 = std::chrono::steady_clock::now();
    printf("fallbacks so far %ld\n", g_fallbacks); printf("%s: %d mismatches; old %.2f us, new %.2f us per inverse (host) [%u]\n", name, bad,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 20000, std::chrono::duration<double, std::micro>(t2 - t1).count() / 20000, s);
    return bad;
}
int main(int argc, char** argv) { int n = argc > 1 ? atoi(argv[1]) : 200000; return run<ff::FrParams>("Fr", n) + run<ff::FpParams>("Fp", n) ? 1 : 0; }
