// Differential check of the chain routines of fftg1.hip (grp::dbl_body<G>, grp::dadd_body<G>, grp::run_chain<G>, G = 1, 2, 4)
// against the single-lane ones of g1_28.hip.h (g1::dbl, g1::dadd, a plain double-and-add) on multiples of the generator.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/grp_check.hip -o tools/grp_check -Lrust-kzg_amd/csrc -lkzg_mi355x
#include "../rust-kzg_amd/csrc/fftg1.hip"

#include <vector>

namespace {
__device__ bool same_point(const Xyzz& a, const Xyzz& b) {
    using namespace fp28;
    if (g1::is_inf(a) || g1::is_inf(b)) return g1::is_inf(a) && g1::is_inf(b);
    const bool x = is_zero_mod_p(sub<4>(mul(a.x, b.zz), mul(b.x, a.zz)));
    const bool y = is_zero_mod_p(sub<4>(mul(a.y, b.zzz), mul(b.y, a.zzz)));
    return x && y;
}

// plain double-and-add over the 128 bits of k with the single-lane routines of g1_28.hip.h: the yardstick
__device__ void mul_plain(Xyzz& acc, const u32 k[4]) {
    const Xyzz base = acc;
    g1::set_inf(acc);
    for (int b = 127; b >= 0; --b) {
        if (!g1::is_inf(acc)) g1::dbl(acc);
        if ((k[b >> 5] >> (b & 31)) & 1) g1::dadd(acc, base);
    }
}

template <int G>
__global__ void __launch_bounds__(64) k_check(int* __restrict__ bad, const Xyzz* __restrict__ pts, Xyzz* __restrict__ tab, size_t ngroups, Xyzz* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ngroups * G) return;
    const int r = (int)(t % G);
    const size_t g = t / G;
    const Xyzz P = pts[g], Q = pts[(g + 1) % ngroups];
    Xyzz a = P, b = P;
    g1::dbl(a);
    if (!same_point(a, a)) atomicOr(&bad[g], 16);   // the comparison itself
    {
        Xyzz c = P;
        g1::dadd(c, P);
        if (!same_point(a, c)) atomicOr(&bad[g], 32);  // single-lane dbl == single-lane P + P
    }
    grp::dbl_body<G>(b, r);
    if (!same_point(a, b)) atomicOr(&bad[g], 1);
    if (g == 0 && out) {
        out[2 * r] = a;
        out[2 * r + 1] = b;
    }
    a = P;
    b = P;
    g1::dadd(a, Q);
    if (grp::dadd_body<G>(b, Q, r)) atomicOr(&bad[g], 64);  // distinct points: no doubling asked for
    if (!same_point(a, b)) atomicOr(&bad[g], 2);
    b = P;
    if (!grp::dadd_body<G>(b, P, r)) atomicOr(&bad[g], 4);  // P + P: the caller is told to double
    u32 k[4] = {0x9e3779b9u * (u32)(g + 1), 0x85ebca6bu ^ (u32)g, 0xc2b2ae35u + (u32)g, 0x27d4eb2fu >> 1};
    a = P;
    b = P;
    mul_plain(a, k);
    grp::run_chain<G>(b, true, k[0], k[1], k[2], k[3], tab + g, ngroups, false, 0, [](int) { return grp::TailOp{nullptr, false}; }, r);
    if (!same_point(a, b)) atomicOr(&bad[g], 8);
}
}  // namespace

int main() {
    const size_t n = 65;  // points G, 2G, ... as XYZZ
    std::vector<blst_p1> host(n);
    void* d_aff = nullptr;
    hipMalloc(&d_aff, n * sizeof(blst_p1_affine));
    RustError e = kzgamd_generate_points(d_aff, n, 7, nullptr);
    if (e.code) return 2;
    std::vector<blst_p1_affine> aff(n);
    hipMemcpy(aff.data(), d_aff, n * sizeof(blst_p1_affine), hipMemcpyDeviceToHost);
    // Z = 1 in Montgomery form
    const uint64_t one[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                             0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
    for (size_t i = 0; i < n; ++i) {
        host[i].x = aff[i].x;
        host[i].y = aff[i].y;
        for (int j = 0; j < 6; ++j) host[i].z.l[j] = one[j];
    }
    void *d_in = nullptr, *d_pts = nullptr, *d_tab = nullptr;
    int* d_bad = nullptr;
    void* d_out = nullptr;
    hipMalloc(&d_out, 8 * sizeof(Xyzz));
    hipMalloc(&d_in, n * sizeof(blst_p1));
    hipMalloc(&d_pts, n * sizeof(Xyzz));
    hipMalloc(&d_tab, 64 * (8 + 15 * 4) * sizeof(Xyzz));
    hipMalloc(&d_bad, 64 * sizeof(int));
    hipMemcpy(d_in, host.data(), n * sizeof(blst_p1), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_g1_load, dim3(1), dim3(256), 0, 0, (Xyzz*)d_pts, (const ff::Fp*)d_in, 64u, 6, (size_t)64);  // 64 distinct points (bit-reversed order)
    int rc = 0;
    for (int G : {1, 2, 4}) {
        hipMemset(d_bad, 0, 64 * sizeof(int));
        const size_t ng = 64;
        if (G == 1) hipLaunchKernelGGL(k_check<1>, dim3(1), dim3(64), 0, 0, d_bad, (const Xyzz*)d_pts, (Xyzz*)d_tab, ng, (Xyzz*)d_out);
        else if (G == 2) hipLaunchKernelGGL(k_check<2>, dim3(2), dim3(64), 0, 0, d_bad, (const Xyzz*)d_pts, (Xyzz*)d_tab, ng, (Xyzz*)d_out);
        else hipLaunchKernelGGL(k_check<4>, dim3(4), dim3(64), 0, 0, d_bad, (const Xyzz*)d_pts, (Xyzz*)d_tab, ng, (Xyzz*)d_out);
        int bad[64];
        hipError_t he = hipMemcpy(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
        int any = 0;
        for (int i = 0; i < 64; ++i) any |= bad[i];
        printf("G=%d: %s  flags(or)=%d  (1 dbl, 2 add, 4 add-as-dbl, 8 scalar mul, 64 spurious dbl request)  first groups: %d %d %d %d  hip=%d\n", G,
               any ? "MISMATCH" : "ok", any, bad[0], bad[1], bad[2], bad[3], (int)he);
        rc |= any;
        Xyzz o[8];
        hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
        for (int l = 0; l < G; ++l)
            for (int w = 0; w < 2; ++w) {
                printf("G=%d lane %d %s:", G, l, w ? "grp" : "one");
                const u32* q = (const u32*)&o[2 * l + w];
                for (int i = 0; i < 56; ++i) printf(" %x", q[i]);
                printf("\n");
            }
    }
    return rc ? 1 : 0;
}
