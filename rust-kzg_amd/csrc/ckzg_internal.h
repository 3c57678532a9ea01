// Internal glue between the c-kzg layer and the engines.
#pragma once
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "ff.hip.h"

struct KzgAmdSettings;

namespace kzgamd {
KzgAmdSettings* device_settings(const CKZGSettings* s);

// roots_of_unity[0..=2^scale] in Montgomery form (expand_root_of_unity,
// blst/src/types/fft_settings.rs:90-106); the generator of the 2-adic subgroup is
// 7^((r-1)/2^32) as tabulated by SCALE2_ROOT_OF_UNITY (blst/src/consts.rs:17-50)
inline ff::Fr scale2_root_of_unity(unsigned scale) {
    // (r-1)/2^32
    const ff::u32 e[8] = {0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u, 0u};
    ff::Fr seven = ff::Fr::zero();
    seven.v[0] = 7;
    seven = ff::to_mont(seven);
    ff::Fr w = ff::pow_u32(seven, e, 8);
    for (unsigned k = 32; k > scale; --k) w = ff::sqr(w);
    return w;
}

inline void expand_roots(std::vector<ff::Fr>& roots, unsigned scale) {
    const size_t W = (size_t)1 << scale;
    ff::Fr root = scale2_root_of_unity(scale);
    roots.resize(W + 1);
    roots[0] = ff::Fr::one();
    for (size_t i = 1; i <= W; ++i) roots[i] = ff::mul(roots[i - 1], root);
}

// blst_p1_is_equal on the host (projective equivalence of two Jacobian points)
inline bool host_p1_equal(const blst_p1* a, const blst_p1* b) {
    const ff::Fp* A = reinterpret_cast<const ff::Fp*>(a);
    const ff::Fp* B = reinterpret_cast<const ff::Fp*>(b);
    const bool ia = A[2].is_zero(), ib = B[2].is_zero();
    if (ia || ib) return ia && ib;
    ff::Fp z1z1 = ff::sqr(A[2]), z2z2 = ff::sqr(B[2]);
    if (ff::mul(A[0], z2z2) != ff::mul(B[0], z1z1)) return false;
    return ff::mul(A[1], ff::mul(z2z2, B[2])) == ff::mul(B[1], ff::mul(z1z1, A[2]));
}
}  // namespace kzgamd
