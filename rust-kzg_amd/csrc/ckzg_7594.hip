// EIP-7594 (kzg/src/das.rs; kzg/src/eth/c_bindings.rs:202-372): cells and cell proofs (FK20 for batches, one fixed-base
// MSM per cell for single blobs), recovery, cell verification.
#include "ckzg_shared.h"

namespace {

// ---------------- EIP-7594 cells + cell proofs ----------------
__device__ __forceinline__ u32 brev32(u32 v, int bits) { return __builtin_bitreverse32(v) >> (32 - bits); }

// blob bytes -> Montgomery Fr in bit-reversed order (blob_to_polynomial + reverse_bit_order of
// poly_lagrange_to_monomial, kzg/src/das.rs:618-629); status = 1 when an element is >= r
__global__ void __launch_bounds__(256) k_blob_to_fr_brp(ff::Fr* __restrict__ out, int* __restrict__ status,
                                                        const u32* __restrict__ blobs, size_t nblobs) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * N) return;
    const size_t b = t / N, i = t % N;
    bool ok;
    ff::Fr v = fr_load_be(blobs + (b * N + brev32((u32)i, 12)) * 8, &ok);
    if (!ok) status[b] = 1;
    out[t] = ff::to_mont(v);
}

// monomial coefficients (4096) -> zero-extended 8192 (das.rs:260-261)
__global__ void __launch_bounds__(256) k_zero_extend(ff::Fr* __restrict__ ext, const ff::Fr* __restrict__ mono, size_t nblobs) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 2 * N) return;
    const size_t b = t / (2 * N), i = t % (2 * N);
    ext[t] = i < N ? mono[b * N + i] : ff::Fr::zero();
}

// evaluations on the 8192 domain -> cells: bit-reversed order, 32-byte big-endian (das.rs:267-275)
__global__ void __launch_bounds__(256) k_cells_out(u32* __restrict__ cells, const ff::Fr* __restrict__ ev, size_t nblobs) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 2 * N) return;
    const size_t b = t / (2 * N), f = t % (2 * N);
    ff::Fr v = ff::from_mont(ev[b * 2 * N + brev32((u32)f, 13)]);
#pragma unroll
    for (int k = 0; k < 8; ++k) cells[t * 8 + k] = __builtin_bswap32(v.v[7 - k]);
}

// Quotient coefficients of the 128 cell proofs: q_k = p div (X^64 - a_k), a_k = w_128^brp7(k).
// The reference reaches the same commitments through FK20 (Toeplitz FFTs + fft_g1, kzg/src/das.rs:660-696);
// with the 4096-point wide table a proof is simply one more fixed-base MSM, so the division recurrence
//   q_j = p_{j+64} + a_k * q_{j+64}
// is run per (cell, residue class mod 64) and the 128 scalar vectors go to the MSM engine.
__global__ void __launch_bounds__(64) k_cell_quotients(u32* __restrict__ q_out, const ff::Fr* __restrict__ mono,
                                                       const ff::Fr* __restrict__ roots8192, size_t nblobs) {
    const size_t b = blockIdx.x / 128, k = blockIdx.x % 128;
    const int r = threadIdx.x;  // residue class
    const ff::Fr a = roots8192[64 * brev32((u32)k, 7)];
    const ff::Fr* p = mono + b * N;
    u32* q = q_out + (b * 128 + k) * N * 8;
    ff::Fr acc = ff::Fr::zero();
    // j = r + 64*t ; top quotient index is N - 65
#pragma unroll 1
    for (int t = 63; t >= 0; --t) {
        const int j = r + 64 * t;
        ff::Fr v;
        if (t == 63) {
            v = ff::Fr::zero();  // q_j = 0 for j >= N - 64
        } else {
            acc = ff::add(p[j + 64], ff::mul(a, acc));
            v = acc;
        }
        ff::Fr c = ff::from_mont(v);
#pragma unroll
        for (int l = 0; l < 8; ++l) q[(size_t)j * 8 + l] = c.v[l];
    }
}

// ---------------- FK20 cell proofs (compute_fk20_proofs, kzg/src/das.rs:630-696) for batches ----------------
// toeplitz_coeffs_stride for every (blob, offset i < 64): a 128-vector with p[4095 - i] at 0 and p[4095 - i - 64 j] at
// 128 - j, j = 1 .. 62 (the circulant embedding of the Toeplitz matrix of every 64th coefficient)
__global__ void __launch_bounds__(256) k_fk20_toeplitz(ff::Fr* __restrict__ out, const ff::Fr* __restrict__ mono, size_t nblobs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 64 * 128) return;
    const size_t b = t >> 13;
    const u32 i = (u32)(t >> 7) & 63, idx = (u32)t & 127;
    const ff::Fr* p = mono + b * N;
    ff::Fr v = ff::Fr::zero();
    if (idx == 0) v = p[N - 1 - i];
    else if (idx >= 66) v = p[N - 1 - i - 64 * (128 - idx)];
    out[t] = v;
}
// coeffs[blob][j][i] = transform_i[j] / 128: the scalars of column j next to each other (the MSM's layout), with the
// 1/128 of the inverse G1 transform that follows folded in (a field multiplication here instead of a scalar
// multiplication per point there)
__global__ void __launch_bounds__(256) k_fk20_transpose(ff::Fr* __restrict__ out, const ff::Fr* __restrict__ in, size_t nblobs,
                                                        ff::Fr inv128) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 128 * 64) return;
    const size_t b = t >> 13;
    const u32 j = (u32)(t >> 6) & 127, i = (u32)t & 63;
    out[t] = ff::mul(in[(b * 64 + i) * 128 + j], inv128);
}
// h[64 .. 128) = identity (das.rs:688-691)
__global__ void __launch_bounds__(256) k_fk20_zero_upper(g1::Xyzz* __restrict__ h, size_t nblobs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 64) return;
    g1::set_inf(h[(t >> 6) * 128 + 64 + (t & 63)]);
}
// reverse_bit_order of the 128 proofs of a blob (das.rs:288)
__global__ void __launch_bounds__(256) k_fk20_brp(g1::Xyzz* __restrict__ out, const g1::Xyzz* __restrict__ in, size_t nblobs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblobs * 128) return;
    out[t] = in[(t & ~(size_t)127) | (__builtin_bitreverse32((u32)t & 127u) >> 25)];
}

// the fixed-base handle over the 128 x 64 points of x_ext_fft_columns (column-major: base j * 64 + i)
void fk20_prepare(KzgAmdSettings* dev, const CKZGSettings* cs) {
    if (dev->msm_xext) return;
    const size_t K2 = 2 * CELLS_PER_BLOB, total = K2 * CELL_SIZE;
    std::vector<ff::Fp> aff(2 * total);  // blst_p1_affine: x, y
    std::vector<ff::Fp> pre(total);
    // Montgomery's trick over the Z coordinates (the columns hold no point at infinity for a valid setup; a zero Z
    // is skipped and its point written as (0, 0), blst's affine infinity)
    ff::Fp run = ff::Fp::one();
    for (size_t k = 0; k < total; ++k) {
        const ff::Fp* P = reinterpret_cast<const ff::Fp*>(&cs->x_ext_fft_columns[k / CELL_SIZE][k % CELL_SIZE]);
        pre[k] = run;
        if (!P[2].is_zero()) run = hfp::mul(run, P[2]);
    }
    ff::Fp inv = ff::inverse_bgcd(run);
    for (size_t k = total; k-- > 0;) {
        const ff::Fp* P = reinterpret_cast<const ff::Fp*>(&cs->x_ext_fft_columns[k / CELL_SIZE][k % CELL_SIZE]);
        if (P[2].is_zero()) {
            aff[2 * k] = aff[2 * k + 1] = ff::Fp::zero();
            continue;
        }
        const ff::Fp zi = hfp::mul(inv, pre[k]), zi2 = hfp::sqr(zi);
        inv = hfp::mul(inv, P[2]);
        aff[2 * k] = hfp::mul(P[0], zi2);
        aff[2 * k + 1] = hfp::mul(P[1], hfp::mul(zi2, zi));
    }
    dev->msm_xext = kzgamd::msm_create(aff.data(), total, false, true, false, kzgamd::G1_TRUSTED, &dev->opt);
}

// The 128 cell proofs of n polynomials whose 4096 monomial coefficients are in dev->d_fr_b, compressed into
// dev->d_proofs (compute_fk20_proofs + reverse_bit_order, kzg/src/das.rs:280-288, 630-696): enqueue only.
// The caller has prepared the handle its `fk20` choice needs and the buffers (ensure_fk20 / ensure_q).
void enqueue_cell_proofs(KzgAmdSettings* dev, size_t n, hipStream_t st, bool fk20) {
    if (fk20) {
        const size_t nv = n * 64 * 128;
        hipLaunchKernelGGL(k_fk20_toeplitz, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, dev->d_fk_a,
                           (const ff::Fr*)dev->d_fr_b, n);
        if (kzgamd_ntt_fr_device(dev->ntt, dev->d_fk_b, dev->d_fk_a, 128, 64 * n, 0, st) != 0) throw CkErr{C_KZG_ERROR, "ntt"};
        ff::Fr k128 = ff::Fr::zero();
        k128.v[0] = 128;
        const ff::Fr inv128 = ff::inverse_bgcd(ff::to_mont(k128));  // Montgomery form of 1/128
        hipLaunchKernelGGL(k_fk20_transpose, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, dev->d_fk_a,
                           (const ff::Fr*)dev->d_fk_b, n, inv128);
        // h_ext_fft[blob][j] = sum_i coeffs[j][i] * x_ext_fft_columns[j][i]: 128 n MSMs of 64 points, column j of the table
        kzgamd::msm_lock(dev->msm_xext);
        try {
            kzgamd::msm_enqueue(dev->msm_xext, dev->d_fk_h, dev->d_fk_a, CELL_SIZE, n * 128, 1, st, kzgamd::OUT_XYZZ, false, 128);
        } catch (...) {
            kzgamd::msm_unlock(dev->msm_xext);
            throw;
        }
        kzgamd::msm_unlock(dev->msm_xext);
        // h = ifft_g1(h_ext_fft), upper half cleared, proofs = fft_g1(h), bit-reversed, compressed
        g1::Xyzz* h = (g1::Xyzz*)kzgamd::fftg1_device((NttCtx*)dev->ntt, dev->d_fk_h, dev->d_fk_h2, 128, n, 1, st, false);
        if (!h) throw CkErr{C_KZG_ERROR, "fft_g1"};
        g1::Xyzz* other = h == dev->d_fk_h ? dev->d_fk_h2 : dev->d_fk_h;
        hipLaunchKernelGGL(k_fk20_zero_upper, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, h, n);
        g1::Xyzz* pr = (g1::Xyzz*)kzgamd::fftg1_device((NttCtx*)dev->ntt, h, other, 128, n, 0, st);
        if (!pr) throw CkErr{C_KZG_ERROR, "fft_g1"};
        g1::Xyzz* fin = pr == h ? other : h;
        hipLaunchKernelGGL(k_fk20_brp, dim3((unsigned)((n * 128 + 255) / 256)), dim3(256), 0, st, fin, (const g1::Xyzz*)pr, n);
        kzgamd::g1_compress_xyzz(dev->d_proofs, fin, n * 128, st);
    } else {
        hipLaunchKernelGGL(k_cell_quotients, dim3((unsigned)(n * 128)), dim3(64), 0, st, dev->d_q, (const ff::Fr*)dev->d_fr_b,
                           (const ff::Fr*)dev->d_roots8192, n);
        kzgamd::msm_lock(dev->msm_monomial);
        try {
            kzgamd::msm_enqueue(dev->msm_monomial, dev->d_proofs, dev->d_q, N, n * 128, 0, st, kzgamd::OUT_COMPRESSED);
        } catch (...) {
            kzgamd::msm_unlock(dev->msm_monomial);
            throw;
        }
        kzgamd::msm_unlock(dev->msm_monomial);
    }
}

// compute_cells_and_kzg_proofs (kzg/src/das.rs:244-292) for n blobs; cells / proofs may be null (not both)
void cells_and_proofs(uint8_t* cells, KZGProof* proofs, const Blob* blobs, size_t n, const CKZGSettings* cs,
                      KzgAmdSettings* dev) {
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    if (!dev->d_roots8192) {
        CK_HIP(hipMalloc(&dev->d_roots8192, (2 * N + 1) * sizeof(ff::Fr)));
        CK_HIP(hipMemcpy(dev->d_roots8192, cs->roots_of_unity, (2 * N + 1) * sizeof(ff::Fr), hipMemcpyHostToDevice));
    }
    // Cell proofs: FK20 for batches (the reference's algorithm: 64 transforms of 128 scalars, 128 MSMs of 64 points
    // over x_ext_fft_columns, two G1 transforms of 128 points — ~25x fewer point additions than 128 MSMs of 4096, but
    // the G1 transforms are 14 serial stages of a 128-bit scalar multiplication each: tens of ms of latency whatever
    // the batch).  A few blobs: the direct form, one more fixed-base MSM per cell over the monomial table.
    // Tuning key fk20 = 0 / 1 forces one or the other.
    bool fk20 = proofs && n >= FK20_MIN_BLOBS;
    if (dev->cfg_fk20 >= 0) fk20 = proofs && dev->cfg_fk20 != 0;
    if (dev->fk20_unavailable) fk20 = false;
    if (proofs && fk20) {
        // no HBM left for the FK20 table (creation throws, or succeeds without a wide table): the direct form computes
        // the same proofs; the useless handle is dropped and the choice remembered
        try {
            fk20_prepare(dev, cs);
        } catch (...) {
            dev->fk20_unavailable = true;
        }
        if (!dev->fk20_unavailable && !kzgamd::msm_has_wide_table(dev->msm_xext)) dev->fk20_unavailable = true;
        if (dev->fk20_unavailable) {
            if (dev->msm_xext) kzgamd::msm_destroy(dev->msm_xext);
            dev->msm_xext = nullptr;
            fk20 = false;
        }
    }
    if (proofs && !fk20 && !dev->msm_monomial) dev->msm_monomial = kzgamd::msm_create(dev->d_monomial, N, true, true, true, kzgamd::G1_TRUSTED, &dev->opt);
    dev->ensure(n);
    dev->ensure_cells(n);
    if (proofs && fk20) dev->ensure_fk20(n);
    if (proofs && !fk20) dev->ensure_q(n);
    hipStream_t st = dev->stream;
    CK_HIP(hipMemcpyAsync(dev->d_blobs, blobs, n * BYTES_PER_BLOB, hipMemcpyHostToDevice, st));
    CK_HIP(hipMemsetAsync(dev->d_status, 0, n * sizeof(int), st));
    hipLaunchKernelGGL(k_blob_to_fr_brp, dim3((unsigned)((n * N + 255) / 256)), dim3(256), 0, st, dev->d_fr_a, dev->d_status,
                       (const u32*)dev->d_blobs, n);
    // poly_lagrange_to_monomial: inverse NTT of the bit-reversed evaluations
    if (kzgamd_ntt_fr_device(dev->ntt, dev->d_fr_b, dev->d_fr_a, N, n, 1, st) != 0) throw CkErr{C_KZG_ERROR, "ntt"};
    if (cells) {
        hipLaunchKernelGGL(k_zero_extend, dim3((unsigned)((n * 2 * N + 255) / 256)), dim3(256), 0, st, dev->d_fr_ext,
                           (const ff::Fr*)dev->d_fr_b, n);
        // d_fr_a is free again only for n*N elements; the 8192-point result needs its own buffer: reuse d_cells
        // as scratch for the transform output, then convert in place through d_fr_ext
        ff::Fr* ev = reinterpret_cast<ff::Fr*>(dev->d_cells);
        if (kzgamd_ntt_fr_device(dev->ntt, ev, dev->d_fr_ext, 2 * N, n, 0, st) != 0) throw CkErr{C_KZG_ERROR, "ntt"};
        hipLaunchKernelGGL(k_cells_out, dim3((unsigned)((n * 2 * N + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<u32*>(dev->d_fr_ext), (const ff::Fr*)ev, n);
        // fetched below, after the proof kernels are enqueued: a copy into pageable memory blocks this thread
    }
    // The cells (256 KiB per blob) go back on the second stream while the proof kernels run on the first: the copy
    // engine is idle during FK20 (256 blobs: 67 MB, ~2.5 ms that used to follow the proofs on the same stream).
    const bool side_copy = cells && proofs && dev->stream2 && dev->stream2 != st;
    if (side_copy) {
        if (!dev->ev_cells) CK_HIP(hipEventCreateWithFlags(&dev->ev_cells, hipEventDisableTiming));
        CK_HIP(hipEventRecord(dev->ev_cells, st));
    }
    std::vector<int> status(n);
    try {
        if (proofs) enqueue_cell_proofs(dev, n, st, fk20);
        if (side_copy) {
            CK_HIP(hipStreamWaitEvent(dev->stream2, dev->ev_cells, 0));
            CK_HIP(hipMemcpyAsync(cells, dev->d_fr_ext, n * 2 * N * 32, hipMemcpyDeviceToHost, dev->stream2));
        } else if (cells) {
            CK_HIP(hipMemcpyAsync(cells, dev->d_fr_ext, n * 2 * N * 32, hipMemcpyDeviceToHost, st));
        }
        if (proofs) CK_HIP(hipMemcpyAsync(proofs, dev->d_proofs, n * 128 * 48, hipMemcpyDeviceToHost, st));
        CK_HIP(hipMemcpyAsync(status.data(), dev->d_status, n * sizeof(int), hipMemcpyDeviceToHost, st));
    } catch (...) {
        // a copy into the caller's `cells` / `proofs` (or into `status`) may be in flight on either stream: nothing of it
        // may outlive this call
        (void)hipStreamSynchronize(st);
        if (side_copy) (void)hipStreamSynchronize(dev->stream2);
        throw;
    }
    const hipError_t e1 = hipStreamSynchronize(st);
    if (side_copy) CK_HIP(hipStreamSynchronize(dev->stream2));  // also on the way out of a failure: `cells` is the caller's
    CK_HIP(e1);
    for (size_t i = 0; i < n; ++i) CK_REQUIRE(status[i] == 0, "Invalid scalar");
}


// ---------------- EIP-7594 recovery (kzg/src/das.rs:101-243, 566-657) ----------------
// provided cells (canonical little-endian limbs, already checked < r on the host) -> the 8192 evaluations in
// bit-reversed order, Montgomery form, missing positions and the reference's "null" sentinel as zero
// (recover_cells: `if cells_brp[i].is_null() { zero }`, das.rs:611-617; Fr::null() = from_u64_arr([u64::MAX; 4]),
// blst/src/types/fr.rs:36-38 — a provided element equal to it is dropped by the reference too).  drop_null is false
// when all 128 cells are given: the reference then skips recover_cells (das.rs:172-181) and hands the values as they
// are to poly_lagrange_to_monomial (:186-188), the sentinel value included.
__global__ void __launch_bounds__(256) k_rec_scatter(ff::Fr* __restrict__ ev_brp, const u32* __restrict__ limbs,
                                                     const u32* __restrict__ cell_idx, size_t ncells, bool drop_null) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncells * CELL_SIZE) return;
    const u32 c = cell_idx[t >> 6], j = (u32)t & 63;
    ff::Fr v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v.v[k] = limbs[t * 8 + k];
    v = ff::to_mont(v);
    ff::Fr nul;
#pragma unroll
    for (int k = 0; k < 8; ++k) nul.v[k] = 0xffffffffu;
    nul = ff::to_mont(nul);  // from_u64_arr reduces: (2^256 - 1) mod r in Montgomery form
    if (drop_null && v == nul) v = ff::Fr::zero();
    ev_brp[brev32(c * (u32)CELL_SIZE + j, 13)] = v;
}
__global__ void __launch_bounds__(256) k_fr_mul(ff::Fr* __restrict__ out, const ff::Fr* __restrict__ a, const ff::Fr* __restrict__ b,
                                                size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = fr29::mul_blst(a[t], b[t]);
}
// 1 / x per element (batch_inverse of the vanishing polynomial over the coset, das.rs:628-630: never zero there)
__global__ void __launch_bounds__(64) k_fr_inverse(ff::Fr* __restrict__ data, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) data[t] = ff::inverse_bgcd(data[t]);
}

}  // namespace

// ================================================================ EIP-7594: cell verification and recovery
namespace {

constexpr size_t CELLS_PER_EXT_BLOB = 2 * CELLS_PER_BLOB;  // 128
constexpr size_t BYTES_PER_CELL = CELL_SIZE * 32;

inline u32 rbl7(u32 i) {  // CELL_INDICES_RBL (das.rs:87-96): reverse_bits_limited(128, i)
    u32 r = 0;
    for (int b = 0; b < 7; ++b)
        if (i & (1u << b)) r |= 1u << (6 - b);
    return r;
}

inline void put_u64_be(uint8_t* p, uint64_t v) {
    for (int i = 0; i < 8; ++i) p[7 - i] = (uint8_t)(v >> (8 * i));
}

// hash_to_bls_field (kzg/src/eip_4844.rs:916-918): 32 big-endian bytes reduced mod r, Montgomery form
inline ff::Fr hash_to_fr(const uint8_t digest[32]) {
    ff::Fr v;
    for (int i = 0; i < 8; ++i) {
        const uint8_t* q = digest + (7 - i) * 4;
        v.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    return ff::mul(v, ff::Fr::r2());
}

// compute_verify_cell_kzg_proof_batch_challenge (kzg/src/das.rs:391-452) on the caller's bytes: FsG1::to_bytes /
// FsFr::to_bytes of a decoded, valid input are the input bytes themselves
ff::Fr cell_batch_challenge(const Bytes48* commitments, size_t ncommit, const uint64_t* commitment_indices,
                            const uint64_t* cell_indices, const Cell* cells, const Bytes48* proofs, size_t ncells) {
    kzgamd::Sha256 h;
    uint8_t head[48];
    memcpy(head, "RCKZGCBATCH__V1_", 16);
    put_u64_be(head + 16, N);
    put_u64_be(head + 24, CELL_SIZE);
    put_u64_be(head + 32, ncommit);
    put_u64_be(head + 40, ncells);
    h.update(head, 48);
    for (size_t i = 0; i < ncommit; ++i) h.update(commitments[i].bytes, 48);
    for (size_t i = 0; i < ncells; ++i) {
        uint8_t ix[16];
        put_u64_be(ix, commitment_indices[i]);
        put_u64_be(ix + 8, cell_indices[i]);
        h.update(ix, 16);
        h.update(cells[i].bytes, BYTES_PER_CELL);
        h.update(proofs[i].bytes, 48);
    }
    uint8_t digest[32];
    h.finish(digest);
    return hash_to_fr(digest);
}

// cells -> field elements (FsFr::from_bytes per element, c_bindings.rs:225-233): canonical limbs, false if any >= r
bool cells_to_limbs(std::vector<ff::Fr>& out, const Cell* cells, size_t ncells) {
    out.resize(ncells * CELL_SIZE);
    bool ok = true;
    for (size_t i = 0; i < ncells; ++i)
        for (size_t j = 0; j < CELL_SIZE; ++j) ok = fr_from_be32_checked(out[i * CELL_SIZE + j], cells[i].bytes + 32 * j) && ok;
    return ok;
}

// decode `np` compressed G1 points on the GPU (stream2): AffPt slots in dev->d_vpts, per-point status in dev->d_vstat
// (0 ok, 1 not an encoding of a curve point, 2 on the curve but outside G1).  Caller holds dev->vmu.
// `tail` (optional): `ntail` slots decoded and checked by an earlier call, appended behind the np decoded ones (status 0)
void decode_points_begin(KzgAmdSettings* dev, const std::vector<uint8_t>& bytes, size_t np, const AffPt* tail = nullptr,
                         size_t ntail = 0) {
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    dev->ensure_verify(np + ntail);
    dev->vstage = bytes;
    hipStream_t st = dev->stream2;
    CK_HIP(hipMemcpyAsync(dev->d_vbytes, dev->vstage.data(), np * 48, hipMemcpyHostToDevice, st));
    CK_HIP(hipMemsetAsync(dev->d_vstat, 0, (np + ntail) * sizeof(int), st));
    CK_HIP(hipMemsetAsync(dev->d_vpts, 0, np * sizeof(AffPt), st));
    if (ntail) CK_HIP(hipMemcpyAsync(dev->d_vpts + np, tail, ntail * sizeof(AffPt), hipMemcpyDeviceToDevice, st));
    if (!dev->ev_decoded) CK_HIP(hipEventCreateWithFlags(&dev->ev_decoded, hipEventDisableTiming));
    decode_check_enqueue(dev->d_vpts, dev->d_vstat, (const unsigned char*)dev->d_vbytes, np, st, dev->cfg_wide_check, dev->ev_decoded);
    CK_HIP(hipGetLastError());
}
std::vector<int> decode_points_status(KzgAmdSettings* dev, size_t np) {
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    std::vector<int> stat(np);
    CK_HIP(hipMemcpyAsync(stat.data(), dev->d_vstat, np * sizeof(int), hipMemcpyDeviceToHost, dev->stream2));
    CK_HIP(hipStreamSynchronize(dev->stream2));
    return stat;
}

// The aggregated interpolation polynomial of verify_cell_kzg_proof_batch
// (compute_commitment_to_aggregated_interpolation_poly, kzg/src/das.rs:778-835) on the GPU — on the host its ~50 000
// field multiplications were three quarters of a 128-cell call.
// k_vcell_agg: agg[col][brp6(f)] = sum over the cells i of column col of r^i * cell_i[f]  (r^i Montgomery, the cell
// elements canonical: the products and sums stay canonical; columns nobody asked about stay zero);
// then 128 inverse transforms of 64 values (ntt.hip);
// k_vcell_interp: interp[k] = sum_col v[col][k] * h_col^-k,  h_col^-k = roots_of_unity[(8192 - rbl7(col)) k mod 8192].
// cols = [start of column 0 .. 128 in `order` (129 words) | order: the cells' indices grouped by column, ascending inside]
__global__ void __launch_bounds__(256) k_vcell_agg(ff::Fr* __restrict__ agg, const u32* __restrict__ cells,
                                                   const u32* __restrict__ cols, const ff::Fr* __restrict__ pw) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= CELLS_PER_EXT_BLOB * CELL_SIZE) return;
    const u32 col = t >> 6, f = t & 63u;
    ff::Fr acc = ff::Fr::zero();
    const u32* order = cols + CELLS_PER_EXT_BLOB + 1;
    for (u32 j = cols[col]; j < cols[col + 1]; ++j) {
        const size_t i = order[j];
        ff::Fr c;
#pragma unroll
        for (int k = 0; k < 8; ++k) c.v[k] = cells[(i * CELL_SIZE + f) * 8 + k];
        acc = ff::add(acc, fmul(pw[i], c));
    }
    agg[col * CELL_SIZE + (__builtin_bitreverse32(f) >> 26)] = acc;
}
__global__ void __launch_bounds__(128) k_vcell_interp(ff::Fr* __restrict__ out, const ff::Fr* __restrict__ v,
                                                      const ff::Fr* __restrict__ roots8192) {
    __shared__ ff::Fr sh[CELLS_PER_EXT_BLOB];
    const u32 k = blockIdx.x, col = threadIdx.x;
    const u32 rbl = __builtin_bitreverse32(col) >> 25;  // CELL_INDICES_RBL (das.rs:87-96)
    const u32 idx = ((2u * (u32)N - rbl) * k) & (2u * (u32)N - 1u);
    sh[col] = fmul(roots8192[idx], v[col * CELL_SIZE + k]);  // Montgomery x canonical -> canonical
    __syncthreads();
    for (u32 off = CELLS_PER_EXT_BLOB / 2; off > 0; off >>= 1) {
        if (col < off) sh[col] = ff::add(sh[col], sh[col + off]);
        __syncthreads();
    }
    if (col == 0) out[k] = sh[0];
}

// verify_cell_kzg_proof_batch (kzg/src/das.rs:294-389).  Host: parsing, the Fiat-Shamir scalar, the powers of r.
// GPU: the aggregated interpolation polynomial (k_vcell_agg, <= 128 inverse transforms of 64 values, k_vcell_interp;
// compute_commitment_to_aggregated_interpolation_poly, :778-835), decoding + subgroup checks of proofs and commitments, and every linear
// combination as ONE two-row MSM over [proofs | unique commitments | g1_monomial[0..64)]:
//     row 0:  r^i             0          0        -> proof_lincomb
//     row 1:  r^i h_k(i)^64   weight_j   -I_k     -> sum_j w_j C_j - [I(s)] + sum_i r^i h^64 proof_i
// then one pairing check e(row 1, G2) == e(row 0, [s^64]G2) on the host.
void verify_cells(bool* ok, const Bytes48* commitments_bytes, const uint64_t* cell_indices, const Cell* cells,
                  const Bytes48* proofs_bytes, size_t n, const CKZGSettings* cs, KzgAmdSettings* dev) {
    for (size_t i = 0; i < n; ++i) CK_REQUIRE(cell_indices[i] < CELLS_PER_EXT_BLOB, "Invalid cell index");
    // deduplicate_with_indices (das.rs:57-76): first occurrences, in order
    std::vector<Bytes48> uniq;
    std::vector<uint64_t> cidx(n);
    for (size_t i = 0; i < n; ++i) {
        size_t j = 0;
        while (j < uniq.size() && memcmp(uniq[j].bytes, commitments_bytes[i].bytes, 48) != 0) ++j;
        if (j == uniq.size()) uniq.push_back(commitments_bytes[i]);
        cidx[i] = j;
    }
    const size_t m = uniq.size(), np = n + m + CELL_SIZE;
    std::lock_guard<std::mutex> vlk(dev->vmu);
    if (dev->mono64_bytes.empty()) {
        dev->mono64_bytes.resize(CELL_SIZE * 48);
        compress_on_host(dev->mono64_bytes.data(), cs->g1_values_monomial, CELL_SIZE);
    }
    // the 64 setup points are decoded and tested by the first call of a settings object and kept as slots (d_mono64)
    const bool have_mono = dev->d_mono64 != nullptr;
    const size_t ndec = have_mono ? n + m : np;
    std::vector<uint8_t> stage(ndec * 48);
    memcpy(stage.data(), proofs_bytes, n * 48);
    memcpy(stage.data() + n * 48, uniq.data(), m * 48);
    if (!have_mono) memcpy(stage.data() + (n + m) * 48, dev->mono64_bytes.data(), CELL_SIZE * 48);
    decode_points_begin(dev, stage, ndec, dev->d_mono64, have_mono ? CELL_SIZE : 0);
    // host, meanwhile (the decode + subgroup tests are 0.75 ms of GPU latency): the cells' field elements, ...
    std::vector<ff::Fr> cf;
    if (!cells_to_limbs(cf, cells, n)) {
        (void)decode_points_status(dev, np);  // nothing of this call stays in flight
        throw CkErr{C_KZG_BADARGS, "Invalid scalar"};
    }
    const ff::Fr* roots = reinterpret_cast<const ff::Fr*>(cs->roots_of_unity);
    const ff::Fr r = cell_batch_challenge(uniq.data(), m, cidx.data(), cell_indices, cells, proofs_bytes, n);
    std::vector<ff::Fr> sc(2 * np, ff::Fr::zero());
    std::vector<ff::Fr> pws(n);
    std::vector<u32> cols32(CELLS_PER_EXT_BLOB + 1 + n, 0u);  // column starts, then the cells grouped by column (k_vcell_agg)
    for (size_t i = 0; i < n; ++i) ++cols32[(size_t)cell_indices[i] + 1];
    for (size_t c = 0; c < CELLS_PER_EXT_BLOB; ++c) cols32[c + 1] += cols32[c];
    {
        std::vector<u32> cursor(cols32.begin(), cols32.begin() + CELLS_PER_EXT_BLOB);
        for (size_t i = 0; i < n; ++i) cols32[CELLS_PER_EXT_BLOB + 1 + cursor[(size_t)cell_indices[i]]++] = (u32)i;
    }
    ff::Fr pw = ff::Fr::one();
    for (size_t i = 0; i < n; ++i) {
        const size_t col = (size_t)cell_indices[i];
        sc[i] = pw;                                                                     // row 0: proofs
        sc[np + i] = ff::mul(pw, roots[rbl7((u32)col) * CELL_SIZE]);                    // row 1: r^i * h_k^64 (:837-884)
        sc[np + n + cidx[i]] = ff::add(sc[np + n + cidx[i]], pw);                       // row 1: commitment weights (:698-743)
        pws[i] = pw;
        pw = ff::mul(pw, r);
    }
    // the aggregated interpolation polynomial (:778-835) on the GPU: k_vcell_agg, 128 inverse transforms of 64, k_vcell_interp
    std::vector<ff::Fr> interp(CELL_SIZE);
    {
        std::lock_guard<std::mutex> lk(dev->mu);
        kzgamd::DeviceGuard on_device(dev->device);
        CK_HIP(on_device.err);
        dev->ensure_recover();
        dev->ensure_vcells(n);
        if (!dev->d_roots8192) {
            CK_HIP(hipMalloc(&dev->d_roots8192, (2 * N + 1) * sizeof(ff::Fr)));
            CK_HIP(hipMemcpy(dev->d_roots8192, cs->roots_of_unity, (2 * N + 1) * sizeof(ff::Fr), hipMemcpyHostToDevice));
        }
        hipStream_t st = dev->stream;
        CK_HIP(hipMemcpyAsync(dev->d_vc_cells, cf.data(), n * CELL_SIZE * 32, hipMemcpyHostToDevice, st));
        CK_HIP(hipMemcpyAsync(dev->d_vc_cols, cols32.data(), cols32.size() * sizeof(u32), hipMemcpyHostToDevice, st));
        CK_HIP(hipMemcpyAsync(dev->d_vc_pw, pws.data(), n * sizeof(ff::Fr), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_vcell_agg, dim3((unsigned)(CELLS_PER_EXT_BLOB * CELL_SIZE / 256)), dim3(256), 0, st, dev->d_rec[0],
                           (const u32*)dev->d_vc_cells, (const u32*)dev->d_vc_cols, (const ff::Fr*)dev->d_vc_pw);
        if (kzgamd_ntt_fr_device(dev->ntt, dev->d_rec[1], dev->d_rec[0], CELL_SIZE, CELLS_PER_EXT_BLOB, 1, st) != 0)
            throw CkErr{C_KZG_ERROR, "ntt"};
        hipLaunchKernelGGL(k_vcell_interp, dim3((unsigned)CELL_SIZE), dim3((unsigned)CELLS_PER_EXT_BLOB), 0, st, dev->d_rec[2],
                           (const ff::Fr*)dev->d_rec[1], (const ff::Fr*)dev->d_roots8192);
        CK_HIP(hipMemcpyAsync(interp.data(), dev->d_rec[2], CELL_SIZE * sizeof(ff::Fr), hipMemcpyDeviceToHost, st));
        CK_HIP(hipStreamSynchronize(st));
    }
    for (size_t k = 0; k < CELL_SIZE; ++k) interp[k] = ff::to_mont(interp[k]);  // the kernels work on canonical values
    for (size_t k = 0; k < CELL_SIZE; ++k) sc[np + n + m + k] = ff::neg(interp[k]);
    // the MSM next to the membership test of its points, as in verify_g1_finish
    blst_p1 out[2];
    try {
        std::lock_guard<std::mutex> lk(dev->mu);
        kzgamd::DeviceGuard on_device(dev->device);
        CK_HIP(on_device.err);
        CK_HIP(hipEventSynchronize(dev->ev_decoded));
        if (!dev->msm_verify) dev->msm_verify = kzgamd::msm_create(dev->d_vpts, np, true, false, true, kzgamd::G1_TRUSTED, &dev->opt);
        else kzgamd::msm_reset_points(dev->msm_verify, dev->d_vpts, np);
        kzgamd::msm_run_host(dev->msm_verify, out, sc.data(), np, 2);
    } catch (...) {
        (void)decode_points_status(dev, np);  // nothing of this call stays in flight
        throw;
    }
    const std::vector<int> stat = decode_points_status(dev, np);
    for (size_t i = 0; i < np; ++i) CK_REQUIRE(stat[i] != 1, "Invalid G1 encoding");
    for (size_t i = 0; i < n; ++i) CK_REQUIRE(stat[i] == 0, "Proof is not valid");
    for (size_t i = n; i < n + m; ++i) CK_REQUIRE(stat[i] == 0, "Commitment is not valid");
    if (!have_mono) {
        bool mono_ok = true;
        for (size_t i = n + m; i < np; ++i) mono_ok = mono_ok && stat[i] == 0;
        if (mono_ok) {
            std::lock_guard<std::mutex> lk(dev->mu);
            kzgamd::DeviceGuard on_device(dev->device);
            CK_HIP(on_device.err);
            AffPt* keep = nullptr;
            CK_HIP(hipMalloc(&keep, CELL_SIZE * sizeof(AffPt)));
            if (hipMemcpy(keep, dev->d_vpts + n + m, CELL_SIZE * sizeof(AffPt), hipMemcpyDeviceToDevice) == hipSuccess) dev->d_mono64 = keep;
            else (void)hipFree(keep);
        }
    }
    blst_p2 g2gen, g2s64;
    const kzgamd::pairing::G2Jac gen = kzgamd::pairing::g2_generator();
    memcpy(&g2gen, &gen, sizeof g2gen);
    memcpy(&g2s64, &dev->g2_monomial[CELL_SIZE], sizeof g2s64);
    *ok = kzgamd::pairing::pairings_verify(&out[1], &g2gen, &out[0], &g2s64);
}

// compute_vanishing_polynomial_from_roots (das.rs:493-518)
std::vector<ff::Fr> vanishing_from_roots(const std::vector<ff::Fr>& rts) {
    std::vector<ff::Fr> poly;
    poly.push_back(ff::neg(rts[0]));
    for (size_t i = 1; i < rts.size(); ++i) {
        const ff::Fr nr = ff::neg(rts[i]);
        poly.push_back(ff::add(nr, poly[i - 1]));
        for (size_t j = i - 1; j >= 1; --j) poly[j] = ff::add(ff::mul(poly[j], nr), poly[j - 1]);
        poly[0] = ff::mul(poly[0], nr);
    }
    poly.push_back(ff::Fr::one());
    return poly;
}

// recover_cells_and_kzg_proofs (kzg/src/das.rs:101-205; recover_cells :566-657): the five 8192-point transforms, the
// pointwise products, the coset shifts and the inversions on the GPU; the vanishing polynomial of the <= 64 missing
// cells (65 coefficients) on the host.
void recover_cells(Cell* recovered_cells, KZGProof* recovered_proofs, const uint64_t* cell_indices, const Cell* cells,
                   size_t ncells, const CKZGSettings* cs, KzgAmdSettings* dev) {
    std::vector<ff::Fr> cf;
    CK_REQUIRE(cells_to_limbs(cf, cells, ncells), "Invalid scalar");
    CK_REQUIRE(ncells <= CELLS_PER_EXT_BLOB, "Cell length cannot be larger than CELLS_PER_EXT_BLOB");
    CK_REQUIRE(ncells >= CELLS_PER_EXT_BLOB / 2, "Impossible to recover");
    std::vector<char> have(CELLS_PER_EXT_BLOB, 0);
    std::vector<u32> idx32(ncells);
    for (size_t i = 0; i < ncells; ++i) {
        CK_REQUIRE(cell_indices[i] < CELLS_PER_EXT_BLOB, "Invalid cell index");
        if (i + 1 < ncells) CK_REQUIRE(cell_indices[i + 1] > cell_indices[i], "Indices must be in strictly ascending order");
        have[cell_indices[i]] = 1;
        idx32[i] = (u32)cell_indices[i];
    }
    const ff::Fr* roots = reinterpret_cast<const ff::Fr*>(cs->roots_of_unity);
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    dev->ensure(1);
    dev->ensure_cells(1);
    dev->ensure_recover();
    hipStream_t st = dev->stream;
    const size_t E = 2 * N;
    ff::Fr *A = dev->d_rec[0], *B = dev->d_rec[1], *C = dev->d_rec[2], *D = dev->d_rec[3];
    auto ntt = [&](ff::Fr* out, const ff::Fr* in, int inverse) {
        if (kzgamd_ntt_fr_device(dev->ntt, out, in, E, 1, inverse, st) != 0) throw CkErr{C_KZG_ERROR, "ntt"};
    };
    auto mul = [&](ff::Fr* out, const ff::Fr* a, const ff::Fr* b) {
        hipLaunchKernelGGL(k_fr_mul, dim3((unsigned)(E / 256)), dim3(256), 0, st, out, a, b, E);
    };
    // the provided evaluations in bit-reversed order, missing ones zero
    CK_HIP(hipMemsetAsync(A, 0, E * sizeof(ff::Fr), st));
    CK_HIP(hipMemcpyAsync(dev->d_rec_in, cf.data(), ncells * CELL_SIZE * 32, hipMemcpyHostToDevice, st));
    CK_HIP(hipMemcpyAsync(dev->d_rec_idx, idx32.data(), ncells * sizeof(u32), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rec_scatter, dim3((unsigned)((ncells * CELL_SIZE + 255) / 256)), dim3(256), 0, st, A,
                       (const u32*)dev->d_rec_in, (const u32*)dev->d_rec_idx, ncells, ncells != CELLS_PER_EXT_BLOB);
    std::vector<ff::Fr> vanishing;  // must outlive the copy below
    if (ncells != CELLS_PER_EXT_BLOB) {
        // vanishing_polynomial_for_missing_cells (:520-551): roots w^(64 * brp7(i)) for the missing cells i, short
        // polynomial stretched by 64
        std::vector<ff::Fr> rts;
        for (u32 i = 0; i < CELLS_PER_EXT_BLOB; ++i)
            if (!have[i]) rts.push_back(roots[(size_t)rbl7(i) * CELL_SIZE]);
        const std::vector<ff::Fr> shortp = vanishing_from_roots(rts);
        vanishing.assign(E, ff::Fr::zero());
        for (size_t i = 0; i < shortp.size(); ++i) vanishing[i * CELL_SIZE] = shortp[i];
        CK_HIP(hipMemcpyAsync(B, vanishing.data(), E * sizeof(ff::Fr), hipMemcpyHostToDevice, st));
        ntt(C, B, 0);                      // vanishing_poly_eval
        mul(A, A, C);                      // extended_evaluation_times_zero
        ntt(D, A, 1);                      // ..._coeffs
        mul(D, D, dev->d_pow7);            // coset_fft: shift_poly by 7, then the transform
        ntt(A, D, 0);                      // extended_evaluations_over_coset
        mul(B, B, dev->d_pow7);
        ntt(C, B, 0);                      // vanishing_poly_over_coset
        hipLaunchKernelGGL(k_fr_inverse, dim3((unsigned)(E / 64)), dim3(64), 0, st, C, E);
        mul(A, A, C);
        ntt(D, A, 1);                      // coset_ifft: the transform, then shift_poly by 1/7
        mul(D, D, dev->d_pow7inv);         // reconstructed_poly_coeff
        ntt(A, D, 0);                      // its 8192 evaluations, natural order
        hipLaunchKernelGGL(k_cells_out, dim3((unsigned)(E / 256)), dim3(256), 0, st, reinterpret_cast<u32*>(dev->d_fr_ext),
                           (const ff::Fr*)A, (size_t)1);
        CK_HIP(hipMemcpyAsync(recovered_cells, dev->d_fr_ext, E * 32, hipMemcpyDeviceToHost, st));
    } else {
        memcpy(recovered_cells, cells, E * 32);
        if (recovered_proofs) ntt(D, A, 1);  // poly_lagrange_to_monomial of the given cells (:186-188)
    }
    if (recovered_proofs) {
        // compute_fk20_proofs reads the first 4096 coefficients (:190-200, toeplitz_coeffs_stride :659-688)
        if (!dev->d_roots8192) {
            CK_HIP(hipMalloc(&dev->d_roots8192, (2 * N + 1) * sizeof(ff::Fr)));
            CK_HIP(hipMemcpy(dev->d_roots8192, cs->roots_of_unity, (2 * N + 1) * sizeof(ff::Fr), hipMemcpyHostToDevice));
        }
        if (!dev->msm_monomial) dev->msm_monomial = kzgamd::msm_create(dev->d_monomial, N, true, true, true, kzgamd::G1_TRUSTED, &dev->opt);
        dev->ensure_q(1);
        CK_HIP(hipMemcpyAsync(dev->d_fr_b, D, N * sizeof(ff::Fr), hipMemcpyDeviceToDevice, st));
        enqueue_cell_proofs(dev, 1, st, false);
        CK_HIP(hipMemcpyAsync(recovered_proofs, dev->d_proofs, CELLS_PER_EXT_BLOB * 48, hipMemcpyDeviceToHost, st));
    }
    CK_HIP(hipStreamSynchronize(st));
}

}  // namespace

// c_bindings.rs:290-355 -> DAS::verify_cell_kzg_proof_batch (kzg/src/das.rs:294-389)
extern "C" C_KZG_RET verify_cell_kzg_proof_batch(bool* ok, const Bytes48* commitments_bytes, const uint64_t* cell_indices,
                                                 const Cell* cells, const Bytes48* proofs_bytes, uint64_t num_cells,
                                                 const CKZGSettings* s) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (num_cells == 0) {
        *ok = true;
        return C_KZG_OK;
    }
    if (!commitments_bytes || !cell_indices || !cells || !proofs_bytes) return C_KZG_BADARGS;
    return guarded([&] { verify_cells(ok, commitments_bytes, cell_indices, cells, proofs_bytes, (size_t)num_cells, s, dev); });
}

// c_bindings.rs:202-289 -> DAS::recover_cells_and_kzg_proofs (kzg/src/das.rs:101-205); recovered_proofs may be NULL
extern "C" C_KZG_RET recover_cells_and_kzg_proofs(Cell* recovered_cells, KZGProof* recovered_proofs, const uint64_t* cell_indices,
                                                  const Cell* cells, uint64_t num_cells, const CKZGSettings* s) {
    if (!recovered_cells) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (num_cells && (!cell_indices || !cells)) return C_KZG_BADARGS;
    return guarded([&] { recover_cells(recovered_cells, recovered_proofs, cell_indices, cells, (size_t)num_cells, s, dev); });
}

// blst/src/eip_7594.rs:35-97: the Fiat-Shamir scalar of a cell batch (no settings: the inputs are only parsed —
// FsG1::from_bytes accepts any curve point, blst/src/types/g1.rs:65-87 — and hashed)
extern "C" C_KZG_RET compute_verify_cell_kzg_proof_batch_challenge(blst_fr* challenge_out, const Bytes48* commitment_bytes,
                                                                   uint64_t num_commitments, const uint64_t* commitment_indices,
                                                                   const uint64_t* cell_indices, const Cell* cells,
                                                                   const Bytes48* proofs_bytes, uint64_t num_cells) {
    if (!challenge_out) return C_KZG_BADARGS;
    memset(challenge_out, 0, sizeof *challenge_out);
    if ((num_commitments && !commitment_bytes) || (num_cells && (!commitment_indices || !cell_indices || !cells || !proofs_bytes)))
        return C_KZG_BADARGS;
    return guarded([&] {
        for (size_t i = 0; i < num_commitments; ++i) {
            blst_p1 t;
            CK_REQUIRE(kzgamd::host_p1_uncompress(&t, commitment_bytes[i].bytes), "Invalid commitment");
        }
        std::vector<ff::Fr> cf;
        CK_REQUIRE(cells_to_limbs(cf, cells, (size_t)num_cells), "Invalid scalar");
        for (size_t i = 0; i < num_cells; ++i) {
            blst_p1 t;
            CK_REQUIRE(kzgamd::host_p1_uncompress(&t, proofs_bytes[i].bytes), "Invalid proof");
        }
        const ff::Fr r = cell_batch_challenge(commitment_bytes, (size_t)num_commitments, commitment_indices, cell_indices, cells,
                                              proofs_bytes, (size_t)num_cells);
        memcpy(challenge_out, &r, sizeof r);
    });
}

// kzg/src/eth/c_bindings.rs:356-372 (EIP-7594).  cells or proofs may be NULL, not both (das.rs:250-252).
extern "C" C_KZG_RET compute_cells_and_kzg_proofs(Cell* cells, KZGProof* proofs, const Blob* blob, const CKZGSettings* s) {
    if (!blob || (!cells && !proofs)) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    return guarded([&] { cells_and_proofs(cells ? cells->bytes : nullptr, proofs, blob, 1, s, dev); });
}

extern "C" C_KZG_RET kzgamd_compute_cells_and_kzg_proofs_batch(Cell* cells, KZGProof* proofs, const Blob* blobs, size_t n,
                                                               const CKZGSettings* s) {
    if (!blobs || (!cells && !proofs)) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    return guarded([&] { cells_and_proofs(cells ? cells->bytes : nullptr, proofs, blobs, n, s, dev); });
}

