// c-kzg-4844 verification (blst/src/eip_4844.rs:383-471 -> kzg/src/eip_4844.rs:328-435, 736-832): the field work and the
// G1 linear combinations on the GPU, one pairing check on the host; and the host-only blst_p2 / pairing exports.
#include "ckzg_shared.h"

namespace {



}  // namespace
namespace ckz {
// verify_kzg_proof_batch (kzg/src/eip_4844.rs:380-435) up to the pairing.  Host: the Fiat-Shamir scalar r
// (compute_r_powers, :328-378) and the three scalar vectors; GPU: decoding + subgroup checks of the 2n points
// (validate_batched_input, :721-734) and the linear combinations — as ONE two-row MSM over [proofs | commitments | G]:
//     row 0:  r^i           0      0                 -> proof_lincomb
//     row 1:  r^i z_i       r^i    -sum r^i y_i      -> rhs  ( = sum r^i (C_i - [y_i]G) + sum r^i z_i proof_i )
// Batched verification, G1 half, in two steps so that the decode + subgroup check of the 2n points (a 1.7 ms latency
// chain on its own stream) runs under whatever the caller does in between — the challenges and evaluations of
// verify_blob_kzg_proof_batch.  The caller holds dev->vmu from begin to finish.
void verify_g1_begin(const Bytes48* commitments, const Bytes48* proofs, size_t n, KzgAmdSettings* dev) {
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    const size_t np = 2 * n + 1;
    dev->ensure_verify(np);
    // device: [proofs | commitments | generator], decoded and checked
    dev->vstage.resize(np * 48);
    memcpy(dev->vstage.data(), proofs, n * 48);
    memcpy(dev->vstage.data() + n * 48, commitments, n * 48);
    static const uint8_t G1_GENERATOR_COMPRESSED[48] = {
        0x97, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f,
        0xc3, 0x68, 0x8c, 0x4f, 0x97, 0x74, 0xb9, 0x05, 0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58,
        0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a, 0xdb, 0x22, 0xc6, 0xbb};
    memcpy(dev->vstage.data() + 2 * n * 48, G1_GENERATOR_COMPRESSED, 48);
    hipStream_t st = dev->stream2;
    CK_HIP(hipMemcpyAsync(dev->d_vbytes, dev->vstage.data(), dev->vstage.size(), hipMemcpyHostToDevice, st));
    CK_HIP(hipMemsetAsync(dev->d_vstat, 0, np * sizeof(int), st));
    CK_HIP(hipMemsetAsync(dev->d_vpts, 0, np * sizeof(AffPt), st));
    if (!dev->ev_decoded) CK_HIP(hipEventCreateWithFlags(&dev->ev_decoded, hipEventDisableTiming));
    decode_check_enqueue(dev->d_vpts, dev->d_vstat, (const unsigned char*)dev->d_vbytes, np, st, dev->cfg_wide_check, dev->ev_decoded);
    CK_HIP(hipGetLastError());
}
}  // namespace ckz
namespace {

}  // namespace
namespace ckz {
void verify_g1_finish(blst_p1* proof_lincomb, blst_p1* rhs, const Bytes48* commitments, const Bytes32* zs, const Bytes32* ys,
                      const Bytes48* proofs, size_t n, KzgAmdSettings* dev) {
    std::vector<ff::Fr> z(n), y(n);
    bool scalars_ok = true;
    for (size_t i = 0; i < n; ++i) {
        scalars_ok = scalars_ok && fr_from_be32_checked(z[i], zs[i].bytes) && fr_from_be32_checked(y[i], ys[i].bytes);
        z[i] = ff::to_mont(z[i]);
        y[i] = ff::to_mont(y[i]);
    }
    std::lock_guard<std::mutex> lk(dev->mu);
    kzgamd::DeviceGuard on_device(dev->device);
    CK_HIP(on_device.err);
    const size_t np = 2 * n + 1;
    hipStream_t st = dev->stream2;
    // host, meanwhile: r = hash_to_bls_field(sha256(domain | 4096 | n | (C_i | z_i | y_i | proof_i)...)), powers of r
    std::vector<ff::Fr> sc(2 * np, ff::Fr::zero());
    {
        kzgamd::Sha256 h;
        uint8_t head[32] = {0};
        memcpy(head, "RCKZGBATCH___V1_", 16);
        const uint64_t nfe = N, nn = n;
        for (int i = 0; i < 8; ++i) {
            head[16 + 7 - i] = (uint8_t)(nfe >> (8 * i));
            head[24 + 7 - i] = (uint8_t)(nn >> (8 * i));
        }
        h.update(head, 32);
        for (size_t i = 0; i < n; ++i) {
            h.update(commitments[i].bytes, 48);
            h.update(zs[i].bytes, 32);
            h.update(ys[i].bytes, 32);
            h.update(proofs[i].bytes, 48);
        }
        uint8_t digest[32];
        h.finish(digest);
        ff::Fr v;
        for (int i = 0; i < 8; ++i) {
            const uint8_t* q = digest + (7 - i) * 4;
            v.v[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
        }
        const ff::Fr r = ff::mul(v, ff::Fr::r2());  // Montgomery form of (v mod r)
        ff::Fr pw = ff::Fr::one(), sy = ff::Fr::zero();
        for (size_t i = 0; i < n; ++i) {
            sc[i] = pw;                          // row 0: proofs
            sc[np + i] = ff::mul(pw, z[i]);      // row 1: proofs
            sc[np + n + i] = pw;                 // row 1: commitments
            sy = ff::add(sy, ff::mul(pw, y[i]));
            pw = ff::mul(pw, r);
        }
        sc[np + 2 * n] = ff::neg(sy);            // row 1: generator
    }
    // The MSM starts as soon as the points are decoded, next to their membership test (0.25 ms on stream2): if a point
    // fails it — or is no encoding at all: its slot stays zero — the sums below are garbage that nobody reads.
    blst_p1 out[2];
    if (scalars_ok) {
        try {
            CK_HIP(hipEventSynchronize(dev->ev_decoded));
            if (!dev->msm_verify) dev->msm_verify = kzgamd::msm_create(dev->d_vpts, np, true, false, true, kzgamd::G1_TRUSTED, &dev->opt);
            else kzgamd::msm_reset_points(dev->msm_verify, dev->d_vpts, np);
            kzgamd::msm_run_host(dev->msm_verify, out, sc.data(), np, 2);
        } catch (...) {
            (void)hipStreamSynchronize(st);  // nothing of this call stays in flight
            throw;
        }
    }
    std::vector<int> stat(np);
    CK_HIP(hipMemcpyAsync(stat.data(), dev->d_vstat, np * sizeof(int), hipMemcpyDeviceToHost, st));
    CK_HIP(hipStreamSynchronize(st));
    CK_REQUIRE(scalars_ok, "Invalid scalar");
    for (size_t i = 0; i < np; ++i) CK_REQUIRE(stat[i] != 1, "Invalid G1 encoding");
    for (size_t i = 0; i < n; ++i) CK_REQUIRE(stat[i] == 0, "Invalid proof");
    for (size_t i = n; i < 2 * n; ++i) CK_REQUIRE(stat[i] == 0, "Invalid commitment");
    *proof_lincomb = out[0];
    *rhs = out[1];
}
}  // namespace ckz
namespace {

void verify_batch_g1(blst_p1* proof_lincomb, blst_p1* rhs, const Bytes48* commitments, const Bytes32* zs, const Bytes32* ys,
                     const Bytes48* proofs, size_t n, KzgAmdSettings* dev) {
    std::lock_guard<std::mutex> vlk(dev->vmu);
    // (the scalars are validated before anything is launched, as before)
    for (size_t i = 0; i < n; ++i) {
        ff::Fr t;
        CK_REQUIRE(fr_from_be32_checked(t, zs[i].bytes) && fr_from_be32_checked(t, ys[i].bytes), "Invalid scalar");
    }
    verify_g1_begin(commitments, proofs, n, dev);
    verify_g1_finish(proof_lincomb, rhs, commitments, zs, ys, proofs, n, dev);
}

}  // namespace

extern "C" C_KZG_RET kzgamd_verify_kzg_proof_batch_g1(blst_p1* proof_lincomb_out, blst_p1* rhs_out, const Bytes48* commitments,
                                                      const Bytes32* zs, const Bytes32* ys, const Bytes48* proofs, size_t n,
                                                      const CKZGSettings* s) {
    if (!proof_lincomb_out || !rhs_out) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) {
        memset(proof_lincomb_out, 0, sizeof *proof_lincomb_out);
        memset(rhs_out, 0, sizeof *rhs_out);
        return C_KZG_OK;
    }
    if (!commitments || !zs || !ys || !proofs) return C_KZG_BADARGS;
    return guarded([&] { verify_batch_g1(proof_lincomb_out, rhs_out, commitments, zs, ys, proofs, n, dev); });
}

// verify_blob_kzg_proof_batch (kzg/src/eip_4844.rs:736-832) up to the pairing: challenges + evaluations on the GPU
// (:690-719), then the G1 half above.  The caller finishes with  e(proof_lincomb, [tau]G2) == e(rhs, G2).
extern "C" C_KZG_RET kzgamd_verify_blob_kzg_proof_batch_g1(blst_p1* proof_lincomb_out, blst_p1* rhs_out, const Blob* blobs,
                                                           const Bytes48* commitments, const Bytes48* proofs, size_t n,
                                                           const CKZGSettings* s) {
    if (!proof_lincomb_out || !rhs_out) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) {
        memset(proof_lincomb_out, 0, sizeof *proof_lincomb_out);
        memset(rhs_out, 0, sizeof *rhs_out);
        return C_KZG_OK;
    }
    if (!blobs || !commitments || !proofs) return C_KZG_BADARGS;
    return guarded([&] {
        std::vector<Bytes32> zs(n), ys(n);
        std::lock_guard<std::mutex> vlk(dev->vmu);
        verify_g1_begin(commitments, proofs, n, dev);  // decode + subgroup check run under the evaluations
        try {
            prove_batch(nullptr, ys.data(), blobs, nullptr, commitments, n, dev, zs.data(), true);
        } catch (...) {
            (void)hipStreamSynchronize(dev->stream2);
            throw;
        }
        verify_g1_finish(proof_lincomb_out, rhs_out, commitments, zs.data(), ys.data(), proofs, n, dev);
    });
}

namespace {

// check_proof_single (blst/src/types/kzg_settings.rs:178-196) on decoded, validated inputs.  The reference tests
//     e(C - [y]G, G2) == e(proof, [tau]G2 - [z]G2);
// with the [z] moved to the G1 side (bilinearity; the proof is a checked r-torsion point) the same statement is
//     e(C - [y]G + [z]proof, G2) == e(proof, [tau]G2),
// which pairs with the two fixed G2 points of the setup only: their line tables are cached (host_pairing.h), and the
// G2 scalar multiplication becomes a G1 one.  One pairing-product check on the host (the reference keeps the pairing
// on the CPU too).
bool check_proof_single(const blst_p1& commitment, const blst_p1& proof, const ff::Fr& z_plain, const ff::Fr& y_plain,
                        KzgAmdSettings* dev) {
    using namespace kzgamd::pairing;
    kzgamd::HostJac g;
    {
        const uint64_t GX[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull,
                                0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
        const uint64_t GY[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull,
                                0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
        for (int k = 0; k < 6; ++k) {
            g.x.v[2 * k] = (u32)GX[k];
            g.x.v[2 * k + 1] = (u32)(GX[k] >> 32);
            g.y.v[2 * k] = (u32)GY[k];
            g.y.v[2 * k + 1] = (u32)(GY[k] >> 32);
        }
        g.z = ff::Fp::one();
    }
    g.y = hfp::neg(g.y);  // -G
    kzgamd::HostJac pi;
    memcpy(&pi, &proof, sizeof pi);
    // [z]proof - [y]G: one joint double-and-add over the 255 bits
    kzgamd::HostJac acc;
    acc.x = acc.y = acc.z = ff::Fp::zero();
    for (int bit = 254; bit >= 0; --bit) {
        acc = kzgamd::host_jac_dbl(acc);
        if ((y_plain.v[bit >> 5] >> (bit & 31)) & 1) acc = kzgamd::host_jac_add(acc, g);
        if ((z_plain.v[bit >> 5] >> (bit & 31)) & 1) acc = kzgamd::host_jac_add(acc, pi);
    }
    kzgamd::HostJac c;
    memcpy(&c, &commitment, sizeof c);
    const kzgamd::HostJac lhs = kzgamd::host_jac_add(c, acc);
    blst_p1 a1;
    memcpy(&a1, &lhs, sizeof a1);
    const G2Jac g2gen = g2_generator();
    blst_p2 a2, b2;
    memcpy(&a2, &g2gen, sizeof a2);
    memcpy(&b2, &dev->g2_monomial[1], sizeof b2);
    return pairings_verify(&a1, &a2, &proof, &b2);
}

// FsG1::from_bytes + the is_inf / is_valid test of verify_kzg_proof_rust (kzg/src/eip_4844.rs:603-608)
void decode_valid_g1(blst_p1& out, const uint8_t* bytes, const char* what) {
    CK_REQUIRE(kzgamd::host_p1_uncompress(&out, bytes), std::string("Invalid ") + what);
    CK_REQUIRE(kzgamd::host_p1_in_g1(&out), std::string("Invalid ") + what);
}

}  // namespace

// blst/src/eip_4844.rs:383-405 -> verify_kzg_proof_raw (kzg/src/eip_4844.rs:613-637)
extern "C" C_KZG_RET verify_kzg_proof(bool* ok, const Bytes48* commitment_bytes, const Bytes32* z_bytes, const Bytes32* y_bytes,
                                      const Bytes48* proof_bytes, const CKZGSettings* s) {
    if (!ok || !commitment_bytes || !z_bytes || !y_bytes || !proof_bytes) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    return guarded([&] {
        blst_p1 c, pr;
        ff::Fr z, y;
        CK_REQUIRE(kzgamd::host_p1_uncompress(&c, commitment_bytes->bytes), "Invalid commitment");
        CK_REQUIRE(fr_from_be32_checked(z, z_bytes->bytes), "Invalid scalar");
        CK_REQUIRE(fr_from_be32_checked(y, y_bytes->bytes), "Invalid scalar");
        CK_REQUIRE(kzgamd::host_p1_uncompress(&pr, proof_bytes->bytes), "Invalid proof");
        CK_REQUIRE(kzgamd::host_p1_in_g1(&c), "Invalid commitment");
        CK_REQUIRE(kzgamd::host_p1_in_g1(&pr), "Invalid proof");
        *ok = check_proof_single(c, pr, z, y, dev);
    });
}

// blst/src/eip_4844.rs:410-430 -> verify_blob_kzg_proof_raw (kzg/src/eip_4844.rs:667-688): challenge and evaluation
// on the GPU, one pairing check on the host
extern "C" C_KZG_RET verify_blob_kzg_proof(bool* ok, const Blob* blob, const Bytes48* commitment_bytes,
                                           const Bytes48* proof_bytes, const CKZGSettings* s) {
    if (!ok || !blob || !commitment_bytes || !proof_bytes) return C_KZG_BADARGS;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    return guarded([&] {
        blst_p1 c, pr;
        CK_REQUIRE(host_blob_valid(blob->bytes), "Invalid scalar");           // bytes_to_blob
        // the two points are decoded and subgroup-checked (0.3 ms of one core each) on two helper threads while this
        // one hashes the challenge and the GPU evaluates the polynomial
        bool c_ok = false, pr_ok = false;
        std::thread tc, tp;
        struct Joiner {  // declared before the threads start: an exception (also out of the second thread's creation)
            std::thread &a, &b;  // must not leave a joinable thread behind
            ~Joiner() {
                if (a.joinable()) a.join();
                if (b.joinable()) b.join();
            }
        } joiner{tc, tp};
        tc = std::thread([&] {
            c_ok = kzgamd::host_p1_uncompress(&c, commitment_bytes->bytes) && kzgamd::host_p1_in_g1(&c);  // infinity passes
        });
        tp = std::thread([&] {
            pr_ok = kzgamd::host_p1_uncompress(&pr, proof_bytes->bytes) && kzgamd::host_p1_in_g1(&pr);
        });
        Bytes32 zb, yb;
        {
            LaneRef lane(dev, 1);
            prove_batch(nullptr, &yb, blob, nullptr, commitment_bytes, 1, lane.use, &zb, true);
        }
        tc.join();
        tp.join();
        CK_REQUIRE(c_ok, "Invalid commitment");
        CK_REQUIRE(pr_ok, "Invalid proof");
        ff::Fr z, y;
        CK_REQUIRE(fr_from_be32_checked(z, zb.bytes) && fr_from_be32_checked(y, yb.bytes), "Invalid scalar");
        *ok = check_proof_single(c, pr, z, y, dev);
    });
}

// blst/src/eip_4844.rs:435-471 -> verify_blob_kzg_proof_batch_raw (kzg/src/eip_4844.rs:736-866): n == 0 is true,
// n == 1 the single verification, otherwise challenges, evaluations and the three linear combinations on the GPU
// and ONE pairing check e(sum r^i proof_i, [tau]G2) == e(rhs, G2) on the host
extern "C" C_KZG_RET verify_blob_kzg_proof_batch(bool* ok, const Blob* blobs, const Bytes48* commitments_bytes,
                                                 const Bytes48* proofs_bytes, size_t n, const CKZGSettings* s) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;
    KzgAmdSettings* dev = lookup(s);
    if (!dev) return C_KZG_BADARGS;
    if (n == 0) {
        *ok = true;
        return C_KZG_OK;
    }
    if (!blobs || !commitments_bytes || !proofs_bytes) return C_KZG_BADARGS;
    if (n == 1) return verify_blob_kzg_proof(ok, blobs, commitments_bytes, proofs_bytes, s);
    return guarded([&] {
        blst_p1 pl, rhs;
        std::vector<Bytes32> zs(n), ys(n);
        {
            std::lock_guard<std::mutex> vlk(dev->vmu);
            verify_g1_begin(commitments_bytes, proofs_bytes, n, dev);  // decode + subgroup check run under the evaluations
            try {
                prove_batch(nullptr, ys.data(), blobs, nullptr, commitments_bytes, n, dev, zs.data(), true);
            } catch (...) {
                (void)hipStreamSynchronize(dev->stream2);  // the decode kernel reads dev->vstage's device copy: drain it
                throw;
            }
            verify_g1_finish(&pl, &rhs, commitments_bytes, zs.data(), ys.data(), proofs_bytes, n, dev);
        }
        blst_p2 g2gen, g2tau;
        const kzgamd::pairing::G2Jac gen = kzgamd::pairing::g2_generator();
        memcpy(&g2gen, &gen, sizeof g2gen);
        memcpy(&g2tau, &dev->g2_monomial[1], sizeof g2tau);
        *ok = kzgamd::pairing::pairings_verify(&pl, &g2tau, &rhs, &g2gen);
    });
}



// ---- host-only helpers over blst_p2 / the pairing (no GPU needed): what a binding test-suite or a caller that
// wants to finish kzgamd_verify_*_g1 itself uses.  pairings_verify = blst/src/kzg_proofs.rs:73-100.
extern "C" int kzgamd_pairings_verify(const blst_p1* a1, const blst_p2* a2, const blst_p1* b1, const blst_p2* b2) {
    if (!a1 || !a2 || !b1 || !b2) return -1;
    return kzgamd::pairing::pairings_verify(a1, a2, b1, b2) ? 1 : 0;
}
extern "C" int kzgamd_p2_uncompress(blst_p2* out, const uint8_t in[96]) {
    kzgamd::pairing::G2Jac p;
    if (!out || !in || !kzgamd::pairing::g2_uncompress(p, in)) return 1;
    memcpy(out, &p, sizeof p);
    return 0;
}
extern "C" void kzgamd_p2_compress(uint8_t out[96], const blst_p2* in) {
    kzgamd::pairing::G2Jac p;
    memcpy(&p, in, sizeof p);
    kzgamd::pairing::g2_compress(out, p);
}
extern "C" void kzgamd_p2_generator(blst_p2* out) {
    const kzgamd::pairing::G2Jac g = kzgamd::pairing::g2_generator();
    memcpy(out, &g, sizeof g);
}
extern "C" void kzgamd_p2_mult(blst_p2* out, const blst_p2* in, const blst_fr* scalar_mont) {
    kzgamd::pairing::G2Jac p;
    memcpy(&p, in, sizeof p);
    ff::Fr k;
    memcpy(&k, scalar_mont, 32);
    k = ff::from_mont(k);
    const kzgamd::pairing::G2Jac r = kzgamd::pairing::g2_mul(p, k.v);
    memcpy(out, &r, sizeof r);
}
extern "C" void kzgamd_p2_add(blst_p2* out, const blst_p2* a, const blst_p2* b) {
    kzgamd::pairing::G2Jac x, y;
    memcpy(&x, a, sizeof x);
    memcpy(&y, b, sizeof y);
    const kzgamd::pairing::G2Jac r = kzgamd::pairing::g2_add(x, y);
    memcpy(out, &r, sizeof r);
}

