// Throughput of single-blob host-buffer calls from T native threads sharing one CKZGSettings — the reference's rayon
// pattern (kzg/src/eip_4844.rs:781-805) without an interpreter lock in the way.
// Build: g++ -O2 -std=c++17 tools/concurrent_bench.cpp -Iinclude -Lrust-kzg_amd/csrc -lkzg_mi355x -lpthread -o tools/concurrent_bench
// Run:   LD_LIBRARY_PATH=rust-kzg_amd/csrc tools/concurrent_bench tests/golden/trusted_setup.txt [seconds]
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "kzg_mi355x.h"
#include "../rust-kzg_amd/csrc/host_g1.h"  // host_p1_compress: results are compared as group elements

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const double secs = argc > 2 ? atof(argv[2]) : 1.5;
    FILE* f = fopen(argv[1], "r");
    if (!f) return 2;
    CKZGSettings s;
    if (load_trusted_setup_file(&s, f) != C_KZG_OK) return 3;
    fclose(f);
    const int NB = 16;
    std::vector<Blob> blobs(NB);
    std::mt19937_64 rng(4);
    for (auto& b : blobs) {
        for (size_t i = 0; i < sizeof b.bytes; i += 8) {
            uint64_t v = rng();
            memcpy(b.bytes + i, &v, 8);
        }
        for (size_t i = 0; i < sizeof b.bytes; i += 32) b.bytes[i] = 0;
    }
    std::vector<KZGCommitment> cm(NB);
    for (int i = 0; i < NB; ++i)
        if (blob_to_kzg_commitment(&cm[i], &blobs[i], &s) != C_KZG_OK) return 4;
    // reference results of the serial calls: every concurrent result is compared with them
    std::vector<KZGProof> pr(NB);
    for (int i = 0; i < NB; ++i)
        if (compute_blob_kzg_proof(&pr[i], &blobs[i], &cm[i], &s) != C_KZG_OK) return 4;
    // B1 state: built when the first b1 measurement starts (see prepare_b1 below)
    const size_t NP = 4096;
    void* msm = nullptr;
    std::vector<std::vector<blst_fr>> sc;
    std::vector<std::array<uint8_t, 48>> want(NB);
    int b1_rc = 0;
    // One prepared handle over the setup's Lagrange points, shared by the threads (what rust-kzg's g1_lincomb does with
    // SpparkPrecomputation, kzg/src/msm/sppark.rs:24-44).  Created AFTER the commitment / proof legs of a run: its streams
    // (the handle's own, two combining lanes) would otherwise shift which hardware queues the settings object's lane
    // streams land on, and the c-kzg legs measured beside it came out a quarter slower (25 k against 31 k commitments/s).
    auto prepare_b1 = [&]() -> int {
        if (msm) return 0;
        std::vector<blst_p1_affine> aff(NP);
        for (size_t i = 0; i < NP; ++i) {
            const blst_p1* P = reinterpret_cast<const blst_p1*>(&s.g1_values_lagrange_brp[i]);
            aff[i].x = P->x;  // the setup's points are affine (Z = 1 in Montgomery form)
            aff[i].y = P->y;
        }
        // its table next to the settings object's (and, under bench.py, the parent process's): an explicit budget instead
        // of "whatever is free", which would leave the lanes of the settings object nothing to allocate their workspaces from
        KzgAmdConfig mcfg;
        kzgamd_config_init(&mcfg);
        mcfg.table_budget_bytes = getenv("B1_TABLE_GB") ? (uint64_t)(atof(getenv("B1_TABLE_GB")) * 1e9) : 24000000000ull;
        msm = kzgamd_prepare_msm_ex(aff.data(), NP, &mcfg);
        if (!msm) return 5;
        sc.assign(NB, std::vector<blst_fr>(NP));
        for (int i = 0; i < NB; ++i) {
            for (auto& x : sc[i]) {
                for (int k = 0; k < 4; ++k) x.l[k] = rng();
                x.l[3] &= 0x3fffffffffffffffull;  // below r: a valid Montgomery representative of some scalar
            }
            blst_p1 out;
            RustError e = mult_pippenger_prepared(msm, &out, NP, sc[i].data());
            if (e.code) return 6;
            kzgamd::host_p1_compress(want[i].data(), &out);
        }
        return 0;
    };
    // B2: one NTT handle shared by the threads (FFTSettings by reference from rayon workers): ntt_fr of 4096 elements
    const size_t NTT_N = 4096;
    void* ntt = nullptr;
    std::vector<std::vector<blst_fr>> ntt_in, ntt_want;
    auto prepare_b2 = [&]() -> int {
        if (ntt) return 0;
        ntt = kzgamd_ntt_new(13);
        if (!ntt) return 7;
        ntt_in.assign(NB, std::vector<blst_fr>(NTT_N));
        ntt_want.assign(NB, std::vector<blst_fr>(NTT_N));
        for (int i = 0; i < NB; ++i) {
            for (auto& x : ntt_in[i]) {
                for (int k = 0; k < 4; ++k) x.l[k] = rng();
                x.l[3] &= 0x3fffffffffffffffull;
            }
            if (ntt_fr(ntt, ntt_want[i].data(), ntt_in[i].data(), NTT_N, 0) != 0) return 7;
        }
        return 0;
    };
    long errors = 0;  // failed calls + results that differ from the serial ones
    printf("{");
    bool first = true;
    std::vector<int> Ts = {1, 2, 4, 8, 16, 32};
    if (argc > 3) Ts = {atoi(argv[3])};
    const int only = argc > 4 ? atoi(argv[4]) : -1;  // 0 = commitments only, 1 = proofs only
    for (int T : Ts) {
        for (int what = 0; what < 4; ++what) {
            if (what == 3 && only != 3) continue;  // the B2 leg only on request
            if (only >= 0 && what != only) continue;
            if (what == 2 && (b1_rc = prepare_b1()) != 0) return b1_rc;
            if (what == 3 && prepare_b2() != 0) return 7;
            std::atomic<bool> stop{false};
            std::atomic<long> total{0};
            std::atomic<int> bad{0};
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    long n = 0;
                    KZGCommitment c;
                    KZGProof p;
                    const int i = t % NB;
                    while (!stop.load(std::memory_order_relaxed)) {
                        if (what == 3) {
                            std::vector<blst_fr> o(NTT_N);
                            if (ntt_fr(ntt, o.data(), ntt_in[i].data(), NTT_N, 0) != 0 ||
                                memcmp(o.data(), ntt_want[i].data(), NTT_N * sizeof(blst_fr)) != 0)
                                bad.fetch_add(1);
                            ++n;
                            continue;
                        }
                        if (what == 2) {
                            blst_p1 out;
                            uint8_t got[48];
                            RustError e = mult_pippenger_prepared(msm, &out, NP, sc[i].data());
                            if (e.code) {
                                free(e.message);
                                bad.fetch_add(1);
                            } else {
                                kzgamd::host_p1_compress(got, &out);
                                if (memcmp(got, want[i].data(), 48) != 0) bad.fetch_add(1);
                            }
                            ++n;
                            continue;
                        }
                        C_KZG_RET rc = what == 0 ? blob_to_kzg_commitment(&c, &blobs[i], &s)
                                                 : compute_blob_kzg_proof(&p, &blobs[i], &cm[i], &s);
                        if (rc != C_KZG_OK || memcmp(what == 0 ? c.bytes : p.bytes, what == 0 ? cm[i].bytes : pr[i].bytes, 48) != 0)
                            bad.fetch_add(1);
                        ++n;
                    }
                    total.fetch_add(n);
                });
            auto t0 = std::chrono::steady_clock::now();
            std::this_thread::sleep_for(std::chrono::duration<double>(secs));
            stop.store(true);
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%s\"%s_threads_%d\": %.0f", first ? "" : ", ", what == 0 ? "commit" : what == 1 ? "proof" : what == 2 ? "b1_prepared" : "b2_ntt_fr_4096", T, total.load() / dt);
            first = false;
            errors += bad.load();
        }
    }
    printf(", \"failed_or_different_from_the_serial_results\": %ld}\n", errors);
    if (msm) free_msm(msm);
    if (ntt) kzgamd_ntt_free(ntt);
    free_trusted_setup(&s);
    return 0;
}
