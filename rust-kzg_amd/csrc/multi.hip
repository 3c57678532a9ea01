// Multi-GPU from inside the library: one process, one settings object per GPU, whole blobs in contiguous
// slabs per device, one host thread per device, results written in place.  No Python, no torch.distributed, no
// collective: blobs are independent units and every device holds its own replica of the fixed-base table.
//
// This is the shape of the reference's own callers: ONE process that parallelises inside itself over groups of blobs
// (kzg/src/eip_4844.rs:770-816, rayon par_chunks), sharing one precomputation handle (kzg/src/msm/sppark.rs:24-44);
// its GPU path is single-device (arkworks3-sppark-wlc/sppark/msm/pippenger.cuh:573-575).  Here the groups are the
// devices.  Host-only code: everything goes through the single-device entry points of include/kzg_mi355x.h (the one
// HIP call is hipHostRegister in kzgamd_pin_host_buffer).
#include "../../include/kzg_mi355x.h"

#include <hip/hip_runtime.h>  // kzgamd_pin_host_buffer only: everything else goes through the entry points
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// contiguous slab [lo, hi) of part `k` of `parts`; sizes differ by at most one; empty slabs allowed
// (the same partition as rust-kzg_amd/sharding.py::shard_range, so that both multi-GPU forms agree on who owns a blob)
inline void shard_range(size_t n, size_t parts, size_t k, size_t& lo, size_t& hi) {
    const size_t base = n / parts, extra = n % parts;
    lo = k * base + (k < extra ? k : extra);
    hi = lo + base + (k < extra ? 1 : 0);
}

// fn(k) for every k < parts: part 0 on the calling thread, the others on a thread each (a device's pipeline blocks its
// host thread in stream synchronisation, so the threads cost no cores).  Every entry point they call selects its
// handle's GPU by itself (DeviceGuard), so the threads need no hipSetDevice.  Returns the worst code:
// BADARGS > MALLOC > ERROR > OK — the reference maps every failure to BadArgs (blst/src/utils.rs:47-56).
template <class F>
C_KZG_RET fan_out(size_t parts, F&& fn) {
    std::vector<C_KZG_RET> rc(parts, C_KZG_OK);
    std::vector<std::thread> th;
    bool spawn_failed = false;
    try {
        th.reserve(parts);
        for (size_t k = 1; k < parts; ++k)
            th.emplace_back([&, k] {
                try {
                    rc[k] = fn(k);
                } catch (...) {
                    rc[k] = C_KZG_BADARGS;
                }
            });
    } catch (...) {
        spawn_failed = true;  // out of threads: what was not started runs here, in turn
    }
    const size_t started = th.size() + 1;
    try {
        rc[0] = fn(0);
        if (spawn_failed)
            for (size_t k = started; k < parts; ++k) rc[k] = fn(k);
    } catch (...) {
        rc[0] = C_KZG_BADARGS;
    }
    for (auto& t : th) t.join();
    C_KZG_RET worst = C_KZG_OK;
    auto rank = [](C_KZG_RET r) { return r == C_KZG_BADARGS ? 3 : r == C_KZG_MALLOC ? 2 : r == C_KZG_ERROR ? 1 : 0; };
    for (C_KZG_RET r : rc)
        if (rank(r) > rank(worst)) worst = r;
    return worst;
}

bool settings_ok(const CKZGSettings* const s[], size_t ndev) {
    if (!s || ndev == 0) return false;
    for (size_t d = 0; d < ndev; ++d)
        if (!s[d] || kzgamd_settings_device(s[d]) < 0) return false;
    return true;
}

}  // namespace

extern "C" int kzgamd_shard_range(size_t n, size_t parts, size_t k, size_t* lo, size_t* hi) {
    if (parts == 0 || k >= parts || !lo || !hi) return 1;
    shard_range(n, parts, k, *lo, *hi);
    return 0;
}

extern "C" C_KZG_RET kzgamd_load_trusted_setup_file_multi(CKZGSettings out[], const int devices[], size_t ndev, FILE* in) {
    return kzgamd_load_trusted_setup_file_multi_ex(out, devices, ndev, in, nullptr);
}

extern "C" C_KZG_RET kzgamd_load_trusted_setup_file_multi_ex(CKZGSettings out[], const int devices[], size_t ndev, FILE* in,
                                                             const KzgAmdConfig* cfg) {
    if (!out) return C_KZG_BADARGS;
    for (size_t d = 0; d < ndev; ++d) memset(&out[d], 0, sizeof out[d]);
    if (!in || ndev == 0) return C_KZG_BADARGS;
    const int avail = kzgamd_device_count();
    for (size_t d = 0; d < ndev; ++d) {
        const int dev = devices ? devices[d] : (int)d;
        if (dev < 0 || dev >= avail) return C_KZG_BADARGS;
    }
    // the text is read once (at most 1 MiB like load_trusted_setup_file, blst/src/eip_4844.rs:244-246) and parsed per device
    std::vector<char> text(1024 * 1024);
    const size_t len = fread(text.data(), 1, text.size(), in);
    if (len == 0) return C_KZG_BADARGS;
    // one loader thread per device: decompression, table build (0.8 s) and the first-use state run side by side
    const C_KZG_RET rc = fan_out(ndev, [&](size_t d) -> C_KZG_RET {
        // the caller's configuration (budget per table, tuning) with the device of this entry
        KzgAmdConfig mine;
        kzgamd_config_init(&mine);
        if (cfg) {
            // field by field: a caller compiled against a later, longer struct is accepted like everywhere else (the
            // single-device loader validates struct_size and the tuning string)
            if (cfg->struct_size < sizeof mine) return C_KZG_BADARGS;
            mine.struct_size = cfg->struct_size;
            mine.table_budget_bytes = cfg->table_budget_bytes;
            mine.tuning = cfg->tuning;
        }
        mine.device = devices ? devices[d] : (int)d;
        FILE* f = fmemopen(text.data(), len, "r");
        C_KZG_RET r = C_KZG_MALLOC;
        if (f) {
            r = kzgamd_load_trusted_setup_file_ex(&out[d], f, &mine);
            fclose(f);
        }
        return r;
    });
    if (rc != C_KZG_OK)
        for (size_t d = 0; d < ndev; ++d) free_trusted_setup(&out[d]);  // all or nothing (free of an empty object is a no-op)
    return rc;
}

// Page-lock a caller's buffer (blobs, commitments, results) so that the batch entry points' copies to and from it are
// true DMA on every device instead of going through the runtime's staging of pageable memory — one CPU pass over the data
// less per device.  On one GPU the runtime's pageable path (≈ 45 GB/s) is not the limit (profiles/NOTES.md §9); with eight
// devices fed from one process the staging copies of eight host threads share the host's memory bandwidth.
extern "C" int kzgamd_pin_host_buffer(void* p, size_t bytes) {
    if (!p || !bytes) return 1;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? 0 : 1;
}
extern "C" int kzgamd_unpin_host_buffer(void* p) {
    if (!p) return 1;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? 0 : 1;
}

extern "C" void kzgamd_free_trusted_setup_multi(CKZGSettings s[], size_t ndev) {
    if (!s) return;
    for (size_t d = 0; d < ndev; ++d) free_trusted_setup(&s[d]);
}

extern "C" C_KZG_RET kzgamd_blob_to_kzg_commitment_batch_multi(KZGCommitment* out, const Blob* blobs, size_t n,
                                                               const CKZGSettings* const s[], size_t ndev) {
    if (!out || !blobs || !settings_ok(s, ndev)) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    const size_t parts = ndev < n ? ndev : n;  // never more slabs than blobs
    return fan_out(parts, [&](size_t k) -> C_KZG_RET {
        size_t lo, hi;
        shard_range(n, parts, k, lo, hi);
        return kzgamd_blob_to_kzg_commitment_batch(out + lo, blobs + lo, hi - lo, s[k]);
    });
}

extern "C" C_KZG_RET kzgamd_compute_blob_kzg_proof_batch_multi(KZGProof* out, const Blob* blobs, const Bytes48* commitments,
                                                               size_t n, const CKZGSettings* const s[], size_t ndev) {
    if (!out || !blobs || !commitments || !settings_ok(s, ndev)) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    const size_t parts = ndev < n ? ndev : n;
    return fan_out(parts, [&](size_t k) -> C_KZG_RET {
        size_t lo, hi;
        shard_range(n, parts, k, lo, hi);
        return kzgamd_compute_blob_kzg_proof_batch(out + lo, blobs + lo, commitments + lo, hi - lo, s[k]);
    });
}

extern "C" C_KZG_RET kzgamd_compute_cells_and_kzg_proofs_batch_multi(Cell* cells, KZGProof* proofs, const Blob* blobs, size_t n,
                                                                     const CKZGSettings* const s[], size_t ndev) {
    if (!blobs || (!cells && !proofs) || !settings_ok(s, ndev)) return C_KZG_BADARGS;
    if (n == 0) return C_KZG_OK;
    const size_t parts = ndev < n ? ndev : n;
    return fan_out(parts, [&](size_t k) -> C_KZG_RET {
        size_t lo, hi;
        shard_range(n, parts, k, lo, hi);
        return kzgamd_compute_cells_and_kzg_proofs_batch(cells ? cells + 128 * lo : nullptr, proofs ? proofs + 128 * lo : nullptr,
                                                         blobs + lo, hi - lo, s[k]);
    });
}

// verify_blob_kzg_proof_batch over several GPUs: the reference cuts a large batch into groups, verifies each group
// on its own and ANDs the verdicts (kzg/src/eip_4844.rs:770-816) — the groups here are the devices' slabs.
extern "C" C_KZG_RET kzgamd_verify_blob_kzg_proof_batch_multi(bool* ok, const Blob* blobs, const Bytes48* commitments,
                                                              const Bytes48* proofs, size_t n, const CKZGSettings* const s[],
                                                              size_t ndev) {
    if (!ok || !settings_ok(s, ndev)) return C_KZG_BADARGS;
    *ok = false;
    if (n == 0) {
        *ok = true;
        return C_KZG_OK;
    }
    if (!blobs || !commitments || !proofs) return C_KZG_BADARGS;
    const size_t parts = ndev < n ? ndev : n;
    std::vector<char> good(parts, 0);
    const C_KZG_RET rc = fan_out(parts, [&](size_t k) -> C_KZG_RET {
        size_t lo, hi;
        shard_range(n, parts, k, lo, hi);
        bool v = false;
        const C_KZG_RET r = verify_blob_kzg_proof_batch(&v, blobs + lo, commitments + lo, proofs + lo, hi - lo, s[k]);
        good[k] = v ? 1 : 0;
        return r;
    });
    if (rc != C_KZG_OK) return rc;
    bool all = true;
    for (char g : good) all = all && g != 0;
    *ok = all;
    return C_KZG_OK;
}

// One large MSM sharded by index range (second bullet): handle d was prepared over
// points[offsets[d], offsets[d + 1]); every device computes the partial sum over its slice of the scalars and the
// ndev 144-byte partials are added on the host (kzgamd_g1_sum) — the "exchange step" is ndev x 144 bytes.
extern "C" RustError kzgamd_mult_pippenger_prepared_multi(void* const msm[], size_t ndev, blst_p1* out, const size_t offsets[],
                                                          const blst_fr scalars[]) {
    auto fail = [](const char* what) {
        RustError e;
        e.code = 1;
        e.message = (char*)malloc(strlen(what) + 1);
        if (e.message) strcpy(e.message, what);
        return e;
    };
    if (!msm || ndev == 0 || !out || !offsets || !scalars) return fail("kzgamd_mult_pippenger_prepared_multi: null argument");
    for (size_t d = 0; d < ndev; ++d)
        if (offsets[d + 1] < offsets[d] || (!msm[d] && offsets[d + 1] != offsets[d]))  // an empty slice needs no handle
            return fail("kzgamd_mult_pippenger_prepared_multi: bad handle or offsets");
    std::vector<blst_p1> part(ndev);
    std::vector<RustError> errs(ndev);
    for (auto& e : errs) e = RustError{0, nullptr};
    (void)fan_out(ndev, [&](size_t d) -> C_KZG_RET {
        const size_t cnt = offsets[d + 1] - offsets[d];
        if (cnt == 0) {
            memset(&part[d], 0, sizeof part[d]);  // empty slice: the point at infinity (Z == 0)
            return C_KZG_OK;
        }
        errs[d] = mult_pippenger_prepared(msm[d], &part[d], cnt, scalars + offsets[d]);
        return errs[d].code == 0 ? C_KZG_OK : C_KZG_BADARGS;
    });
    RustError first{0, nullptr};
    for (size_t d = 0; d < ndev; ++d) {
        if (errs[d].code != 0 && first.code == 0) first = errs[d];
        else if (errs[d].message) free(errs[d].message);
    }
    if (first.code != 0) return first;
    kzgamd_g1_sum(out, part.data(), ndev);
    return first;
}
