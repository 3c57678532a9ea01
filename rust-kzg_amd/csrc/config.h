// Configuration of a handle: what the caller passes in a KzgAmdConfig (include/kzg_mi355x.h) resolved into the values
// the engines read.  One table holds every tuning key of the library (name, default, range, meaning) — DESIGN.md §9 is
// generated from the same list (kzgamd_tuning_keys) — and the only environment variables the library itself reads are
//   KZGAMD_TUNING       "key=value;key=value": the same string a caller puts in KzgAmdConfig.tuning (measurement tools)
//   KZGAMD_FBW_MAX_GB   table budget per fixed-base table, when the caller passed none
//   KZGAMD_VERBOSE      say on stderr what shape a prepared handle ended up with
//   KZGAMD_DEBUG        say on stderr why a c-kzg entry point returned C_KZG_BADARGS
// Order of precedence: defaults < environment < KzgAmdConfig.  Values are read ONCE, when a handle is created.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/kzg_mi355x.h"

namespace kzgamd {

enum TuneId {
    // wide-table (fixed-base) path of the MSM
    T_SPL, T_NO_WIDE_TAIL, T_NO_HYBRID_FOLD, T_NO_WIDE_TREE, T_QUAD_ACCUM_MAX, T_HYBRID_MAX, T_WIDE_FOLD_MAX, T_SPL1_MAX, T_BLOCKSUM_THREADS,
    // bucket engine
    T_LGC, T_GROUPS, T_TAIL_PIECES, T_SUB_STREAMS, T_SUB_PRIO, T_SUB_LARGE, T_SORT_AHEAD, T_TILE_ROWS, T_TILE_QUAD, T_DIGIT_MIN_LOG, T_FINE_BITS, T_ONE_LEVEL_SORT, T_TREE_TAIL, T_FLAT_DIGITS, T_DIRECT_SCATTER, T_SCATTER_ATOMICS,
    // shape of a handle
    T_WINDOW, T_WINDOW_PREPARED, T_FIXED_AS_VARIABLE_MIN, T_GLV, T_FBW_GLV,
    // concurrent host-buffer callers of one prepared handle (B1)
    T_COMBINE, T_COMBINE_LANES, T_COMBINE_GATHER_MIN, T_COMBINE_GATHER_US,
    // G1 transforms: which stage form by the number of half-butterflies
    T_G1_WIDE_MAX, T_G1_QUAD_MAX, T_G1_PAIR_MAX,
    // c-kzg surface
    T_FK20, T_DEVICE_SHA, T_SHA_LANES, T_HOST_CHECK_MAX, T_WIDE_CHECK, T_PROVE_CHUNK, T_PROVE_FIRST, T_COMMIT_CHUNK, T_COMMIT_FIRST,
    T_LEADERS, T_GATHER_MIN, T_GATHER_US,
    T_COUNT
};

struct TuneKey {
    const char* name;
    long dflt, lo, hi;
    const char* what;
};

inline const TuneKey* tune_keys() {
    static const TuneKey k[T_COUNT] = {
        {"spl", 0, 0, 16, "scalars per lane of the wide-table accumulation (0 = by batch size; 1, 2, 4, 8, 16)"},
        {"no_wide_tail", 0, 0, 1, "1: single-lane instead of limb-parallel tails and folds"},
        {"no_hybrid_fold", 0, 0, 1, "1: two launches of k_blocksum instead of k_blocksum_hybrid for 5..16 MSMs per call"},
        {"no_wide_tree", 0, 0, 1, "1: a lone commitment's partial sums folded by two launches of k_wide_fold64 instead of the one-launch tree k_wide_tree"},
        {"quad_accum_max", 4, 0, 64, "commitments per call whose accumulation runs four lanes per (scalar, half) chain (k_fbw_accum_quad; 0 = never; only calls that get a lane per (scalar, half), see spl1_max)"},
        {"hybrid_max", 0, 0, 1 << 20, "MSMs per call folded by k_blocksum_hybrid (0 = built-in)"},
        {"wide_fold_max", 0, 0, 1 << 20, "MSMs per call folded limb-parallel by k_wide_tree (0 = built-in: 4)"},
        {"spl1_max", 0, 0, 1 << 20, "MSMs per call that get a lane per (scalar, half) (0 = 8)"},
        {"blocksum_threads", 0, 0, 256, "threads of k_blocksum: 64, 128 or 256 (0 = by batch size)"},
        {"lgc", 0, 0, 8, "log2 of the accumulation chunk of the bucket engine (0 = by size; 2 ... 8)"},
        {"groups", 0, 0, 4, "window groups of the bucket engine on their own streams (0 = one)"},
        {"tail_pieces", 0, 0, 4, "a single large MSM accumulates its window sets in this many pieces, the reduction of a piece beside the accumulation of the next (0, 1 = one piece; measured: nothing gained)"},
        {"sub_streams", 6, 0, 6, "a batch of large MSMs runs as sub-batches whose accumulations stay on the caller's stream while the sort and the reduction chains of each run on one of this many high-priority side streams (own workspaces) beside another sub-batch's accumulation (0 = one stream, one workspace); MSMs below 2^18 points unless sub_large = 1"},
        {"sub_prio", 1, 0, 1, "1: the side streams of sub_streams are created with the device's highest stream priority, 0: with the default priority"},
        {"sub_large", 0, 0, 1, "1: the side streams of sub_streams also for batches of MSMs of 2^18 points and more (measured slower: nothing runs well beside a chip-filling accumulation)"},
        {"sort_ahead", 0, 0, 1, "1: a batch of MSMs of 2^18 points and more runs as sub-batches whose SORT runs on one of two side streams beside the limb-parallel chains of the previous sub-batch's reduction (after its tile sums; nothing beside an accumulation); measured +-1 %: off"},
        {"tile_rows", 0, 0, 32, "rows of 32 buckets per tile of the tiled bucket reduction: 32 (a 512-lane workgroup, a CU each) or 16 (256 lanes, a wave per SIMD); 0 = 32"},
        {"tile_quad", 1, 0, 1, "1: the tree levels of the tiled bucket reduction that keep at most a quarter of the lanes busy run four lanes per addition (an addition 4 multiplications deep instead of 14); 0: one lane per addition throughout"},
        {"digit_min_log", 14, 10, 30, "log2 of the smallest bucket count per set that takes the digit-decomposed (tiled) bucket reduction instead of the (sum, weighted sum) tree (round 6 measured 10 ... 13 on 2^10 ... 2^18 points: the tree is faster below 2^14 buckets)"},
        {"fine_bits", 0, 0, 10, "width of the second sort level (0 = default; 7 ... 10)"},
        {"one_level_sort", 0, 0, 1, "1: the one-level sort at every size (it is the form small bucket counts take anyway)"},
        {"tree_tail", 0, 0, 1, "1: the tree reduction at every size (the form of fewer than 16384 buckets)"},
        {"flat_digits", 0, 0, 1, "1: untiled digit sums (the form bucket counts that are no multiple of 1024 take)"},
        {"direct_scatter", 0, 0, 1, "1: scatter without the LDS staging (the form of very long scalars per tile)"},
        {"scatter_atomics", 0, 0, 1, "1: ranks from global atomics instead of the kept histogram"},
        {"window", 0, 0, 22, "window bits of a variable-base handle (0 = by size; 2 ... 22)"},
        {"window_prepared", 0, 0, 22, "window bits of a prepared handle: bucket engine over table rows, no wide table (0 = by size; 2 ... 22)"},
        {"fixed_as_variable_min", 19, 0, 39, "log2 of the smallest prepared handle that, without room for a wide table, runs the variable-base engine (0 = never)"},
        {"glv", 1, 0, 1, "0: no endomorphism split in the variable-base engine"},
        {"fbw_glv", 1, 0, 1, "0: the wide table never takes the GLV form (rows over 128-bit halves), whatever it would save"},
        {"combine", 1, 0, 1, "0: concurrent mult_pippenger_prepared / ntt_fr / das_fft_extension calls on one handle queue on its mutex, one launch each"},
        {"combine_lanes", 3, 1, 4, "batches of combined calls in flight at once"},
        {"combine_gather_min", 6, 1, 32, "with a batch in flight, wait for this many queued calls ..."},
        {"combine_gather_us", 60, 0, 100000, "... but at most this long (microseconds)"},
        {"g1_wide_max", 4096, 0, 1L << 40, "G1 stages of up to this many half-butterflies run a wave each"},
        {"g1_quad_max", 16384, 0, 1L << 40, "... up to this many, four lanes each"},
        {"g1_pair_max", 32768, 0, 1L << 40, "... up to this many, two lanes each; above, one lane each"},
        {"fk20", -1, -1, 1, "cell proofs by FK20 (1), by one fixed-base MSM per cell (0), or by batch size (-1)"},
        {"device_sha", 0, 0, 1, "1: Fiat-Shamir SHA-256 of host-buffer batches on the GPU"},
        {"sha_lanes", 0, 0, 4, "lanes per blob of the device Fiat-Shamir hash: 4 (the message schedules of four blocks side by side, the compression chain 0.6 as long, 2.4 x the instructions), 1, or 0 = 4 up to 512 blobs per launch and 1 above"},
        {"host_check_max", 64, 0, 1 << 20, "commitments of a proof batch up to this many are validated on the host pool"},
        {"wide_check", 1, 0, 1, "0: single-lane commitment checks"},
        {"prove_chunk", 0, 0, 1 << 20, "blobs per pipelined chunk of a large proof batch (0 = by size)"},
        {"prove_first", 0, 0, 1 << 20, "blobs of its first chunk (0 = by size)"},
        {"commit_chunk", 0, 0, 1 << 20, "blobs per pipelined chunk of a large commitment batch (0 = by size)"},
        {"commit_first", 0, 0, 1 << 20, "blobs of its first chunk (0 = by size)"},
        {"leaders", 3, 1, 64, "batches of coalesced single-blob calls in flight per settings object"},
        {"gather_min", 6, 1, 16, "with a batch in flight, wait for this many queued single-blob calls ..."},
        {"gather_us", 60, 0, 100000, "... but at most this long (microseconds)"},
    };
    return k;
}

// values inside a key's range that no engine would honour: rejected like a value outside it
inline bool tune_value_allowed(int id, long v) {
    switch (id) {
        case T_SPL: return v == 0 || v == 1 || v == 2 || v == 4 || v == 8 || v == 16;
        case T_BLOCKSUM_THREADS: return v == 0 || v == 64 || v == 128 || v == 256;
        case T_FINE_BITS: return v == 0 || (v >= 7 && v <= 10);
        case T_LGC: return v == 0 || (v >= 2 && v <= 8);
        case T_WINDOW:
        case T_WINDOW_PREPARED: return v == 0 || (v >= 2 && v <= 22);
        case T_SHA_LANES: return v == 0 || v == 1 || v == 4;
        case T_TILE_ROWS: return v == 0 || v == 16 || v == 32;
        default: return true;
    }
}

struct Options {
    int device = -1;               // -1: the calling thread's current device
    double table_budget_gb = -1;   // per fixed-base table; < 0: 160 GB, capped by what is free
    long t[T_COUNT];
    Options() {
        const TuneKey* k = tune_keys();
        for (int i = 0; i < T_COUNT; ++i) t[i] = k[i].dflt;
    }
    // "key=value;key=value" (also ',' or whitespace as separators); false with *err set on an unknown key, a missing or
    // non-numeric value or a value outside the key's range
    bool parse(const char* str, std::string* err) {
        if (!str) return true;
        const TuneKey* k = tune_keys();
        const char* p = str;
        while (*p) {
            while (*p == ';' || *p == ',' || *p == ' ' || *p == '\t' || *p == '\n') ++p;
            if (!*p) break;
            const char* e = p;
            while (*e && *e != '=' && *e != ';' && *e != ',' && *e != ' ') ++e;
            const std::string name(p, e);
            if (*e != '=') {
                if (err) *err = "tuning: '" + name + "' has no value";
                return false;
            }
            char* end = nullptr;
            const long v = strtol(e + 1, &end, 10);
            if (end == e + 1) {
                if (err) *err = "tuning: '" + name + "' has a non-numeric value";
                return false;
            }
            int id = -1;
            for (int i = 0; i < T_COUNT; ++i)
                if (name == k[i].name) id = i;
            if (id < 0) {
                if (err) *err = "tuning: unknown key '" + name + "'";
                return false;
            }
            if (v < k[id].lo || v > k[id].hi || !tune_value_allowed(id, v)) {
                if (err) *err = "tuning: '" + name + "' out of range";
                return false;
            }
            t[id] = v;
            p = end;
        }
        return true;
    }
    // keys that do not combine: each is a form of the same stretch of the engine (how a batch, or a lone MSM, is cut)
    bool consistent(std::string* err) const {
        if (t[T_SORT_AHEAD] && (t[T_GROUPS] > 1 || t[T_TAIL_PIECES] > 1 || t[T_SUB_LARGE])) {
            if (err) *err = "tuning: sort_ahead does not combine with groups, tail_pieces or sub_large";
            return false;
        }
        return true;
    }
    // defaults < environment < cfg; false (with *err) when a string does not parse, the keys contradict each other or cfg
    // is malformed
    static bool resolve(Options& o, const KzgAmdConfig* cfg, std::string* err) {
        o = Options();
        if (!o.parse(getenv("KZGAMD_TUNING"), err)) return false;
        if (const char* e = getenv("KZGAMD_FBW_MAX_GB")) o.table_budget_gb = atof(e);
        if (!cfg) return o.consistent(err);
        if (cfg->struct_size < offsetof(KzgAmdConfig, tuning) + sizeof(cfg->tuning)) {
            if (err) *err = "KzgAmdConfig.struct_size is not that of any version of the struct";
            return false;
        }
        o.device = cfg->device;
        if (cfg->table_budget_bytes == KZGAMD_NO_TABLES) o.table_budget_gb = 0;
        else if (cfg->table_budget_bytes) o.table_budget_gb = (double)cfg->table_budget_bytes / 1e9;
        return o.parse(cfg->tuning, err) && o.consistent(err);
    }
};

}  // namespace kzgamd
