// Host-side planner of the Fr NTT tile kernel (ntt.hip).  No HIP in here: the same code is reachable through
// kzgamd_ntt_plan_dump() on a machine without a GPU, and tests/test_ntt_plan_cpu.py checks it table for table
// against the prototype it was ported from (tools/ntt_plan_sim.py, which also simulates every plan against a direct
// transform).
//
// A pass runs T butterfly stages of the DIT network (fft_fr_fast's recursion unrolled, blst/src/fft_fr.rs:49-108)
// on tiles of 4096 elements.  Tile-local element index idx (12 bits): bits [0, T) are the stage bits (stage s pairs
// idx and idx ^ (1 << s)), bits [T, 12) are column bits (independent sub-problems).  1024 threads = 16 waves x 64
// lanes hold 4 elements each per round.  A round runs two stages (M = 2: idxA, idxA | 1 << pos, idxB = idxA | 2 << pos,
// idxB | 1 << pos) or one (M = 1: two unrelated pairs) or none (M = 0: a copy; T <= 2 is padded to two rounds).
// A phase = consecutive rounds in which every wave keeps the same 256 elements, so that its exchanges go through LDS
// without a workgroup barrier: a 4096-point transform is 6 + 6 stages with ONE barrier.  Per round the planner decides which idx bit every thread-id
// bit stands for: the wave bits are fixed per phase; the lane bits follow the low address bits in the round that
// loads from / stores to global memory (coalescing) and are otherwise chosen so that the 32 lanes of a ds_read_b32
// group hit 32 distinct banks under the XOR swizzle swz().
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace nttplan {

constexpr int LOGT = 12;
constexpr int TILE = 1 << LOGT;
constexpr int NT = TILE / 4;  // threads per workgroup
constexpr int MAXR = 6;       // rounds per pass
constexpr int MAXR_DAS = 12;  // rounds of the fused DAS extension (inverse + forward transform of <= 4096 points in one pass)
enum Kind { KIND_A1 = 0, KIND_A2 = 1, KIND_B = 2 };
// KIND_A1: the whole transform (n <= 4096), 4096 / n contiguous transforms per tile, bit reversal folded into the load
// KIND_A2: first pass of a longer transform: 4096 >> T blocks of 2^T positions of the bit-reversed sequence whose
//          natural-order pieces are neighbours in memory (so that a load instruction reads runs of >= 128 bytes)
// KIND_B : a later pass: 2^T rows x (4096 >> T) consecutive columns, in place

// LDS position of tile element i.  XOR-linear (swz(a ^ b) = swz(a) ^ swz(b)), bits 5..11 pass through, so it is a
// bijection; the seven constants were found by tools/ntt_plan_sim.py's search: with them every round of every plan is
// conflict-free except the first round of KIND_B, T = 4, 5 (2-way).
constexpr uint32_t SWZ_COL[7] = {23, 29, 19, 10, 19, 13, 27};
inline uint32_t swz(uint32_t i) {
    uint32_t x = 0;
    for (int k = 0; k < 7; ++k)
        if ((i >> (5 + k)) & 1) x ^= SWZ_COL[k];
    return i ^ x;
}

struct Round {
    int pos = 0, M = 0, barrier_after = 0, npair = 0, conflicts = 0;
    // fused DAS plans: part 0 = inverse transform, 1 = forward; unit = stages 0 / 1 at position 0 multiply by 1;
    // twist = the round multiplies its results by the per-position factor before they go to LDS
    int part = 0, unit = 0, twist = 0;
    int lane_bits[6] = {0, 0, 0, 0, 0, 0};
    int pair_bits[2] = {0, 0};
    int wave_bits[4] = {0, 0, 0, 0};
};

struct Plan {
    int kind = 0, T = 0, nrounds = 0;
    Round rounds[MAXR_DAS];
    // [round][thread][4] = idxA, idxB, lds(idxA), lds(idxB); the thread's other two elements are idx | bit(round),
    // at LDS positions lds(idx) ^ lds_bit(round) (the position map is XOR-linear)
    std::vector<uint16_t> tab;
    std::vector<uint16_t> sbit;  // lds_bit(round)
    // bit(round): 1 << pos when the round has stages, else the first pair bit
    int elem_bit(int r) const { return rounds[r].M ? rounds[r].pos : rounds[r].pair_bits[0]; }
};

// LDS position (before the swizzle) of DIT position i when the tile holds the data in NATURAL order — the forward half
// of a fused DAS plan reads what the inverse half left: low T bits bit-reversed, column bits unchanged
inline uint32_t brev_pos(uint32_t i, int T) {
    const uint32_t mT = (1u << T) - 1u;
    uint32_t p = i & mT, r = 0;
    for (int k = 0; k < T; ++k)
        if ((p >> k) & 1) r |= 1u << (T - 1 - k);
    return (i & ~mT) | r;
}

namespace detail {
struct Phase {
    int lo, hi;
    int F[8], W[4];
};

inline Phase mk_phase(int lo, int hi, const std::vector<int>& prefer) {
    Phase p;
    p.lo = lo;
    p.hi = hi;
    int nf = 0;
    auto has = [&](int b) {
        for (int i = 0; i < nf; ++i)
            if (p.F[i] == b) return true;
        return false;
    };
    for (int b = lo; b < hi; ++b) p.F[nf++] = b;
    for (int b : prefer)
        if (nf < 8 && !has(b)) p.F[nf++] = b;
    for (int b = 0; b < LOGT; ++b)
        if (nf < 8 && !has(b)) p.F[nf++] = b;
    int nw = 0;
    for (int b = 0; b < LOGT; ++b)
        if (!has(b)) p.W[nw++] = b;
    return p;
}

inline std::vector<Phase> phases_for(int kind, int T) {
    auto range = [](int a, int b) {
        std::vector<int> v;
        for (int i = a; i < b; ++i) v.push_back(i);
        return v;
    };
    if (kind == KIND_A1) {
        if (T <= 8) return {mk_phase(0, T, range(T, 8))};
        return {mk_phase(0, 6, {T - 2, T - 1}), mk_phase(6, T, {})};
    }
    if (T <= 6) return {mk_phase(0, T, range(T, 8))};
    if (kind == KIND_A2) return {mk_phase(0, 6, {T, T + 1}), mk_phase(6, T, {})};
    return {mk_phase(0, 6, {T, T + 1}), mk_phase(6, T, {T, T + 1})};
}

// significance of idx bit `bit` in the global address of the loading (store = false) / storing round
inline int addr_rank(int kind, int T, int bit, bool store) {
    const bool stage = bit < T;
    if (kind == KIND_A1) {
        if (store) return bit;
        return stage ? T - 1 - bit : bit;
    }
    if (kind == KIND_A2) {
        if (store) return stage ? bit : 100 + bit;
        return stage ? 50 + (T - 1 - bit) : bit - T;
    }
    return stage ? 50 + bit : bit - T;
}

// extra LDS cycles of a 32-lane group whose lanes vary the idx bits bits5[0..4]; brevT >= 0: positions through brev_pos
inline int conflicts(const int* bits5, int brevT = -1) {
    int cnt[32] = {0};
    int worst = 0;
    for (int l = 0; l < 32; ++l) {
        uint32_t i = 0;
        for (int k = 0; k < 5; ++k)
            if ((l >> k) & 1) i |= 1u << bits5[k];
        if (brevT >= 0) i = brev_pos(i, brevT);
        const int c = ++cnt[swz(i) & 31];
        worst = std::max(worst, c);
    }
    return worst - 1;
}
}  // namespace detail

// first_io / last_io: the first round loads from / the last stores to global memory (its lane bits follow the
// addresses); brev_lds: LDS positions through brev_pos (the forward half of a fused DAS plan)
inline Plan make_plan(int kind, int T, bool first_io = true, bool last_io = true, bool brev_lds = false) {
    using namespace detail;
    const int bT = brev_lds ? T : -1;
    Plan pl;
    pl.kind = kind;
    pl.T = T;
    std::vector<Phase> ph = phases_for(kind, T);
    const Phase* of_round[MAXR];
    bool last_of_phase[MAXR];
    int n = 0;
    for (const Phase& p : ph) {
        int s = p.lo;
        while (s < p.hi) {
            const int m = s + 2 <= p.hi ? 2 : 1;
            pl.rounds[n].pos = s;
            pl.rounds[n].M = m;
            of_round[n] = &p;
            s += m;
            last_of_phase[n] = s >= p.hi;
            ++n;
        }
    }
    while (n < 2) {  // the kernel peels a loading and a storing round: T <= 2 gets rounds without stages (copies)
        if (n) last_of_phase[n - 1] = false;
        pl.rounds[n].pos = 0;
        pl.rounds[n].M = 0;
        of_round[n] = &ph.back();
        last_of_phase[n] = true;
        ++n;
    }
    pl.nrounds = n;
    for (int r = 0; r < n; ++r) {
        Round& R = pl.rounds[r];
        const Phase& p = *of_round[r];
        for (int k = 0; k < 4; ++k) R.wave_bits[k] = p.W[k];
        std::vector<int> rest;
        for (int k = 0; k < 8; ++k) {
            const int b = p.F[k];
            if (!(b >= R.pos && b < R.pos + R.M)) rest.push_back(b);
        }
        const bool first = r == 0, last = r == n - 1;
        R.barrier_after = (!last && last_of_phase[r]) ? 1 : 0;
        R.npair = (int)rest.size() - 6;
        if ((first && first_io) || (last && last_io)) {
            const bool store = !(first && first_io);
            std::stable_sort(rest.begin(), rest.end(),
                             [&](int a, int b) { return addr_rank(kind, T, a, store) < addr_rank(kind, T, b, store); });
            for (int k = 0; k < 6; ++k) R.lane_bits[k] = rest[k];
            for (int k = 0; k < R.npair; ++k) R.pair_bits[k] = rest[6 + k];
        } else {
            // which of the spare bits tell a thread's elements apart (M = 1) and which lane bit is the half-wave
            // bit: the choice with the fewest bank conflicts, first one found in this order
            int best = 1 << 30;
            const int nrest = (int)rest.size();
            // the sets of npair (0, 1 or 2: M = 2, 1 or 0) bits of `rest`, in lexicographic order
            std::vector<std::pair<int, int>> sets;
            if (R.npair == 0) sets.push_back({-1, -1});
            for (int a = 0; a < nrest && R.npair >= 1; ++a) {
                if (R.npair == 1) sets.push_back({a, -1});
                for (int b = a + 1; b < nrest && R.npair == 2; ++b) sets.push_back({a, b});
            }
            for (const auto& ps : sets) {
                std::vector<int> lanes6;
                for (int k = 0; k < nrest; ++k)
                    if (k != ps.first && k != ps.second) lanes6.push_back(rest[k]);
                for (int h = 0; h < 6; ++h) {
                    int l5[5], m = 0;
                    for (int k = 0; k < 6; ++k)
                        if (k != h) l5[m++] = lanes6[k];
                    const int c = conflicts(l5, bT);
                    if (c < best) {
                        best = c;
                        for (int k = 0; k < 5; ++k) R.lane_bits[k] = l5[k];
                        R.lane_bits[5] = lanes6[h];
                        if (R.npair >= 1) R.pair_bits[0] = rest[ps.first];
                        if (R.npair == 2) R.pair_bits[1] = rest[ps.second];
                    }
                }
            }
        }
        R.conflicts = conflicts(R.lane_bits, bT);
    }
    pl.tab.resize((size_t)n * NT * 4);
    pl.sbit.resize(n);
    auto lds = [&](uint32_t i) { return swz(brev_lds ? brev_pos(i, T) : i); };
    for (int r = 0; r < n; ++r) pl.sbit[r] = (uint16_t)lds(1u << pl.elem_bit(r));
    for (int r = 0; r < n; ++r) {
        const Round& R = pl.rounds[r];
        for (int u = 0; u < NT; ++u) {
            const int l = u & 63, w = u >> 6;
            uint32_t base = 0;
            for (int k = 0; k < 6; ++k)
                if ((l >> k) & 1) base |= 1u << R.lane_bits[k];
            for (int k = 0; k < 4; ++k)
                if ((w >> k) & 1) base |= 1u << R.wave_bits[k];
            uint32_t b2;
            if (R.M == 2) b2 = base | (2u << R.pos);
            else if (R.M == 1) b2 = base | (1u << R.pair_bits[0]);
            else b2 = base | (1u << R.pair_bits[1]);
            uint16_t* t = &pl.tab[((size_t)r * NT + u) * 4];
            t[0] = (uint16_t)base;
            t[1] = (uint16_t)b2;
            t[2] = (uint16_t)lds(base);
            t[3] = (uint16_t)lds(b2);
        }
    }
    return pl;
}

// The DAS extension of lists of 2^T <= 4096 elements in ONE tile pass (data_availability_sampling.rs:14-100 computes
// FFT(w^j IFFT(evens)_j)): the rounds of the inverse transform, whose last round multiplies result j by the twist and
// leaves the tile in LDS in natural order, then the rounds of the forward transform reading it at bit-reversed positions.
inline Plan make_das_plan(int T) {
    const Plan inv = make_plan(KIND_A1, T, true, false, false);
    const Plan fwd = make_plan(KIND_A1, T, false, true, true);
    Plan pl;
    pl.kind = KIND_A1;
    pl.T = T;
    pl.nrounds = inv.nrounds + fwd.nrounds;
    for (int r = 0; r < inv.nrounds; ++r) {
        pl.rounds[r] = inv.rounds[r];
        pl.rounds[r].part = 0;
        pl.rounds[r].unit = r == 0;
        pl.rounds[r].twist = r == inv.nrounds - 1;
        if (pl.rounds[r].twist) pl.rounds[r].barrier_after = 1;
    }
    for (int r = 0; r < fwd.nrounds; ++r) {
        Round& R = pl.rounds[inv.nrounds + r];
        R = fwd.rounds[r];
        R.part = 1;
        R.unit = r == 0;
        R.twist = 0;
    }
    pl.tab = inv.tab;
    pl.tab.insert(pl.tab.end(), fwd.tab.begin(), fwd.tab.end());
    pl.sbit = inv.sbit;
    pl.sbit.insert(pl.sbit.end(), fwd.sbit.begin(), fwd.sbit.end());
    return pl;
}

// stage counts per pass for n = 2^L > 4096: at most 10 stages per pass (>= 4 columns per tile: 128-byte runs), balanced
inline std::vector<int> split_passes(int L) {
    const int np = (L + 9) / 10;
    std::vector<int> v(np, L / np);
    for (int i = 0; i < L % np; ++i) ++v[i];
    return v;
}

}  // namespace nttplan
