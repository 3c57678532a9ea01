// MI355X-native Pippenger MSM over BLS12-381 G1 — kernels + host driver behind the
// sppark-compatible boundary (include/kzg_mi355x.h, B1).
//
// Shape of the computation (one "set" = one bucket array):
//   prepared handle : fixed-base table rows  T[j][i] = 2^(c*j) * P_i  (built once in prepare_msm,
//                     the reference's BGMW idea, kzg/src/msm/bgmw.rs:206-227) so that all
//                     ceil(255/c) signed windows of a scalar share ONE bucket set and the result
//                     needs no doublings.  nsets = nbatch.
//   unprepared call : classic windowed Pippenger, one bucket set per window, Horner at the end.
//                     nsets = windows.
//   k_digits   : scalar -> signed c-bit digits, per-bucket histogram (global atomics)
//   k_scan     : exclusive scan of the histogram per set
//   k_scatter  : counting-sort of (point index | sign) by bucket
//   k_accum    : one lane per 16 sorted entries, chain of XYZZ mixed adds, pieces per (bucket, chunk)
//   k_heavy    : one wave per over-full bucket: combine its pieces
//   k_level    : bucket reduction sum (k+1)*B_k as an 8-ary tree of (plain sum, weighted sum) pairs
//   k_final    : Horner over windows (unprepared), convert to blst Jacobian
// All arithmetic is integer VALU (v_mad_u64_u32); there is no MFMA-shaped work here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "g1_io.hip.h"
#include "g1w.hip.h"
#include "g1grp.hip.h"
#include "glv.hip.h"
#include "host_g1.h"
#include "config.h"
#include "device_guard.h"
#include "msm_internal.h"

using ff::u32;
using ff::u64;
using g1::AffPt;
using g1::WidePt;
using g1::Xyzz;

namespace {

constexpr int GRP = 8;      // children folded per lane in the bucket-reduction tree
constexpr u32 HEAVY = 512;  // entries above which a bucket's pieces are combined by a whole wave
constexpr double FBW_DEFAULT_GB = 160.0;  // default HBM budget of one wide fixed-base table
// sorted entries per k_accum lane ("chunk", a power of two passed as its log2): 32 for large MSMs (fewer pieces
// per bucket for the reduction tree: n = 2^20 5.47 -> 5.38 ms, 2^22 19.2 -> 17.7 ms), 16 below 2^18, where the
// accumulation is a latency chain of `chunk` mixed additions

// ---------------------------------------------------------------- helpers
struct HipErr {
    hipError_t e;
    const char* what;
};
#define HIP_TRY(x)                                  \
    do {                                            \
        hipError_t _e = (x);                        \
        if (_e != hipSuccess) throw HipErr{_e, #x}; \
    } while (0)

RustError ok_error() { return RustError{0, nullptr}; }
RustError make_error(int code, const std::string& msg) {
    char* m = (char*)malloc(msg.size() + 1);
    if (m) memcpy(m, msg.c_str(), msg.size() + 1);
    return RustError{code, m};
}

// ---------------------------------------------------------------- kernels

// cube root of unity beta (Montgomery 2^392): phi(x, y) = (beta*x, y) = -[x^2](x, y) on G1, x the BLS parameter
// (the relation the subgroup check in ckzg.hip tests)
__device__ __forceinline__ fp28::Fe beta28() {
    constexpr u32 t[14] = {0xa75929au, 0x681b798u, 0x22a3e9du, 0xabc02bfu, 0x4e5bb45u, 0x55e6e7eu, 0x4814117u,
                           0x6d04f1bu, 0xae3387du, 0x54acb0cu, 0xa4c74bu, 0x56138b5u, 0xb64e066u, 0x76f2u};
    fp28::Fe b;
#pragma unroll
    for (int k = 0; k < 14; ++k) b.v[k] = t[k];
    return b;
}

// [x^2]P = (beta*x, -y): the second base of the GLV split k = k1 + k2*x^2 (k_digits)
__device__ __forceinline__ AffPt x2_image(const AffPt& a) {
    AffPt o = a;
    if (!(a.flags & 1)) {
        o.x = fp28::canon(fp28::mul(a.x, beta28()));
        o.y = fp28::canon(fp28::neg<2>(a.y));
    }
    return o;
}

// blst affine (2 x 12 u32, Montgomery 2^384) -> table slot (fp28, Montgomery 2^392);
// glv: slot n + i receives [x^2]P_i
__global__ void __launch_bounds__(256) k_points_in(AffPt* __restrict__ dst, const ff::Fp* __restrict__ src, size_t n,
                                                   int glv) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ff::Fp x = src[2 * i], y = src[2 * i + 1];
    AffPt o;
    o.flags = (x.is_zero() && y.is_zero()) ? 1u : 0u;
    o.pad[0] = o.pad[1] = o.pad[2] = 0;
    // canonical residues, so that y and 2p - y are both valid normalized inputs
    o.x = fp28::canon(fp28::from_blst(x));
    o.y = fp28::canon(fp28::from_blst(y));
    dst[i] = o;
    if (glv) dst[n + i] = x2_image(o);
}

// table rows j = 1..rows-1:  T[j][i] = 2^c * T[j-1][i], kept affine (one inversion per entry)
__global__ void __launch_bounds__(128) k_table_rows(AffPt* __restrict__ table, size_t n, int rows, int c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    AffPt cur = table[i];
    for (int j = 1; j < rows; ++j) {
        AffPt nxt;
        nxt.flags = cur.flags;
        nxt.pad[0] = nxt.pad[1] = nxt.pad[2] = 0;
        if (cur.flags & 1) {
            nxt.x = fp28::zero();
            nxt.y = fp28::zero();
        } else {
            Xyzz acc;
            g1::dbl_affine(acc, cur.x, cur.y);
            bool inf = false;
            for (int k = 1; k < c; ++k) {
                if (fp28::is_zero_mod_p(acc.y)) {  // order-2 point: doubling gives infinity
                    inf = true;
                    break;
                }
                g1::dbl(acc);
            }
            if (inf || g1::is_inf(acc)) {
                nxt.flags = 1;
                nxt.x = fp28::zero();
                nxt.y = fp28::zero();
            } else {
                fp28::Fe zi = g1io::inverse(fp28::mul(acc.zz, acc.zzz));  // 1/(ZZ*ZZZ)
                fp28::Fe izz = fp28::mul(zi, acc.zzz), izzz = fp28::mul(zi, acc.zz);
                nxt.x = fp28::canon(fp28::mul(acc.x, izz));
                nxt.y = fp28::canon(fp28::mul(acc.y, izzz));
            }
        }
        table[(size_t)j * n + i] = nxt;
        cur = nxt;
    }
}

struct DigitParams {
    size_t n;        // points per MSM
    size_t nbatch;   // MSMs in this launch
    int c;           // window bits
    int nwin;        // windows per scalar
    int prepared;    // 1: all windows share one bucket set
    int mont;        // scalars are Montgomery blst_fr
    size_t nb;       // buckets per set = 2^(c-1)
    size_t row_stride;  // prepared: points per table row; glv: offset of the [x^2]P half of the table
    int glv;         // 1: scalars are split k = k1 + k2*x^2 (two 128-bit halves, nwin windows each)
    int w0, w1;      // windows [w0, w1) are emitted by this launch (the others only feed the digit carry)
    u32 nseg;        // wide-table path: MSM b runs over table bases (b % nseg) * n .. + n (0: every MSM over bases 0 .. n)
};

// canonical 256-bit scalar (8 x u32) -> signed digit of window w, carrying from below
__device__ __forceinline__ void load_scalar(u32 s[8], const u32* __restrict__ scalars, size_t idx, int mont) {
    ff::Fr f;
#pragma unroll
    for (int k = 0; k < 8; ++k) f.v[k] = scalars[idx * 8 + k];
    if (mont) f = ff::from_mont(f);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = f.v[k];
}

__device__ __forceinline__ u32 window_bits(const u32 s[8], int bit, int c) {
    // c <= 24 bits starting at `bit` (bits above 255 read as zero)
    int w = bit >> 5, sh = bit & 31;
    u64 two = 0;
    if (w < 8) two = s[w];
    if (w + 1 < 8) two |= (u64)s[w + 1] << 32;
    return (u32)(two >> sh) & ((1u << c) - 1);
}

// pass 0: histogram — the value the atomic returns is the entry's rank inside its bucket, kept in `ranks`
// (one word per (part, window, scalar), coalesced); pass 1: scatter to offsets[bucket] + rank with no atomics at
// all (recomputes the digits instead of storing them)
template <int PASS>
__global__ void __launch_bounds__(256) k_digits(DigitParams P, const u32* __restrict__ scalars,
                                                const AffPt* __restrict__ pts, u32* __restrict__ counts,
                                                const u32* __restrict__ offsets, u32* __restrict__ sorted,
                                                size_t set_cap, u32* __restrict__ ranks) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.n * P.nbatch) return;
    size_t b = t / P.n, i = t % P.n;
    if (pts[i].flags & 1) return;  // infinity base contributes nothing (rows share the flag)
    u32 s[8], s2[8];
    load_scalar(s, scalars, t, P.mont);
    u32 pneg[2] = {0, 0};
    if (P.glv) {
        u32 k[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) k[q] = s[q];
        kzgamd::glv_split(k, s, s2, pneg[0], pneg[1]);
    }
    const u32 half = 1u << (P.c - 1);
    const int lane = threadIdx.x & 63;
    const u64 lt_mask = ((u64)1 << lane) - 1;
    const u32 cmask = (1u << P.c) - 1u;
    for (int part = 0; part <= P.glv; ++part) {
        // the scalar (half) in registers, shifted down one window at a time: static indices only (a dynamic index
        // into the words is scratch traffic)
        u32 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = part ? s2[q] : s[q];
        u32 carry = 0;
        for (int w = 0; w < P.w1; ++w) {
            u32 d = (v[0] & cmask) + carry;
#pragma unroll
            for (int x = 0; x < 7; ++x) v[x] = (v[x] >> P.c) | (v[x + 1] << (32 - P.c));
            v[7] >>= P.c;
            u32 neg = 0;
            carry = 0;
            if (d > half) {
                d = (1u << P.c) - d;
                neg = 1;
                carry = 1;
            }
            if (w < P.w0) continue;
            bool todo = d != 0;
            const size_t set = P.prepared ? b : b * P.nwin + w;
            const size_t slot = set * P.nb + (todo ? d - 1 : 0);
            u32* rank_slot = ranks + ((size_t)(part * P.nwin + w) * P.nbatch * P.n + t);
            if (PASS == 1) {
                if (todo) {
                    const u32 pos = offsets[set * (P.nb + 1) + (d - 1)] + *rank_slot;
                    const u32 pidx = P.prepared ? (u32)((size_t)w * P.row_stride + i) : (u32)(part ? P.row_stride + i : i);
                    sorted[set * set_cap + pos] = pidx | ((neg ^ pneg[part]) << 31);
                }
                continue;
            }
            u32 rank = 0;  // position inside the bucket
            // Lanes of a wave that hit the same bucket share one atomic: a short top window, the carry-only
            // window and blobs of equal elements put thousands of entries on one counter, and same-address
            // atomics serialise.  Groups of fewer than 4 lanes are left to the plain per-lane atomic below, and
            // the search stops after two such groups, so uniformly spread digits pay two ballots and no extra
            // atomic round trip.
            u64 pending = __ballot(todo);
            for (int round = 0, misses = 0; round < 8 && pending && misses < 2; ++round) {
                const int leader = __ffsll((unsigned long long)pending) - 1;
                const u32 lo = __shfl((u32)slot, leader, 64), hi = __shfl((u32)(slot >> 32), leader, 64);
                const bool same = todo && (u32)slot == lo && (u32)(slot >> 32) == hi;
                const u64 m = __ballot(same);
                pending &= ~m;
                if (__popcll(m) < 4) {
                    ++misses;
                    continue;
                }
                u32 base = 0;
                if (lane == leader) base = atomicAdd(&counts[slot], (u32)__popcll(m));
                base = __shfl(base, leader, 64);
                if (same) {
                    rank = base + (u32)__popcll(m & lt_mask);
                    todo = false;
                }
            }
            const bool mine = d != 0;
            if (todo) rank = atomicAdd(&counts[slot], 1u);
            if (mine) *rank_slot = rank;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Two-level sort (default for one MSM up to n = 2^22).  The one-level sort above pays two global atomics' worth
// of L2 traffic per entry (histogram with return value, then a scattered 4-byte write): 1.2 ms at n = 2^20.
// Here a bucket id is split into a coarse bin (bucket >> fb) and a fine bucket (low fb bits; fb = 7, tuning key fine_bits
// = 7..10 for experiments: fewer, larger bins make k_part_scatter's runs longer and the kernel faster — 169 vs 267 us at
// n = 2^20 with fb = 10 — but k_bin_sort then has 256 workgroups of 65536 entries and loses the same time again;
// 7, 8, 9 and 10 are within noise of each other end to end):
//   k_part_count   : a workgroup counts its 1024 scalars' entries per coarse bin in LDS (<= 4096 bins) and adds
//                    the non-zero counters to the global bin counts — ~8x fewer global atomics, all LDS otherwise
//   k_part_scan    : exclusive scan of the bin counts
//   k_part_scatter : same count again in LDS, one returning atomic per (workgroup, bin) reserves a run inside the
//                    bin, entries are written there as  fine << (32 - fb) | sign << (31 - fb) | point index
//   k_bin_sort     : one workgroup per coarse bin streams its run twice: LDS histogram of the 2^fb fine buckets ->
//                    bucket offsets (+ heavy flags), then an LDS counting sort into the final order
// Entries keep the format the accumulation expects (point index | sign << 31).
constexpr int FINE_BITS_MIN = 7, FINE_BITS_MAX = 10;
constexpr u32 FINE_MAX = 1u << FINE_BITS_MAX;
constexpr u32 MAX_BINS = 4096;     // coarse bins per launch (LDS counters)
constexpr int PART_SCALARS = 4;    // scalars per lane in the partition kernels (1024 per workgroup)

// calls emit(set, bucket, sign, point index) for every non-zero digit of scalar t in windows [w0, w1)
template <class F>
__device__ __forceinline__ void scalar_entries(const DigitParams& P, const u32* __restrict__ scalars,
                                               const AffPt* __restrict__ pts, size_t t, F emit) {
    const size_t b = t / P.n, i = t % P.n;
    if (pts[i].flags & 1) return;  // infinity base contributes nothing (rows share the flag)
    u32 s[8], s2[8];
    load_scalar(s, scalars, t, P.mont);
    u32 pneg[2] = {0, 0};
    if (P.glv) {
        u32 k[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) k[q] = s[q];
        kzgamd::glv_split(k, s, s2, pneg[0], pneg[1]);
    }
    // the digits come off the low end of a register copy that is shifted down one window at a time: every index is
    // static (a window addressed by its bit position would index the words dynamically and push them to scratch)
    const u32 half = 1u << (P.c - 1), cmask = (1u << P.c) - 1u;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        if (part > P.glv) break;
        u32 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = part ? s2[q] : s[q];
        u32 carry = 0;
        for (int w = 0; w < P.w1; ++w) {
            u32 d = (v[0] & cmask) + carry;
#pragma unroll
            for (int q = 0; q < 7; ++q) v[q] = (v[q] >> P.c) | (v[q + 1] << (32 - P.c));
            v[7] >>= P.c;
            u32 neg = 0;
            carry = 0;
            if (d > half) {
                d = (1u << P.c) - d;
                neg = 1;
                carry = 1;
            }
            if (w < P.w0 || d == 0) continue;
            const size_t set = P.prepared ? b : b * P.nwin + w;
            const u32 pidx = P.prepared ? (u32)((size_t)w * P.row_stride + i) : (u32)(part ? P.row_stride + i : i);
            emit(set, d - 1, neg ^ pneg[part], pidx);
        }
    }
}

// set0 = first set of this launch's group, cb = coarse bins per set; bins are numbered (set - set0) * cb + coarse
__global__ void __launch_bounds__(256) k_part_count(DigitParams P, const u32* __restrict__ scalars,
                                                    const AffPt* __restrict__ pts, u32* __restrict__ bin_count, u32 set0,
                                                    u32 cb, u32 nbins, int fb, u32* __restrict__ wg_hist) {
    extern __shared__ u32 lds_bins[];
    for (u32 k = threadIdx.x; k < nbins; k += 256) lds_bins[k] = 0;
    __syncthreads();
    const size_t total = P.n * P.nbatch;
#pragma unroll 1
    for (int q = 0; q < PART_SCALARS; ++q) {
        const size_t t = ((size_t)blockIdx.x * PART_SCALARS + q) * 256 + threadIdx.x;
        if (t < total)
            scalar_entries(P, scalars, pts, t, [&](size_t set, u32 bucket, u32, u32) {
                atomicAdd(&lds_bins[(u32)(set - set0) * cb + (bucket >> fb)], 1u);
            });
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < nbins; k += 256) {
        const u32 v = lds_bins[k];
        if (v) atomicAdd(&bin_count[k], v);
        if (wg_hist) wg_hist[(size_t)blockIdx.x * nbins + k] = v;  // kept for k_part_offsets / the staged scatter
    }
}

// wg_off[wg][bin] = bin_start[bin] + sum of wg_hist[wg'][bin] over wg' < wg: where workgroup wg's run starts inside
// the bin (the staged scatter then needs neither a second count nor a returning global atomic per (workgroup, bin)).
// A workgroup takes 64 bins; lane (seg, bin) first sums, then rescans, its segment of the workgroups.
constexpr int OFF_SEGS = 16;
__global__ void __launch_bounds__(64 * OFF_SEGS) k_part_offsets(const u32* __restrict__ wg_hist, u32* __restrict__ wg_off,
                                                              const u32* __restrict__ bin_start, u32 nwg, u32 nbins) {
    __shared__ u32 seg_tot[OFF_SEGS][64];
    const u32 b = blockIdx.x * 64 + (threadIdx.x & 63), seg = threadIdx.x >> 6;
    const u32 per = (nwg + OFF_SEGS - 1) / OFF_SEGS, w0 = seg * per, w1 = w0 + per < nwg ? w0 + per : nwg;
    u32 tot = 0;
    if (b < nbins)
        for (u32 w = w0; w < w1; ++w) tot += wg_hist[(size_t)w * nbins + b];
    seg_tot[seg][threadIdx.x & 63] = tot;
    __syncthreads();
    if (b >= nbins) return;
    u32 run = bin_start[b];
    for (u32 s2 = 0; s2 < seg; ++s2) run += seg_tot[s2][threadIdx.x & 63];
    for (u32 w = w0; w < w1; ++w) {
        wg_off[(size_t)w * nbins + b] = run;
        run += wg_hist[(size_t)w * nbins + b];
    }
}

// bin_start[0..nbins] = exclusive scan of bin_count; bin_cursor zeroed
__global__ void __launch_bounds__(1024) k_part_scan(const u32* __restrict__ bin_count, u32* __restrict__ bin_start,
                                                    u32* __restrict__ bin_cursor, u32 nbins) {
    __shared__ u32 wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // nbins <= 4096: four consecutive bins per lane
    u32 v[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 idx = threadIdx.x * 4 + k;
        v[k] = idx < nbins ? bin_count[idx] : 0;
        if (idx < nbins) bin_cursor[idx] = 0;
        tot += v[k];
    }
    u32 x = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    u32 base = x - tot;
    for (int w2 = 0; w2 < wave; ++w2) base += wsum[w2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 idx = threadIdx.x * 4 + k;
        if (idx < nbins) bin_start[idx] = base;
        base += v[k];
        if (idx + 1 == nbins) bin_start[nbins] = base;
    }
}

__global__ void __launch_bounds__(256) k_part_scatter(DigitParams P, const u32* __restrict__ scalars,
                                                      const AffPt* __restrict__ pts, const u32* __restrict__ bin_start,
                                                      u32* __restrict__ bin_cursor, u32* __restrict__ tmp, u32 set0, u32 cb,
                                                      u32 nbins, int fb) {
    extern __shared__ u32 lds_bins[];
    u32* cnt = lds_bins;
    u32* base = lds_bins + nbins;
    for (u32 k = threadIdx.x; k < nbins; k += 256) cnt[k] = 0;
    __syncthreads();
    const size_t total = P.n * P.nbatch;
#pragma unroll 1
    for (int q = 0; q < PART_SCALARS; ++q) {
        const size_t t = ((size_t)blockIdx.x * PART_SCALARS + q) * 256 + threadIdx.x;
        if (t < total)
            scalar_entries(P, scalars, pts, t, [&](size_t set, u32 bucket, u32, u32) {
                atomicAdd(&cnt[(u32)(set - set0) * cb + (bucket >> fb)], 1u);
            });
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < nbins; k += 256) {
        const u32 v = cnt[k];
        base[k] = v ? bin_start[k] + atomicAdd(&bin_cursor[k], v) : 0u;
        cnt[k] = 0;
    }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < PART_SCALARS; ++q) {
        const size_t t = ((size_t)blockIdx.x * PART_SCALARS + q) * 256 + threadIdx.x;
        if (t < total)
            scalar_entries(P, scalars, pts, t, [&](size_t set, u32 bucket, u32 neg, u32 pidx) {
                const u32 bin = (u32)(set - set0) * cb + (bucket >> fb);
                const u32 r = atomicAdd(&cnt[bin], 1u);
                tmp[base[bin] + r] = ((bucket & ((1u << fb) - 1)) << (32 - fb)) | (neg << (31 - fb)) | pidx;
            });
    }
}

// k_part_scatter with the workgroup's entries staged in LDS in bin order and written out by consecutive lanes: a wave
// then writes ~8 runs of consecutive words per store instruction instead of 64 scattered 4-byte words.  1024 scalars
// per workgroup of 512 lanes, at most STAGE_CAP entries (16 per scalar): 64 KB of entries + 32 KB of bin ids + the
// three per-bin arrays.
constexpr u32 STAGE_CAP = 16384;
constexpr int STAGE_T = 512, STAGE_SCALARS = 2;
__global__ void __launch_bounds__(STAGE_T) k_part_scatter_staged(DigitParams P, const u32* __restrict__ scalars,
                                                                 const AffPt* __restrict__ pts,
                                                                 const u32* __restrict__ bin_start, u32* __restrict__ bin_cursor,
                                                                 u32* __restrict__ tmp, u32 set0, u32 cb, u32 nbins, int fb,
                                                                 const u32* __restrict__ wg_hist,
                                                                 const u32* __restrict__ wg_off) {
    extern __shared__ u32 lds_bins[];
    u32* cnt = lds_bins;               // per bin: count, then the running rank
    u32* gbase = lds_bins + nbins;     // per bin: start of this workgroup's run in tmp
    u32* lstart = lds_bins + 2 * nbins;  // per bin: start of the bin's entries in the staging area
    u32* stage = lds_bins + 3 * nbins;
    unsigned short* binid = (unsigned short*)(stage + STAGE_CAP);
    __shared__ u32 wsum[STAGE_T / 64];
    const size_t total = P.n * P.nbatch;
    if (wg_hist) {
        // k_part_count counted the same 1024 scalars: its histogram, and the run starts k_part_offsets derived from it
        for (u32 k = threadIdx.x; k < nbins; k += STAGE_T) cnt[k] = wg_hist[(size_t)blockIdx.x * nbins + k];
    } else {
        for (u32 k = threadIdx.x; k < nbins; k += STAGE_T) cnt[k] = 0;
        __syncthreads();
#pragma unroll 1
        for (int q = 0; q < STAGE_SCALARS; ++q) {
            const size_t t = ((size_t)blockIdx.x * STAGE_SCALARS + q) * STAGE_T + threadIdx.x;
            if (t < total)
                scalar_entries(P, scalars, pts, t, [&](size_t set, u32 bucket, u32, u32) {
                    atomicAdd(&cnt[(u32)(set - set0) * cb + (bucket >> fb)], 1u);
                });
        }
    }
    __syncthreads();
    {
        // exclusive scan of the counts (K consecutive bins per lane), and one returning atomic per non-empty bin
        const u32 K = (nbins + STAGE_T - 1) / STAGE_T;  // <= 8 for 4096 bins
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        u32 v[8], tot = 0;
        for (u32 k = 0; k < K; ++k) {
            const u32 idx = threadIdx.x * K + k;
            v[k] = idx < nbins ? cnt[idx] : 0;
            tot += v[k];
        }
        u32 x = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            u32 y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        u32 ex = x - tot;
        for (int w2 = 0; w2 < wave; ++w2) ex += wsum[w2];
        for (u32 k = 0; k < K; ++k) {
            const u32 idx = threadIdx.x * K + k;
            if (idx < nbins) {
                lstart[idx] = ex;
                if (wg_off) gbase[idx] = wg_off[(size_t)blockIdx.x * nbins + idx];
                else gbase[idx] = v[k] ? bin_start[idx] + atomicAdd(&bin_cursor[idx], v[k]) : 0u;
                cnt[idx] = 0;
                ex += v[k];
            }
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < STAGE_SCALARS; ++q) {
        const size_t t = ((size_t)blockIdx.x * STAGE_SCALARS + q) * STAGE_T + threadIdx.x;
        if (t < total)
            scalar_entries(P, scalars, pts, t, [&](size_t set, u32 bucket, u32 neg, u32 pidx) {
                const u32 bin = (u32)(set - set0) * cb + (bucket >> fb);
                const u32 slot = lstart[bin] + atomicAdd(&cnt[bin], 1u);
                stage[slot] = ((bucket & ((1u << fb) - 1)) << (32 - fb)) | (neg << (31 - fb)) | pidx;
                binid[slot] = (unsigned short)bin;
            });
    }
    __syncthreads();
    const u32 nent = lstart[nbins - 1] + cnt[nbins - 1];
    for (u32 e = threadIdx.x; e < nent; e += STAGE_T) {
        const u32 bin = binid[e];
        tmp[gbase[bin] + (e - lstart[bin])] = stage[e];
    }
}

// offsets / heavy / sorted are the group's (already shifted to its first set); heavy buckets are listed as in k_scan
__global__ void __launch_bounds__(256) k_bin_sort(const u32* __restrict__ tmp, const u32* __restrict__ bin_start,
                                                  u32* __restrict__ offsets, u32* __restrict__ sorted,
                                                  unsigned char* __restrict__ heavy, u32* __restrict__ heavy_list,
                                                  u32* __restrict__ nheavy, u32 heavy_cap, u32 cb, size_t nb, size_t set_cap,
                                                  int fb) {
    __shared__ u32 hist[FINE_MAX], pref[FINE_MAX], cur[FINE_MAX];
    __shared__ u32 wsum[4];
    const u32 F = 1u << fb;
    const int sh = 32 - fb;
    const u32 bin = blockIdx.x, set = bin / cb, coarse = bin % cb;
    const u32 beg = bin_start[bin], end = bin_start[bin + 1], set_start = bin_start[set * cb];
    for (u32 f = threadIdx.x; f < F; f += 256) {
        hist[f] = 0;
        cur[f] = 0;
    }
    __syncthreads();
    for (u32 e = beg + threadIdx.x; e < end; e += 256) atomicAdd(&hist[tmp[e] >> sh], 1u);
    __syncthreads();
    {
        // exclusive scan over the F fine buckets: K consecutive ones per thread, wave scan, the waves' totals
        const u32 K = F > 256 ? F >> 8 : 1;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        u32 v[4] = {0, 0, 0, 0}, tot = 0;
        if (threadIdx.x * K < F)
            for (u32 k = 0; k < K; ++k) {
                v[k] = hist[threadIdx.x * K + k];
                tot += v[k];
            }
        u32 x = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            u32 y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        u32 ex = x - tot;
        for (int w2 = 0; w2 < wave; ++w2) ex += wsum[w2];
        if (threadIdx.x * K < F)
            for (u32 k = 0; k < K; ++k) {
                pref[threadIdx.x * K + k] = ex;
                ex += v[k];
            }
    }
    __syncthreads();
    for (u32 f = threadIdx.x; f < F; f += 256) {
        const u32 ex = pref[f];
        const size_t bucket = (size_t)coarse * F + f;
        u32* off = offsets + (size_t)set * (nb + 1);
        off[bucket] = (beg - set_start) + ex;
        if (coarse + 1 == cb && f == F - 1) off[nb] = bin_start[(set + 1) * cb] - set_start;
        // bucket size = next prefix - own prefix
        const u32 nx = f + 1 < F ? pref[f + 1] : (end - beg);
        const u32 cntb = nx - ex;
        const bool hv = cntb > HEAVY;
        heavy[(size_t)set * nb + bucket] = hv ? 1 : 0;
        if (hv) {
            const u32 slot = atomicAdd(nheavy, 1u);
            if (slot < heavy_cap) {
                heavy_list[2 * slot] = set;
                heavy_list[2 * slot + 1] = (u32)bucket;
            }
        }
    }
    u32* dst = sorted + (size_t)set * set_cap + (beg - set_start);
    const u32 pmask = (1u << (sh - 1)) - 1u;
    for (u32 e = beg + threadIdx.x; e < end; e += 256) {
        const u32 w = tmp[e];
        const u32 f = w >> sh;
        const u32 r = atomicAdd(&cur[f], 1u);
        dst[pref[f] + r] = (w & pmask) | (((w >> (sh - 1)) & 1u) << 31);
    }
}

// exclusive scan of counts[set][0..nb) -> offsets[set][0..nb]; zeroes counts for the scatter pass.
// Buckets with more than HEAVY entries (> HEAVY/chunk pieces after k_accum) are flagged and listed
// so that k_heavy can combine their pieces with a whole wave instead of one lane.
__global__ void __launch_bounds__(1024) k_scan(u32* __restrict__ counts, u32* __restrict__ offsets, size_t nb,
                                               unsigned char* __restrict__ heavy, u32* __restrict__ heavy_list,
                                               u32* __restrict__ nheavy, u32 heavy_cap) {
    __shared__ u32 wsum[16];
    __shared__ u32 base_s;
    size_t set = blockIdx.x;
    u32* cnt = counts + set * nb;
    u32* off = offsets + set * (nb + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (size_t start = 0; start < nb; start += 1024) {
        size_t k = start + threadIdx.x;
        u32 v = k < nb ? cnt[k] : 0;
        if (k < nb) {
            cnt[k] = 0;
            const bool hv = v > HEAVY;
            heavy[set * nb + k] = hv ? 1 : 0;
            if (hv) {
                u32 slot = atomicAdd(nheavy, 1u);
                if (slot < heavy_cap) {
                    heavy_list[2 * slot] = (u32)set;
                    heavy_list[2 * slot + 1] = (u32)k;
                }
            }
        }
        u32 x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            u32 y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        u32 wbase = 0;
        for (int w2 = 0; w2 < wave; ++w2) wbase += wsum[w2];
        u32 base = base_s;
        if (k < nb) off[k] = base + wbase + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) base_s = base + wbase + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[nb] = base_s;
}

// Entries per accumulation lane, 2^lgc, chosen per set on the device: the host's choice `hi` assumes full-length
// uniform scalars; a set with fewer entries (short scalars leave the upper windows empty) takes smaller chunks, down to
// `lo`, until it has `target` lanes.  Every kernel that walks the pieces derives the same value from the set's total.
struct ChunkSel {
    int lo, hi;
    u32 target;
};
__device__ __forceinline__ int eff_lgc(u32 total, const ChunkSel& cs) {
    int l = cs.hi;
    while (l > cs.lo && (total >> l) < cs.target) --l;
    return l;
}

// Bucket accumulation, load-balanced: one lane per `chunk` consecutive entries of the sorted list
// (not per bucket), so skewed digit distributions (e.g. the < 2^248 elements of a "random blob",
// or a blob of equal elements) cost the same as uniform ones.  A lane walks its chunk, keeps the
// running sum of the current bucket and writes it out whenever the bucket changes and at the end
// of the chunk.  The piece of bucket b inside chunk t goes to slot b + t: along the sorted list
// both b and t are non-decreasing and consecutive pieces differ in at least one, so slots are
// unique, and bucket b's value is the sum of slots b + t for the chunks t its run touches.
__global__ void __launch_bounds__(256) k_accum(const u32* __restrict__ offsets, const u32* __restrict__ sorted,
                                               const AffPt* __restrict__ pts, Xyzz* __restrict__ partials, size_t nb,
                                               size_t nsets, size_t set_cap, size_t nchunk, ChunkSel cs) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nchunk * nsets) return;
    const size_t set = tid / nchunk, t = tid % nchunk;
    const u32* off = offsets + set * (nb + 1);
    const u32 total = off[nb];
    const u32 CHUNK = 1u << eff_lgc(total, cs);
    const u32 base = (u32)(t * CHUNK);
    if (base >= total) return;
    const u32 end = base + CHUNK < total ? base + CHUNK : total;
    const u32* run = sorted + set * set_cap;
    Xyzz* out = partials + set * (nb + nchunk) + t;
    // bucket containing position `base`: largest b with off[b] <= base
    u32 lo = 0, hi = (u32)nb;
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (off[mid] <= base) lo = mid;
        else hi = mid;
    }
    u32 b = lo, next_off = off[b + 1];
    Xyzz acc;
    g1::set_inf(acc);
    for (u32 k = base; k < end; ++k) {
        if (k >= next_off) {
            out[b] = acc;  // non-empty: position k-1 belonged to bucket b and to this chunk
            g1::set_inf(acc);
            do {
                ++b;
                next_off = off[b + 1];
            } while (k >= next_off);
        }
        u32 e = run[k];
        const AffPt* p = pts + (e & 0x7fffffffu);
        fp28::Fe x = p->x, y = p->y;
        if (e >> 31) y = fp28::neg<2>(y);
        g1::madd(acc, x, y);
    }
    out[b] = acc;
}

// Heavy buckets: their pieces are combined by whole waves instead of one lane.  Two passes so that a bucket
// holding a large share of all entries (short top window, equal scalars) is spread over many waves:
//   pass 1 (step 1, span HSEG): wave (bucket, segment) sums HSEG consecutive pieces into the segment's first slot
//   pass 2 (step HSEG, span all): one wave per bucket sums the segment sums into the bucket's first piece,
// which is the only one load_bucket reads for flagged buckets.
constexpr u32 HSEG = 1024;
__global__ void __launch_bounds__(64) k_heavy(Xyzz* __restrict__ partials, const u32* __restrict__ offsets,
                                              const u32* __restrict__ heavy_list, const u32* __restrict__ nheavy,
                                              u32 heavy_cap, size_t nb, size_t nchunk, u32 step, u32 span, ChunkSel cs,
                                              u32 set_lo, u32 set_hi) {
    __shared__ Xyzz sh[64];
    u32 cnt = *nheavy;
    if (cnt > heavy_cap) cnt = heavy_cap;
    const int lane = threadIdx.x;
    for (u32 idx = blockIdx.x; idx < cnt; idx += gridDim.x) {
        const size_t set = heavy_list[2 * idx], bk = heavy_list[2 * idx + 1];
        if (set < set_lo || set >= set_hi) continue;  // the sets of this launch's piece (workgroup-uniform)
        const u32* off = offsets + set * (nb + 1);
        Xyzz* pz = partials + set * (nb + nchunk) + bk;
        const int lgc = eff_lgc(off[nb], cs);
        const u32 t0 = off[bk] >> lgc, t1 = (off[bk + 1] - 1) >> lgc;
        const u32 nelem = (t1 - t0) / step + 1;  // elements t0 + k*step, k < nelem
        const u32 nseg = span ? (nelem + span - 1) / span : 1;
        for (u32 seg = blockIdx.y; seg < nseg; seg += gridDim.y) {
            const u32 k0 = span ? seg * span : 0;
            const u32 k1 = span && k0 + span < nelem ? k0 + span : nelem;
            Xyzz acc;
            g1::set_inf(acc);
            for (u32 k = k0 + lane; k < k1; k += 64) {
                Xyzz q = pz[t0 + k * step];
                g1::dadd(acc, q);
            }
            sh[lane] = acc;
            __syncthreads();
            for (int stride = 32; stride > 0; stride >>= 1) {
                if (lane < stride) {
                    Xyzz q = sh[lane + stride];
                    g1::dadd(acc, q);
                    sh[lane] = acc;
                }
                __syncthreads();
            }
            if (lane == 0) pz[t0 + k0 * step] = acc;
            __syncthreads();
        }
    }
}

// value of bucket bk of a set: sum of its pieces
__device__ __forceinline__ void load_bucket(Xyzz& v, const Xyzz* __restrict__ partials, const u32* __restrict__ off,
                                            const unsigned char* __restrict__ heavy, size_t bk, size_t nb,
                                            const ChunkSel& cs) {
    const int lgc = eff_lgc(off[nb], cs);
    const u32 beg = off[bk], end = off[bk + 1];
    g1::set_inf(v);
    if (end == beg) return;
    const u32 t0 = beg >> lgc, t1 = heavy[bk] ? t0 : (end - 1) >> lgc;
    v = partials[bk + t0];
    for (u32 t = t0 + 1; t <= t1; ++t) {
        Xyzz pz = partials[bk + t];
        g1::dadd(v, pz);
    }
}

// Bucket reduction  sum_k (k+1) * B_k  as a tree of (A, M) pairs:
//   A = plain sum of the subtree's buckets,  M = sum (k - first_k) * B_k over the subtree.
// A lane folds grp children (a power of two, GRP by default; level 0 takes fewer when the buckets alone would not
// fill the chip, since it also sums each bucket's pieces):  A = sum A_j,  M = sum M_j + S * sum_j j*A_j  with S = buckets per child
// (a power of two -> doublings).  Level 0 reads the buckets themselves (M_j = 0, S = 1).
// Every level is a short chain (<= ~3*GRP adds) over many lanes instead of one long running sum.
// Registers (hipcc 7.2, code-object metadata): three live XYZZ points and four inlined additions put k_level<true> at
// 388 VGPRs + 132 AGPRs and k_level<false> at 441 + 185 with FOUR spilled VGPRs (16 B of scratch per lane, touched once
// per folded child).  It stays: the kernel runs one wave per SIMD either way (launch bound 128, > 256 registers), it is
// the many-sets path only (batches of small variable-base MSMs, nsets > 64: throughput work of thousands of
// workgroups), and the spilled registers are four of ~630 (a store and a load of 16 B per lane next to ~13 000
// arithmetic instructions per folded child).  The remedy that took the spills out of k_tile_sums — one addition site in
// a loop whose operand comes from memory — has not been applied here.
template <bool FIRST>
__global__ void __launch_bounds__(128) k_level(const Xyzz* __restrict__ inA, const Xyzz* __restrict__ inM,
                                               Xyzz* __restrict__ outA, Xyzz* __restrict__ outM, size_t nin, size_t nsets,
                                               int logS, const u32* __restrict__ offsets,
                                               const unsigned char* __restrict__ heavy, size_t nb, size_t nchunk,
                                               int grp, ChunkSel cs) {
    const size_t nout = (nin + grp - 1) / grp;
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nout * nsets) return;
    const size_t set = tid / nout, g = tid % nout;
    const size_t lo = g * grp, hi = lo + grp < nin ? lo + grp : nin;
    Xyzz run, wsum, msum;
    g1::set_inf(run);
    g1::set_inf(wsum);
    g1::set_inf(msum);
    for (size_t k = hi; k-- > lo;) {
        Xyzz a;
        if (FIRST) {
            load_bucket(a, inA + set * (nb + nchunk), offsets + set * (nb + 1), heavy + set * nb, k, nb, cs);
        } else {
            a = inA[set * nin + k];
            Xyzz m = inM[set * nin + k];
            g1::dadd(msum, m);
        }
        g1::dadd(run, a);
        if (k > lo) g1::dadd(wsum, run);  // after the loop: wsum = sum_j j * A_j
    }
    if (!g1::is_inf(wsum))
        for (int d = 0; d < logS; ++d) g1::dbl(wsum);
    g1::dadd(msum, wsum);
    outA[set * nout + g] = run;
    outM[set * nout + g] = msum;
}

// ---- digit-decomposed bucket reduction (one or a few large MSMs) ----
// sum_k (k+1) B_k = T + sum_k k B_k,  T = sum_k B_k.  Write the bucket index in base 32, k = sum_j 32^j d_j(k):
//     sum_k k B_k = sum_j 32^j sum_d d * S[j][d],      S[j][d] = sum over { k : d_j(k) = d } of B_k
// and each digit value in binary:  sum_d d S[j][d] = sum_b 2^b sum over { d : bit b of d } of S[j][d].
// So the whole reduction is  (a) J * 32 PLAIN sums of nb/32 buckets (throughput work, one workgroup each, every
// bucket read J = ceil(log2(nb)/5) times),  (b) log2(nb) + 1 plain sums of <= 32 points (k_digit_bits),  (c) the
// Horner over the log2(nb) bit sums that k_winsum_wide already does.  Depth: 4 + 8 tree steps, 5 tree steps, the
// Horner — against ~24-addition chains per level of the (A, M) tree, which this replaces when there are few sets.
constexpr int DIGIT_BITS = 5, DIGIT_T = 128;
// every bucket's pieces folded once into a dense array (each bucket is then read once per digit)
__global__ void __launch_bounds__(128) k_fold_buckets(const Xyzz* __restrict__ partials, const u32* __restrict__ offsets,
                                                      const unsigned char* __restrict__ heavy, Xyzz* __restrict__ dense,
                                                      size_t nb, size_t nsets, size_t nchunk, ChunkSel cs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * nsets) return;
    const size_t set = t / nb, k = t % nb;
    Xyzz v;
    load_bucket(v, partials + set * (nb + nchunk), offsets + set * (nb + 1), heavy + set * nb, k, nb, cs);
    dense[t] = v;
}

__global__ void __launch_bounds__(DIGIT_T) k_digit_sums(const Xyzz* __restrict__ dense, Xyzz* __restrict__ S, size_t nb,
                                                        int logNb, int J) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ds[];
    Xyzz* sh = (Xyzz*)smem_ds;
    const int per_set = J << DIGIT_BITS;
    const size_t set = blockIdx.x / per_set;
    const int jd = (int)(blockIdx.x % per_set), j = jd >> DIGIT_BITS, d = jd & ((1 << DIGIT_BITS) - 1);
    const int lo_bits = DIGIT_BITS * j;
    const int w = logNb - lo_bits < DIGIT_BITS ? logNb - lo_bits : DIGIT_BITS;  // width of digit j
    const int lane = threadIdx.x;
    Xyzz acc;
    g1::set_inf(acc);
    if (d < (1 << w)) {
        const size_t cnt = nb >> w;  // buckets whose digit j equals d
        const Xyzz* bk = dense + set * nb;
        for (size_t m = lane; m < cnt; m += DIGIT_T) {
            const size_t k = ((m >> lo_bits) << (lo_bits + w)) | ((size_t)d << lo_bits) | (m & (((size_t)1 << lo_bits) - 1));
            Xyzz v = bk[k];
            g1::dadd(acc, v);
        }
    }
    sh[lane] = acc;
    __syncthreads();
    for (int stride = DIGIT_T / 2; stride > 0; stride >>= 1) {
        if (lane < stride) {
            Xyzz v = sh[lane + stride];
            g1::dadd(acc, v);
            sh[lane] = acc;
        }
        __syncthreads();
    }
    if (lane == 0) S[blockIdx.x] = acc;
}

// Tiled form of the digit sums (nb a multiple of 1024).  A workgroup takes 1024 consecutive buckets as a 32 x 32
// matrix M[g][d] (bucket = 32 g + d): the row sums are the GROUP sums G[g] — all the higher digits need, since digits
// 1.. of a bucket are digits of its group — and the column sums are this tile's share of S[0][d].  Two additions per
// bucket instead of J = 3, every bucket's pieces folded exactly once in the same kernel, and nsets * nb / 1024 =
// 256 workgroups of eight waves at n = 2^20 (a chain of ~15 additions).  The (digit, value) cells that are left have
// nb / 1024 values each: few enough additions for one wave per addition (k_digit_sums_wide, k_digit_bits_wide).
constexpr int TILE_T = 512;  // ROWS = 32: two buckets per lane and phase, a full CU (two waves per SIMD) per tile
// Every addition is at one of TWO inlined sites, each the body of a loop whose second operand comes from
// memory (the form that took the scratch out of the G1 stages, profiles/NOTES.md §16): loop 1 folds the pieces of the lane's two
// buckets, loop 2 is a schedule of steps — the pair of buckets, four row-tree levels, the pair of rows, the column-tree
// levels — in which a step only chooses where the operand comes from and where the sum goes.  (Round 3's form had five
// sites, each with its own never-taken doubling: 256 VGPRs, 86 of them spilled; removed in round 5.)
//
// ROWS = 32 (tiles of 1024 buckets, 512 lanes: the default) or 16 (tiles of 512 buckets, 256 lanes; tuning key tile_rows).
// The kernel needs 240 VGPRs, so a 512-lane workgroup wants ALL registers of all four SIMDs of a CU; a 256-lane one is a
// wave per SIMD and could take the place of one retiring workgroup of a running accumulation (230 VGPRs, four waves).
// Round 6 built the 16-row form for that — the reduction of one MSM under the accumulation of the next — and measured
// that it does not happen: beside an accumulation the tiles still wait (2.9 ms instead of 0.32) whatever the stream
// priority, and alone the 16-row form is slower (0.936 vs 0.889 ms at 2^16, 3.52 vs 3.42 at 2^20: twice the tiles, twice
// the values per S[0][d] cell).  profiles/NOTES.md §20.
// QUAD (the default, tuning key tile_quad): the tree levels with at most a quarter of the lanes busy run four lanes per
// addition — 237 VGPRs, no scratch in either form (hipcc 7.2, code-object metadata).
template <int ROWS, bool QUAD>
__global__ void __launch_bounds__(ROWS * 16) k_tile_sums_loop(const Xyzz* __restrict__ partials, const u32* __restrict__ offsets,
                                                              const unsigned char* __restrict__ heavy, Xyzz* __restrict__ dense,
                                                              Xyzz* __restrict__ Gs, Xyzz* __restrict__ Cp, size_t nb,
                                                              size_t nchunk, ChunkSel cs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ts[];
    Xyzz* sh = (Xyzz*)smem_ts;
    constexpr int T = ROWS * 16;
    constexpr int CL = ROWS == 32 ? 4 : 3;  // column-tree levels: ROWS / 2 row pairs -> 1
    const size_t ntiles = nb / (size_t)(ROWS * 32);
    const size_t set = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
    const size_t k0 = tile * (size_t)(ROWS * 32);
    const int t = threadIdx.x;
    const int r = t >> 4, q = t & 15;   // rows: buckets 32 r + 2 q, + 1
    const int c = t & 31, rg = t >> 5;  // columns: rows 2 rg, 2 rg + 1 of column c
    const size_t kb = k0 + (size_t)(32 * r + 2 * q);
    Xyzz* dn = dense + set * nb;
    Xyzz acc;
    g1::set_inf(acc);
    {
        const Xyzz* pz = partials + set * (nb + nchunk);
        const u32* off = offsets + set * (nb + 1);
        const unsigned char* hv = heavy + set * nb;
        const int lgc = eff_lgc(off[nb], cs);
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
            const size_t bk = kb + i;
            const u32 beg = off[bk], end = off[bk + 1];
            g1::set_inf(acc);
            if (end != beg) {
                const u32 t0 = beg >> lgc, t1 = hv[bk] ? t0 : (end - 1) >> lgc;
                acc = pz[bk + t0];
#pragma unroll 1
                for (u32 tt = t0 + 1; tt <= t1; ++tt) {
                    Xyzz v = pz[bk + tt];
                    if (g1::dadd_unequal(acc, v)) g1::dbl(acc);
                }
            }
            dn[bk] = acc;
        }
    }
    // acc = bucket kb + 1; step 0 adds bucket kb (this lane wrote it)
    if constexpr (QUAD) {
        // Two halves of the same shape — rows, then columns: the pair step and tree level 0 with one lane per addition
        // (every lane, then half of them, busy), the remaining levels (at most T / 4 additions) with four lanes per
        // addition (g1grp.hip.h: an addition 4 multiplications deep instead of 14).  Two loops, two inlined sites: as a
        // third branch of ONE step loop the four-lane site took the kernel to 256 VGPRs with 184 spilled.
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int unit = half ? 32 : ROWS;     // slots between the operands of a tree step, per stride
            const int st0 = half ? ROWS / 4 : 8;   // stride of tree level 0
            const int nlev = half ? CL : 4;
#pragma unroll 1
            for (int j = 0; j < 2; ++j) {
                bool active = true;
                Xyzz v;
                if (j == 0) {
                    v = half ? dn[k0 + (size_t)(32 * (2 * rg + 1) + c)] : dn[kb];
                } else {
                    active = t < unit * st0;
                    if (active) v = sh[t + unit * st0];
                }
                if (active && g1::dadd_unequal(acc, v)) g1::dbl(acc);
                if (j == 0 && half == 0) {
                    // the row tree runs q-major (slot ROWS q + r): the lanes still adding at a level are the first
                    // ROWS * stride of the workgroup, whole waves drop out level by level
                    sh[ROWS * q + r] = acc;
                    __syncthreads();
                    acc = sh[t];  // slot t is only ever written by lane t from here on
                } else {
                    if (active) sh[t] = acc;
                    __syncthreads();
                }
            }
#pragma unroll 1
            for (int lvl = 1; lvl < nlev; ++lvl) {
                const int n = unit * (st0 >> lvl);  // additions of this level: slots [0, n) += slots [n, 2 n)
                const int g = t >> 2, role = t & 3;
                if (g < n) {
                    Xyzz a = sh[g];
                    const Xyzz b = sh[g + n];
                    if (grp::dadd_body<4>(a, b, role)) grp::dbl_body<4>(a, role);
                    if (role == 0) sh[g] = a;
                }
                __syncthreads();
            }
            acc = sh[t];  // lanes below ROWS / 32 hold a row sum / a column sum of the tile
            if (half == 0) {
                if (t < ROWS) Gs[set * (nb >> 5) + (k0 >> 5) + t] = acc;
                __threadfence();  // the folded buckets this workgroup wrote are read back by other lanes
                __syncthreads();
                acc = dn[k0 + (size_t)(32 * (2 * rg) + c)];
            }
        }
    } else {
#pragma unroll 1
    for (int s = 0; s < 6 + CL; ++s) {
        const int lvl = s < 5 ? s - 1 : s - 6;                        // tree steps 1..4 / 6..: strides 8, 4, 2, 1 / ROWS/4 .. 1
        const int stride = lvl >= 0 ? (s < 5 ? 8 >> lvl : (ROWS / 4) >> lvl) : 0;
        const int unit = s < 5 ? ROWS : 32;                           // slots between the operands of a tree step, per stride
        bool active = true;
        Xyzz v;
        if (s == 0) {
            v = dn[kb];
        } else if (s == 5) {
            v = dn[k0 + (size_t)(32 * (2 * rg + 1) + c)];
        } else {
            active = t < unit * stride;
            if (active) v = sh[t + unit * stride];
        }
        if (active && g1::dadd_unequal(acc, v)) g1::dbl(acc);
        if (s == 0) {
            // the row tree runs q-major (slot ROWS q + r): the lanes still adding at a level are the first ROWS * stride of
            // the workgroup, whole waves drop out level by level
            sh[ROWS * q + r] = acc;
            __syncthreads();
            acc = sh[t];  // slot t is only ever written by lane t from here on
        } else {
            if (active) sh[t] = acc;
            __syncthreads();
        }
        if (s == 4) {
            if (t < ROWS) Gs[set * (nb >> 5) + (k0 >> 5) + t] = acc;
            __threadfence();  // the folded buckets this workgroup wrote are read back by other lanes
            __syncthreads();
            acc = dn[k0 + (size_t)(32 * (2 * rg) + c)];
        }
    }
    }
    if (rg == 0) Cp[(set * ntiles + tile) * 32 + c] = acc;
}

// S[set][0][d] = sum over the tiles of Cp[set][tile][d];  S[set][j][d], j >= 1, = sum of the groups whose digit j - 1
// (base 32, of the group index) equals d.  One 64-thread workgroup per cell, nb / 1024 values each (32 at 2^15 buckets).
__global__ void __launch_bounds__(64) k_digit_sums2(const Xyzz* __restrict__ Gs, const Xyzz* __restrict__ Cp,
                                                    Xyzz* __restrict__ S, size_t nb, int logNb, int J, size_t ntiles) {
    __shared__ Xyzz sh[64];
    const int per_set = J << DIGIT_BITS;
    const size_t set = blockIdx.x / per_set;
    const int jd = (int)(blockIdx.x % per_set), j = jd >> DIGIT_BITS, d = jd & ((1 << DIGIT_BITS) - 1);
    const size_t ng = nb >> 5;
    const int lane = threadIdx.x;
    Xyzz acc;
    g1::set_inf(acc);
    if (j == 0) {
        for (size_t e = lane; e < ntiles; e += 64) {
            Xyzz v = Cp[(set * ntiles + e) * 32 + d];
            g1::dadd(acc, v);
        }
    } else {
        const int logNg = logNb - DIGIT_BITS, lo_bits = DIGIT_BITS * (j - 1);
        const int w = logNg - lo_bits < DIGIT_BITS ? logNg - lo_bits : DIGIT_BITS;  // width of this digit
        if (d < (1 << w)) {
            const size_t cnt = ng >> w;
            for (size_t m = lane; m < cnt; m += 64) {
                const size_t g = ((m >> lo_bits) << (lo_bits + w)) | ((size_t)d << lo_bits) | (m & (((size_t)1 << lo_bits) - 1));
                Xyzz v = Gs[set * ng + g];
                g1::dadd(acc, v);
            }
        }
    }
    sh[lane] = acc;
    __syncthreads();
    for (int stride = 32; stride > 0; stride >>= 1) {
        if (lane < stride) {
            Xyzz v = sh[lane + stride];
            g1::dadd(acc, v);
            sh[lane] = acc;
        }
        __syncthreads();
    }
    if (lane == 0) S[blockIdx.x] = acc;
}

// top[set][q] = sum over { d : bit (q mod 5) of d } of S[set][q / 5][d]  for q < logNb;  top[set][logNb] = T = sum_d
// S[set][0][d];  top[set][logNb + 1] = infinity  — the layout k_winsum(_wide) expects (R_q, A, M) with logS = 0
__global__ void __launch_bounds__(64) k_digit_bits(const Xyzz* __restrict__ S, Xyzz* __restrict__ top, int logNb, int J) {
    __shared__ Xyzz sh[32];
    const int per_top = logNb + 2;
    const size_t set = blockIdx.x / per_top;
    const int q = (int)(blockIdx.x % per_top);
    const int lane = threadIdx.x;
    Xyzz acc;
    g1::set_inf(acc);
    if (lane < 32 && q <= logNb) {
        const int j = q < logNb ? q / DIGIT_BITS : 0, b = q % DIGIT_BITS;
        if (q == logNb || ((lane >> b) & 1)) acc = S[(set * J + j) * 32 + lane];
    }
    if (lane < 32) sh[lane] = acc;
    __syncthreads();
    for (int stride = 16; stride > 0; stride >>= 1) {
        if (lane < stride) {
            Xyzz v = sh[lane + stride];
            g1::dadd(acc, v);
            sh[lane] = acc;
        }
        __syncthreads();
    }
    if (lane == 0) top[blockIdx.x] = acc;
}

// Top of the reduction tree.  Once a set is down to nin <= TOP_MAX nodes (A_j, M_j) of S buckets each, further
// GRP-ary levels are pure latency (a handful of waves, ~24 dependent additions plus log2(S) doublings per level).
// Instead:   sum_k (k+1) B_k = sum_j A_j + sum_j M_j + S * sum_q 2^q R_q ,   R_q = sum over { j : bit q of j } of A_j
// — B + 2 plain sums per set (B = ceil(log2 nin)), each a strided accumulate plus an LDS tree in its own workgroup,
// all concurrent; k_winsum then runs the short Horner over q.
constexpr int TOPT = 256;
constexpr size_t TOP_MAX = 2048;
__global__ void __launch_bounds__(TOPT) k_top(const Xyzz* __restrict__ inA, const Xyzz* __restrict__ inM,
                                              Xyzz* __restrict__ top, size_t nin, int B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_top[];
    Xyzz* sh = (Xyzz*)smem_top;
    const size_t set = blockIdx.x / (B + 2);
    const int q = (int)(blockIdx.x % (B + 2));
    const Xyzz* src = (q == B + 1 ? inM : inA) + set * nin;
    const int lane = threadIdx.x;
    Xyzz acc;
    g1::set_inf(acc);
    for (size_t j = lane; j < nin; j += TOPT) {
        if (q >= B || ((j >> q) & 1)) {
            Xyzz v = src[j];
            g1::dadd(acc, v);
        }
    }
    sh[lane] = acc;
    __syncthreads();
    for (int stride = TOPT / 2; stride > 0; stride >>= 1) {
        if (lane < stride) {
            Xyzz v = sh[lane + stride];
            g1::dadd(acc, v);
            sh[lane] = acc;
        }
        __syncthreads();
    }
    if (lane == 0) top[blockIdx.x] = acc;
}

// one lane per set: window sum = A + M + 2^logS * sum_q 2^q R_q
__global__ void __launch_bounds__(64) k_winsum(const Xyzz* __restrict__ top, Xyzz* __restrict__ win, size_t nsets, int B,
                                               int logS) {
    size_t set = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (set >= nsets) return;
    const Xyzz* t = top + set * (B + 2);
    Xyzz acc;
    g1::set_inf(acc);
    for (int q = B - 1; q >= 0; --q) {
        if (!g1::is_inf(acc)) g1::dbl(acc);
        Xyzz r = t[q];
        g1::dadd(acc, r);
    }
    g1::dbl_k(acc, logS);
    Xyzz r = t[B + 1];
    g1::dadd(acc, r);
    r = t[B];
    g1::dadd(acc, r);
    win[set] = acc;
}

// one lane per MSM: Horner over windows (unprepared) and conversion to blst Jacobian
__global__ void __launch_bounds__(64) k_final(const Xyzz* __restrict__ rootA, const Xyzz* __restrict__ rootM,
                                              void* __restrict__ out_v, size_t nbatch, int nwin, int c, int prepared,
                                              int out_mode) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (out_mode == kzgamd::OUT_WINDOWS) {
        // b indexes (MSM, window): window sum = M + A, handed to the host as a Jacobian point
        if (b >= nbatch * (size_t)nwin) return;
        Xyzz acc = rootM[b];
        if (rootA) {
            Xyzz a = rootA[b];
            g1::dadd(acc, a);
        }
        ff::Fp* out = (ff::Fp*)out_v;
        ff::Fp j[3];
        g1::to_blst_jacobian(j, acc);
        out[3 * b] = j[0];
        out[3 * b + 1] = j[1];
        out[3 * b + 2] = j[2];
        return;
    }
    const bool live = b < nbatch;
    if (!live) b = nbatch - 1;  // idle lanes shadow the last MSM so that the whole block reaches the barriers
    Xyzz acc;
    if (prepared) {
        acc = rootM[b];  // sum (k+1) B_k = M + A   (wide-table path: rootM is the sum, rootA absent)
        if (rootA) {
            Xyzz a = rootA[b];
            g1::dadd(acc, a);
        }
    } else {
        g1::set_inf(acc);
        for (int w = nwin - 1; w >= 0; --w) {
            g1::dbl_k(acc, c);  // the curve has odd order: doubling never reaches infinity
            Xyzz r = rootM[b * nwin + w];
            g1::dadd(acc, r);
            if (rootA) {
                r = rootA[b * nwin + w];
                g1::dadd(acc, r);
            }
        }
    }
    if (out_mode == kzgamd::OUT_COMPRESSED) {
        // affine conversion needs 1/(ZZ*ZZZ) per point: one inversion per 64-lane block (Montgomery's trick
        // over an LDS product scan) instead of one per lane
        __shared__ fp28::Fe sh_pre[64], sh_suf[64];
        __shared__ fp28::Fe sh_inv;
        const int t = threadIdx.x;
        fp28::Fe z = (!live || g1::is_inf(acc)) ? fp28::one() : fp28::mul(acc.zz, acc.zzz);
        sh_pre[t] = z;
        sh_suf[t] = z;
        __syncthreads();
        for (int off = 1; off < 64; off <<= 1) {
            fp28::Fe a = sh_pre[t], c2 = sh_suf[t];
            if (t >= off) a = fp28::mul(sh_pre[t - off], a);
            if (t + off < 64) c2 = fp28::mul(c2, sh_suf[t + off]);
            __syncthreads();
            sh_pre[t] = a;
            sh_suf[t] = c2;
            __syncthreads();
        }
        if (t == 0) sh_inv = g1io::inverse(sh_pre[63]);
        __syncthreads();
        fp28::Fe zi = sh_inv;
        if (t > 0) zi = fp28::mul(zi, sh_pre[t - 1]);
        if (t < 63) zi = fp28::mul(zi, sh_suf[t + 1]);
        if (!live) return;
        unsigned char buf[48];
        g1io::compress_with_inverse(buf, acc, zi);
        u32* o = (u32*)out_v + 12 * b;
#pragma unroll
        for (int k = 0; k < 12; ++k)
            o[k] = (u32)buf[4 * k] | ((u32)buf[4 * k + 1] << 8) | ((u32)buf[4 * k + 2] << 16) | ((u32)buf[4 * k + 3] << 24);
        return;
    }
    if (!live) return;
    ff::Fp* out = (ff::Fp*)out_v;
    ff::Fp j[3];
    g1::to_blst_jacobian(j, acc);
    out[3 * b] = j[0];
    out[3 * b + 1] = j[1];
    out[3 * b + 2] = j[2];
}

// The serial tails with limb-parallel arithmetic (fpw.hip.h / g1w.hip.h): one point operation per wave, the limbs of a
// coordinate across the 16 lanes of a row and the independent products of a formula across the four rows — a doubling
// is 3 dependent multiplications deep (7 in the single-lane code), an addition 4 (14).  Used when there are only a few
// chains (one large MSM).
//
// k_winsum_wide: one wave per set,  window sum = A + M + 2^logS * sum_q 2^q R_q
__global__ void __launch_bounds__(64) k_winsum_wide(const Xyzz* __restrict__ top, Xyzz* __restrict__ win, int B, int logS) {
    __shared__ u32 sh[16];
    const int lane = threadIdx.x;
    const fpw::Lane lc = fpw::lane_consts(lane);
    const Xyzz* t = top + (size_t)blockIdx.x * (B + 2);
    g1w::WPt acc;
    g1w::set_inf(acc);
    for (int q = B - 1; q >= 0; --q) {
        if (!g1w::is_inf(acc)) g1w::dbl(acc, lc, lane);
        g1w::dadd(acc, g1w::load(t + q, lane), lc, sh, lane);
    }
    g1w::dbl_k(acc, logS, lc, lane);
    g1w::dadd(acc, g1w::load(t + B + 1, lane), lc, sh, lane);
    g1w::dadd(acc, g1w::load(t + B, lane), lc, sh, lane);
    g1w::store(win + blockIdx.x, acc, lc, lane);
}

// k_final_wide: Horner over the window sums of one variable-base MSM per wave
__global__ void __launch_bounds__(64) k_final_wide(const Xyzz* __restrict__ win, void* __restrict__ out_v, int nwin, int c) {
    __shared__ u32 sh[16];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt wacc = g1w::load(win + b * nwin + (nwin - 1), lane);
    for (int w = nwin - 2; w >= 0; --w) {
        g1w::dbl_k(wacc, c, lc, lane);  // the curve has odd order: doubling never reaches infinity
        g1w::dadd(wacc, g1w::load(win + b * nwin + w, lane), lc, sh, lane);
    }
    Xyzz acc = g1w::to_single(wacc, lc, sh, lane);
    ff::Fp j[3];
    g1::to_blst_jacobian(j, acc);
    if (lane == 0) {
        ff::Fp* out = (ff::Fp*)out_v;
        out[3 * b] = j[0];
        out[3 * b + 1] = j[1];
        out[3 * b + 2] = j[2];
    }
}


// Small plain sums with one limb-parallel addition per wave.  A single-lane XYZZ addition is a ~27 us dependency
// chain (6 700 instructions at one issue per ~8.5 cycles for a lone wave), the row-parallel one ~2.7 us, but it
// occupies a whole wave: worth it where a stage has at most a few thousand additions per level.  A cell (one output
// sum) is cut into WSPLIT parts; wave (cell, part) adds its elements one after the other, stores its partial sum, and
// the last wave of a cell to finish (a counter per cell) adds the partial sums.
constexpr int WSPLIT = 4;   // parts per cell of the bit sums (<= 32 values per cell)
constexpr int WSPLIT_S = 2; // parts per cell of the digit sums: fewer, longer chains — the limb-parallel code is a
                            // dependent instruction stream, and more than ~2 such waves per SIMD only queue up
template <int NSPLIT>
__device__ __forceinline__ void wide_cell_finish(g1w::WPt& acc, Xyzz* __restrict__ part, Xyzz* __restrict__ out,
                                                 u32* __restrict__ counter, size_t cell, int sub, const fpw::Lane& lc, u32* sh,
                                                 u32* last_s, int lane) {
    g1w::store(part + cell * NSPLIT + sub, acc, lc, lane);
    __threadfence();
    if (lane == 0) *last_s = atomicAdd(counter + cell, 1u);
    __syncthreads();
    if (*last_s != (u32)(NSPLIT - 1)) return;
    __threadfence();
    g1w::WPt tot;
    g1w::set_inf(tot);
    g1w::add_n(tot, part + cell * NSPLIT, 1, NSPLIT, lc, sh, lane);
    g1w::store(out + cell, tot, lc, lane);
    if (lane == 0) counter[cell] = 0;  // ready for the next launch
}

// k_digit_sums2 with limb-parallel additions: grid = cells * WSPLIT waves
__global__ void __launch_bounds__(64) k_digit_sums_wide(const Xyzz* __restrict__ Gs, const Xyzz* __restrict__ Cp,
                                                        Xyzz* __restrict__ S, Xyzz* __restrict__ part,
                                                        u32* __restrict__ counter, size_t nb, int logNb, int J, size_t ntiles) {
    __shared__ u32 sh[16];
    __shared__ u32 last_s;
    const int lane = threadIdx.x;
    const size_t cell = blockIdx.x / WSPLIT_S;
    const int sub = (int)(blockIdx.x % WSPLIT_S);
    const int per_set = J << DIGIT_BITS;
    const size_t set = cell / per_set;
    const int jd = (int)(cell % per_set), j = jd >> DIGIT_BITS, d = jd & ((1 << DIGIT_BITS) - 1);
    const size_t ng = nb >> 5;
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt acc;
    g1w::set_inf(acc);
    // the operand of the NEXT addition is loaded before the current one starts: the values come from other CUs' tiles
    // (another XCD's L2 or HBM), and a load issued when its addition is due leaves the chain waiting for it every time
    if (j == 0) {
        const Xyzz* src = Cp + set * ntiles * 32 + d;
        g1w::WPt nx;
        g1w::set_inf(nx);
        if ((size_t)sub < ntiles) nx = g1w::load(src + (size_t)sub * 32, lane);
        for (size_t e = sub; e < ntiles; e += WSPLIT_S) {
            const g1w::WPt cur = nx;
            if (e + WSPLIT_S < ntiles) nx = g1w::load(src + (e + WSPLIT_S) * 32, lane);
            g1w::dadd(acc, cur, lc, sh, lane);
        }
    } else {
        const int logNg = logNb - DIGIT_BITS, lo_bits = DIGIT_BITS * (j - 1);
        const int w = logNg - lo_bits < DIGIT_BITS ? logNg - lo_bits : DIGIT_BITS;  // width of this digit
        if (d < (1 << w)) {
            const size_t cnt = ng >> w;
            auto group_of = [&](size_t m) {
                return ((m >> lo_bits) << (lo_bits + w)) | ((size_t)d << lo_bits) | (m & (((size_t)1 << lo_bits) - 1));
            };
            const Xyzz* src = Gs + set * ng;
            g1w::WPt nx;
            g1w::set_inf(nx);
            if ((size_t)sub < cnt) nx = g1w::load(src + group_of(sub), lane);
            for (size_t m = sub; m < cnt; m += WSPLIT_S) {
                const g1w::WPt cur = nx;
                if (m + WSPLIT_S < cnt) nx = g1w::load(src + group_of(m + WSPLIT_S), lane);
                g1w::dadd(acc, cur, lc, sh, lane);
            }
        }
    }
    wide_cell_finish<WSPLIT_S>(acc, part, S, counter, cell, sub, lc, sh, &last_s, lane);
}

// k_digit_bits with limb-parallel additions: part `sub` of cell (set, q) takes the digit values 8 sub .. 8 sub + 7
__global__ void __launch_bounds__(64) k_digit_bits_wide(const Xyzz* __restrict__ S, Xyzz* __restrict__ top,
                                                        Xyzz* __restrict__ part, u32* __restrict__ counter, int logNb, int J) {
    __shared__ u32 sh[16];
    __shared__ u32 last_s;
    const int lane = threadIdx.x;
    const size_t cell = blockIdx.x / WSPLIT;
    const int sub = (int)(blockIdx.x % WSPLIT);
    const int per_top = logNb + 2;
    const size_t set = cell / per_top;
    const int q = (int)(cell % per_top);
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt acc;
    g1w::set_inf(acc);
    if (q <= logNb) {
        const int j = q < logNb ? q / DIGIT_BITS : 0, b = q % DIGIT_BITS;
        // every digit sum of the part is loaded an addition ahead of its use, taken or not (they were written by other CUs
        // a kernel ago)
        constexpr int PER = 32 / WSPLIT;
        const Xyzz* src = S + (set * J + j) * 32 + sub * PER;
        g1w::WPt nx = g1w::load(src, lane);
#pragma unroll 1
        for (int k = 0; k < PER; ++k) {
            const g1w::WPt cur = nx;
            if (k + 1 < PER) nx = g1w::load(src + k + 1, lane);
            if (q == logNb || (((sub * PER + k) >> b) & 1)) g1w::dadd(acc, cur, lc, sh, lane);
        }
    }
    wide_cell_finish<WSPLIT>(acc, part, top, counter, cell, sub, lc, sh, &last_s, lane);
}

// out[c] = sum of the 8 * per_wave consecutive points of cell c: eight waves per cell add per_wave points each, the last
// one to finish adds the eight partial sums (the fold of a single commitment's partial sums: two launches instead of
// the 13 single-lane tree levels of k_blocksum)
constexpr int WFOLD = 8;
constexpr size_t WIDE_FOLD_MAX = 4;  // MSMs per launch folded limb-parallel (k_wide_tree).  Round 6, host-buffer commitment calls of
                                     // 1 / 2 / 3 / 4 / 8 blobs, ms: 1 -> 0.185 / 0.227 / 0.269 / 0.277 / 0.309, 4 -> 0.182 / 0.186 / 0.230 /
                                     // 0.244 / 0.301, 8 -> . / 0.193 / 0.230 / 0.252 / 0.381 (k_blocksum_hybrid beyond 4)
__global__ void __launch_bounds__(64) k_wide_fold64(const Xyzz* __restrict__ in, Xyzz* __restrict__ out, Xyzz* __restrict__ part,
                                                    u32* __restrict__ counter, int per_wave) {
    __shared__ u32 sh[16];
    __shared__ u32 last_s;
    const int lane = threadIdx.x;
    const size_t cell = blockIdx.x / WFOLD;
    const int sub = (int)(blockIdx.x % WFOLD);
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt acc;
    g1w::set_inf(acc);
    const Xyzz* src = in + (cell * WFOLD + (size_t)sub) * per_wave;  // a cell = WFOLD * per_wave consecutive points
    g1w::add_n(acc, src, 1, per_wave, lc, sh, lane);
    wide_cell_finish<WFOLD>(acc, part, out, counter, cell, sub, lc, sh, &last_s, lane);
}

// The fold of a few commitments' partial sums in one launch of limb-parallel additions (round 6).  A limb-parallel addition
// is ~2.2 us for a wave that has its SIMD to itself, so the shape is: never more than one wave per SIMD while the chip
// has room, and as few device-wide hand-overs as the depth allows (a hand-over — store, fence, atomic, fence, reload
// across the XCDs' L2s — costs ~10 us, as much as four additions).
//   stage 0: a workgroup of FOUR waves (one per SIMD of its CU) takes 32 consecutive partial sums: every wave adds 8,
//            wave 0 adds the four results through LDS                                       (7 + 3 additions)
//   stage 1: the LAST of 16 neighbouring workgroups to finish (a counter per group) adds their 16 sums the same way:
//            four per wave, then the four                                                    (3 + 3)
//   stage 2: the last group of an MSM to finish (a counter per MSM) adds the group sums     (3 + 3 for 8192 partial sums)
// 22 additions and two hand-overs deep for 8192 partial sums.  Measured under the profiler (one commitment): 0.085 ms for a
// first form with 16-wave workgroups (four waves per SIMD on a quarter of the chip) and one hand-over, 0.098 ms for a form
// with a hand-over per radix-4 level, 0.116 ms for the two launches of k_wide_fold64 with their memsets (round 5).
// Counters are zero between launches (the wave that takes a group or an MSM resets its counter).
constexpr int WTREE_PER_WG = 32, WTREE_GROUP = 16;
__global__ void __launch_bounds__(256) k_wide_tree(const Xyzz* __restrict__ in, Xyzz* __restrict__ out, Xyzz* __restrict__ part,
                                                   u32* __restrict__ counter, size_t n, u32 nwg) {
    __shared__ Xyzz shp[4];
    __shared__ u32 scr[4][16];
    __shared__ u32 seen_s;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t set = blockIdx.x / nwg;
    const u32 wg = blockIdx.x % nwg;
    const u32 ngroups = nwg / WTREE_GROUP;
    const fpw::Lane lc = fpw::lane_consts(lane);
    u32* sh = scr[wave];
    // per MSM: nwg workgroup sums, then ngroups group sums; counters: ngroups, then one
    Xyzz* p1 = part + set * (size_t)(nwg + ngroups);
    Xyzz* p2 = p1 + nwg;
    u32* cnt = counter + set * (size_t)(ngroups + 1);
    g1w::WPt acc;
    // acc of the four waves -> wave 0's acc (workgroup-uniform control flow)
    auto fold4 = [&]() {
        g1w::store(shp + wave, acc, lc, lane);
        __syncthreads();
        if (wave == 0) {
            g1w::set_inf(acc);
            for (int k = 0; k < 4; ++k) g1w::dadd(acc, g1w::load(shp + k, lane), lc, sh, lane);
        }
        __syncthreads();  // shp is free again
    };
    // wave 0 publishes acc at `slot` and counts this workgroup at `c`; true for the last of `expect` to arrive
    auto hand_over = [&](Xyzz* slot, u32* c, u32 expect) {
        if (wave == 0) {
            g1w::store(slot, acc, lc, lane);
            __threadfence();
            if (lane == 0) seen_s = atomicAdd(c, 1u);
        }
        __syncthreads();
        const bool last = seen_s == expect - 1;
        __syncthreads();
        if (last) {
            __threadfence();
            if (threadIdx.x == 0) *c = 0;  // ready for the next launch
        }
        return last;
    };
    g1w::set_inf(acc);
    {
        const Xyzz* src = in + set * n + ((size_t)wg * 4 + wave) * (WTREE_PER_WG / 4);
        g1w::add_n(acc, src, 1, WTREE_PER_WG / 4, lc, sh, lane);
    }
    fold4();
    const u32 group = wg / WTREE_GROUP;
    if (!hand_over(p1 + wg, cnt + group, WTREE_GROUP)) return;
    g1w::set_inf(acc);
    g1w::add_n(acc, p1 + (size_t)group * WTREE_GROUP + wave * (WTREE_GROUP / 4), 1, WTREE_GROUP / 4, lc, sh, lane);
    fold4();
    if (ngroups > 1) {
        if (!hand_over(p2 + group, cnt + ngroups, ngroups)) return;
        g1w::set_inf(acc);
        {
            g1w::WPt nx;
            g1w::set_inf(nx);
            if ((u32)wave < ngroups) nx = g1w::load(p2 + wave, lane);
            for (u32 k = wave; k < ngroups; k += 4) {
                const g1w::WPt cur = nx;
                if (k + 4 < ngroups) nx = g1w::load(p2 + k + 4, lane);
                g1w::dadd(acc, cur, lc, sh, lane);
            }
        }
        fold4();
    }
    if (wave == 0) g1w::store(out + set, acc, lc, lane);
}

// ============================ wide fixed-base table ("FBW") ============================
// With 288 GB of HBM per GPU the 4096-point setup can afford the full signed-window table
//     W[w][i][m-1] = m * 2^(c*w) * P_i ,   m = 1 .. 2^(c-1)
// (c = 15: 18 x 4096 x 16384 slots of 128 B = 154 GB; c = 14: 19 x 4096 x 8192 slots = 82 GB).  A commitment is then a plain sum of
// n * ceil(256/c) gathered affine points: no sort, no buckets, no bucket reduction, and every
// lane does the same amount of work whatever the digit distribution.  This is the reference's
// "wbits" fixed-base idea (kzg/src/msm/wbits.rs:357-373 table, :442-488 evaluation) resized
// from CPU-cache-sized windows (w = 8) to HBM-sized ones.

// chain kernel: lane (slot, seg) writes FBW_SEG consecutive multiples of base `slot` as XYZZ
constexpr int FBW_SEG = 64;
__global__ void __launch_bounds__(128) k_fbw_chain(Xyzz* __restrict__ tmp, const AffPt* __restrict__ rows, size_t slot0,
                                                   size_t nslots, size_t mults) {
    const size_t segs = mults / FBW_SEG;
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nslots * segs) return;
    const size_t sl = tid / segs, sg = tid % segs;
    const AffPt base = rows[slot0 + sl];
    Xyzz* out = tmp + (sl * mults + sg * FBW_SEG);
    Xyzz acc;
    g1::set_inf(acc);
    if (base.flags & 1) {
        for (int k = 0; k < FBW_SEG; ++k) out[k] = acc;
        return;
    }
    if (sg != 0) {
        g1::set_affine(acc, base.x, base.y);
        g1::mul_small(acc, (u32)(sg * FBW_SEG));
    }
    for (int k = 0; k < FBW_SEG; ++k) {
        g1::madd(acc, base.x, base.y);
        out[k] = acc;
    }
}

// XYZZ -> affine table slots, one Montgomery batch inversion per FBW_INV consecutive entries
constexpr int FBW_INV = 32;
__global__ void __launch_bounds__(128) k_fbw_affine(WidePt* __restrict__ wide, const Xyzz* __restrict__ tmp,
                                                    fp28::Fe* __restrict__ pref, size_t count) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = tid * FBW_INV;
    if (lo >= count) return;
    size_t hi = lo + FBW_INV < count ? lo + FBW_INV : count;
    fp28::Fe p = fp28::one();
    for (size_t k = lo; k < hi; ++k) {
        pref[k] = p;
        const Xyzz* q = tmp + k;
        if (!fp28::is_zero_limbs(q->zz)) p = fp28::mul(p, fp28::mul(q->zz, q->zzz));
    }
    fp28::Fe inv = g1io::inverse(p);
    for (size_t k = hi; k-- > lo;) {
        Xyzz q = tmp[k];
        WidePt o;
        o.pad[0] = o.pad[1] = o.pad[2] = o.pad[3] = 0;
        if (g1::is_inf(q)) {
            o.x = fp28::zero();
            o.y = fp28::zero();
            o.pad[0] = 1;  // infinity flag
        } else {
            fp28::Fe zi = fp28::mul(inv, pref[k]);                  // 1 / (ZZ*ZZZ)
            inv = fp28::mul(inv, fp28::mul(q.zz, q.zzz));
            o.x = fp28::canon(fp28::mul(q.x, fp28::mul(zi, q.zzz)));  // X / ZZ
            o.y = fp28::canon(fp28::mul(q.y, fp28::mul(zi, q.zz)));   // Y / ZZZ
        }
        wide[k] = o;
    }
}

// r-torsion test of a table base by the endomorphism identity phi(P) == -[x^2]P (the check k_check_commitments
// applies to commitments): decides whether a prepared handle may use the GLV form of the wide table
__global__ void __launch_bounds__(64) k_bases_in_g1(const AffPt* __restrict__ pts, size_t n, int* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AffPt p = pts[i];
    if (p.flags & 1) return;
    const unsigned long long BLS_X = 0xd201000000010000ull;
    Xyzz q1, q2;
    g1::set_inf(q1);
    for (int bit = 63; bit >= 0; --bit) {
        if (!g1::is_inf(q1)) g1::dbl(q1);
        if ((BLS_X >> bit) & 1) g1::madd(q1, p.x, p.y);
    }
    g1::set_inf(q2);
    for (int bit = 63; bit >= 0; --bit) {
        if (!g1::is_inf(q2)) g1::dbl(q2);
        if ((BLS_X >> bit) & 1) g1::dadd(q2, q1);
    }
    if (g1::is_inf(q2)) {
        atomicAdd(bad, 1);
        return;
    }
    fp28::Fe dx = fp28::sub<16>(fp28::mul(fp28::mul(beta28(), p.x), q2.zz), q2.x);
    fp28::Fe dy = fp28::addn(fp28::mul(p.y, q2.zzz), q2.y);
    if (!fp28::is_zero_mod_p(dx) || !fp28::is_zero_mod_p(dy)) atomicAdd(bad, 1);
}

// one lane per (MSM, `spl` consecutive scalars): all windows of those scalars against the wide table.
// spl > 1 (large batches) leaves fewer partial sums for k_blocksum to fold.
// GLV: the table holds rows for 128-bit scalars only (2^(c j) m P_i, j < ceil(128/c)); a scalar is split
// k = +-k1 +- k2 x^2 and the k2 half is accumulated against the SAME table entries: psi(x, y) = (beta x, -y) = [x^2](x, y)
// is a group endomorphism, so  sum k2_digit * psi(T) = psi(sum k2_digit * T)  — the lane that sums the k2 digits of its
// scalars applies psi to its partial sum once (one multiplication); its neighbour sums the k1 digits.  16 additions per scalar
// at c = 16 from a 137 GB table, against 18 at c = 15 from 154 GB without the split.
// GLV form, step 1: one lane per scalar writes its table selectors — for each half (k1, k2) and window the entry
// (|digit| - 1) | sign << 31, or FBW_SKIP for a zero digit; DW = nwin rounded up to a multiple of 4 words per half.  The accumulation kernel then reads 32 bytes per (scalar,
// half) instead of redoing the split in both lanes of a pair and extracting digits from a register array with
// select chains (≈3 % of its instructions).
constexpr u32 FBW_SKIP = 0xffffffffu;
__global__ void __launch_bounds__(256) k_fbw_digits(DigitParams P, const u32* __restrict__ scalars, u32* __restrict__ digits) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.n * P.nbatch) return;
    u32 s[8], s1[8], s2[8], n1, n2;
    load_scalar(s, scalars, t, P.mont);
    kzgamd::glv_split(s, s1, s2, n1, n2);
    const u32 half = 1u << (P.c - 1);
    const int DW = (P.nwin + 3) & ~3;
    const u32 cmask = (1u << P.c) - 1u;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        u32 v[4];  // a half is below 2^128; shifted down one window at a time (static indices only)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = part ? s2[q] : s1[q];
        const u32 pneg = part ? n2 : n1;
        u32 carry = 0;
        uint4* dst = reinterpret_cast<uint4*>(digits + (t * 2 + part) * DW);
        for (int w4 = 0; w4 < DW; w4 += 4) {
            u32 e4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int w = w4 + q;
                u32 e = FBW_SKIP;
                if (w < P.nwin) {
                    u32 d = (v[0] & cmask) + carry;
#pragma unroll
                    for (int x = 0; x < 3; ++x) v[x] = (v[x] >> P.c) | (v[x + 1] << (32 - P.c));
                    v[3] >>= P.c;
                    u32 neg = pneg;
                    carry = 0;
                    if (d > half) {
                        d = (1u << P.c) - d;
                        neg ^= 1;
                        carry = 1;
                    }
                    if (d != 0) e = (d - 1) | (neg << 31);
                }
                e4[q] = e;
            }
            dst[w4 >> 2] = make_uint4(e4[0], e4[1], e4[2], e4[3]);
        }
    }
}

template <int SPL, bool GLV>
__global__ void __launch_bounds__(256) k_fbw_accum(DigitParams P, const u32* __restrict__ scalars,
                                                   const WidePt* __restrict__ wide, Xyzz* __restrict__ partial,
                                                   size_t lanes_per_msm) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes_per_msm * P.nbatch) return;
    const size_t b = t / lanes_per_msm;
    size_t l = t % lanes_per_msm;
    // GLV: neighbouring lanes take the two halves of the same scalars (k1 digits / k2 digits); `scalars` then holds
    // the selectors k_fbw_digits wrote, DW words per (scalar, half)
    const u32 part = GLV ? (u32)(l & 1) : 0u;
    if (GLV) l >>= 1;
    Xyzz acc;
    g1::set_inf(acc);
    const u32 half = 1u << (P.c - 1);
    const int sh = P.c - 1;
    const size_t seg0 = P.nseg ? (b % P.nseg) * P.n : 0;  // first table base of this MSM
    // (the pragma asks; with the addition inlined the body is beyond the unroller's size limit for every SPL > 1 and
    // one copy of it is kept)
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const size_t i = l * (size_t)SPL + k;
        if (SPL > 1 && i >= P.n) break;
        if (GLV) {
            const int DW = (P.nwin + 3) & ~3;
            const uint4* dg = reinterpret_cast<const uint4*>(scalars + ((b * P.n + i) * 2 + part) * DW);
            for (int w4 = 0; w4 < DW; w4 += 4) {
                const uint4 v = dg[w4 >> 2];
                const u32 e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32 e = e4[q];
                    if (e == FBW_SKIP) continue;
                    const WidePt pk = wide[(((size_t)(w4 + q) * P.row_stride + seg0 + i) << sh) + (e & 0x7fffffffu)];
                    if (pk.pad[0]) continue;  // multiple of a base at infinity
                    fp28::Fe x = pk.x, y = pk.y;
                    if (e >> 31) y = fp28::neg<2>(y);
                    g1::madd(acc, x, y);
                }
            }
        } else {
            u32 s[8];
            load_scalar(s, scalars, b * P.n + i, P.mont);
            u32 carry = 0;
            for (int w = 0; w < P.nwin; ++w) {
                u32 d = window_bits(s, w * P.c, P.c) + carry;
                u32 neg = 0;
                carry = 0;
                if (d > half) {
                    d = (1u << P.c) - d;
                    neg = 1;
                    carry = 1;
                }
                if (d == 0) continue;
                const WidePt pk = wide[(((size_t)w * P.row_stride + seg0 + i) << sh) + (d - 1)];
                if (pk.pad[0]) continue;  // multiple of a base at infinity
                fp28::Fe x = pk.x, y = pk.y;
                if (neg) y = fp28::neg<2>(y);
                g1::madd(acc, x, y);
            }
        }
    }
    if (GLV && part && !g1::is_inf(acc)) {
        acc.x = fp28::mul(acc.x, beta28());                     // psi(X, Y, ZZ, ZZZ) = (beta X, -Y, ZZ, ZZZ)
        acc.y = fp28::mul(fp28::neg<8>(acc.y), fp28::one());    // -Y, back under the 2p bound
    }
    partial[t] = acc;
}

// k_fbw_accum<1, true> with FOUR lanes per (scalar, half) chain (g1grp.hip.h): the eight mixed additions of a chain are
// 4 multiplications deep instead of 10, on four times the lanes — for the few commitments of a latency-bound call
// (8192 chains per commitment: 128 waves leave most of the chip idle, 512 waves still do).  ONE loop with ONE inlined
// addition in its body.  Role 0 of a group writes the chain's sum.
__global__ void __launch_bounds__(256) k_fbw_accum_quad(DigitParams P, const u32* __restrict__ digits,
                                                        const WidePt* __restrict__ wide, Xyzz* __restrict__ partial,
                                                        size_t lanes_per_msm) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(threadIdx.x & 3);
    const size_t chain = t >> 2;
    if (chain >= lanes_per_msm * P.nbatch) return;  // a whole group leaves: chains never straddle a quad
    const size_t b = chain / lanes_per_msm;
    const size_t l = chain % lanes_per_msm;
    const u32 part = (u32)(l & 1);
    const size_t i = l >> 1;
    Xyzz acc;
    g1::set_inf(acc);
    const int sh = P.c - 1;
    const size_t seg0 = P.nseg ? (b % P.nseg) * P.n : 0;
    const int DW = (P.nwin + 3) & ~3;
    const uint4* dg = reinterpret_cast<const uint4*>(digits + ((b * P.n + i) * 2 + part) * DW);
    uint4 v = make_uint4(FBW_SKIP, FBW_SKIP, FBW_SKIP, FBW_SKIP);
#pragma unroll 1
    for (int w = 0; w < DW; ++w) {
        if ((w & 3) == 0) v = dg[w >> 2];
        const int k = w & 3;
        const u32 e = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
        if (e == FBW_SKIP) continue;
        const WidePt pk = wide[(((size_t)w * P.row_stride + seg0 + i) << sh) + (e & 0x7fffffffu)];
        if (pk.pad[0]) continue;  // multiple of a base at infinity
        fp28::Fe x = pk.x, y = pk.y;
        if (e >> 31) y = fp28::neg<2>(y);
        grp::madd_body4(acc, x, y, r);
    }
    if (part && !g1::is_inf(acc)) {
        acc.x = fp28::mul(acc.x, beta28());                     // psi(X, Y, ZZ, ZZZ) = (beta X, -Y, ZZ, ZZZ)
        acc.y = fp28::mul(fp28::neg<8>(acc.y), fp28::one());    // -Y, back under the 2p bound
    }
    if (r == 0) partial[chain] = acc;
}

// one workgroup per MSM: plain sum of its n partial sums
__global__ void __launch_bounds__(256) k_blocksum(const Xyzz* __restrict__ in_all, Xyzz* __restrict__ out, size_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Xyzz* sh = reinterpret_cast<Xyzz*>(smem);
    const size_t set = blockIdx.x;
    const Xyzz* in = in_all + set * n;
    Xyzz acc;
    g1::set_inf(acc);
    for (size_t k = threadIdx.x; k < n; k += blockDim.x) {
        Xyzz b = in[k];
        g1::dadd(acc, b);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int stride = blockDim.x / 2; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride) {
            Xyzz b = sh[threadIdx.x + stride];
            g1::dadd(acc, b);
            sh[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[set] = acc;
}

// The fold of a few MSMs' partial sums (2 .. 16 commitments per call: the batches of concurrent callers), shaped by
// the two addition latencies: a single-lane XYZZ addition is a ~16 us chain whatever the number of live lanes, a
// limb-parallel one (g1w) ~2.7 us but takes the whole wave.  BSH_PARTS workgroups per MSM: every lane adds its strided
// share, three single-lane tree rounds (256 -> 32 sums), then each wave adds 8 of them limb-parallel, wave 0 adds the
// four results and the workgroup's sum goes to `part`; the last workgroup of an MSM to finish (a counter per MSM,
// reset for the next call) adds the BSH_PARTS sums.  One launch, 1 + 3 single-lane and 8 + 4 + 16 limb-parallel
// additions deep, against 1 + 8 and then 1 + 6 single-lane ones in two launches of k_blocksum.
constexpr int BSH_PARTS = 16;
constexpr size_t BSH_MAX = 64;   // MSMs per call folded this way by default (tuning key hybrid_max)
constexpr size_t BSH_CAP = 128;  // ... at most
__global__ void __launch_bounds__(256) k_blocksum_hybrid(const Xyzz* __restrict__ in_all, Xyzz* __restrict__ out,
                                                         Xyzz* __restrict__ part, u32* __restrict__ counter, size_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Xyzz* sh = reinterpret_cast<Xyzz*>(smem);  // 256 sums, then 4 wave sums at [64 ..]
    __shared__ u32 scr[4][16];
    const size_t set = blockIdx.x / BSH_PARTS;
    const int sub = (int)(blockIdx.x % BSH_PARTS);
    const size_t per = n / BSH_PARTS;
    const Xyzz* in = in_all + set * n + (size_t)sub * per;
    Xyzz acc;
    g1::set_inf(acc);
    for (size_t k = threadIdx.x; k < per; k += blockDim.x) {
        Xyzz b = in[k];
        g1::dadd(acc, b);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int stride = 128; stride >= 32; stride >>= 1) {
        if ((int)threadIdx.x < stride) {
            Xyzz b = sh[threadIdx.x + stride];
            g1::dadd(acc, b);
            sh[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const fpw::Lane lc = fpw::lane_consts(lane);
    g1w::WPt w;
    g1w::set_inf(w);
    for (int k = 0; k < 8; ++k) g1w::dadd(w, g1w::load(sh + wave * 8 + k, lane), lc, scr[wave], lane);
    g1w::store(sh + 64 + wave, w, lc, lane);
    __syncthreads();
    if (wave != 0) return;
    g1w::set_inf(w);
    for (int k = 0; k < 4; ++k) g1w::dadd(w, g1w::load(sh + 64 + k, lane), lc, scr[0], lane);
    g1w::store(part + set * BSH_PARTS + sub, w, lc, lane);
    __threadfence();
    u32 seen = 0;
    if (lane == 0) seen = atomicAdd(counter + set, 1u);
    seen = (u32)__builtin_amdgcn_readfirstlane((int)seen);
    if (seen != (u32)(BSH_PARTS - 1)) return;
    __threadfence();
    g1w::set_inf(w);
    for (int k = 0; k < BSH_PARTS; ++k) g1w::dadd(w, g1w::load(part + set * BSH_PARTS + k, lane), lc, scr[0], lane);
    g1w::store(out + set, w, lc, lane);
    if (lane == 0) counter[set] = 0;
}

// out[m] = sum of the n (<= 64) consecutive partial sums of MSM m: one lane per MSM
__global__ void __launch_bounds__(64) k_lane_sum(const Xyzz* __restrict__ in, Xyzz* __restrict__ out, size_t n, size_t count) {
    const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= count) return;
    Xyzz acc = in[m * n];
    for (size_t k = 1; k < n; ++k) {
        Xyzz b = in[m * n + k];
        g1::dadd(acc, b);
    }
    out[m] = acc;
}

// device copy of already-converted table slots (row 0)
__global__ void __launch_bounds__(256) k_copy_affpt(AffPt* __restrict__ dst, const AffPt* __restrict__ src, size_t n,
                                                    int glv) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    AffPt a = src[i];
    dst[i] = a;
    if (glv) dst[n + i] = x2_image(a);
}

// P_i = h_i * G with h_i a 248-bit value from splitmix64(seed, i); output in blst affine layout
__device__ __forceinline__ u64 splitmix64(u64& x) {
    x += 0x9e3779b97f4a7c15ull;
    u64 z = x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(128) k_gen_points(ff::Fp* __restrict__ out, size_t n, u64 seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 st = seed ^ (0xd1b54a32d192ed03ull * (u64)(i + 1));
    u64 h[4];
    for (int k = 0; k < 4; ++k) h[k] = splitmix64(st);
    h[3] &= 0x00ffffffffffffffull;  // < 2^248 < r
    // generator in blst layout (blst/src/consts.rs:52-84)
    ff::Fp gx, gy;
    const u64 GX[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull,
                       0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
    const u64 GY[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull,
                       0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
    for (int k = 0; k < 6; ++k) {
        gx.v[2 * k] = (u32)GX[k];
        gx.v[2 * k + 1] = (u32)(GX[k] >> 32);
        gy.v[2 * k] = (u32)GY[k];
        gy.v[2 * k + 1] = (u32)(GY[k] >> 32);
    }
    fp28::Fe x = fp28::canon(fp28::from_blst(gx)), y = fp28::canon(fp28::from_blst(gy));
    Xyzz acc;
    g1::set_inf(acc);
    for (int bit = 247; bit >= 0; --bit) {
        if (!g1::is_inf(acc)) g1::dbl(acc);
        if ((h[bit >> 6] >> (bit & 63)) & 1) g1::madd(acc, x, y);
    }
    if (g1::is_inf(acc)) {
        out[2 * i] = ff::Fp::zero();
        out[2 * i + 1] = ff::Fp::zero();
        return;
    }
    fp28::Fe zi = g1io::inverse(fp28::mul(acc.zz, acc.zzz));
    out[2 * i] = fp28::to_blst(fp28::mul(acc.x, fp28::mul(zi, acc.zzz)));
    out[2 * i + 1] = fp28::to_blst(fp28::mul(acc.y, fp28::mul(zi, acc.zz)));
}

// ---------------------------------------------------------------- host side

// HBM the wide table may take: the handle's configured budget (Options::table_budget_gb; < 0 = the default below),
// capped by what is free right now (leaving room for the build scratch and the per-call workspaces)
double fbw_budget_gb(double configured) {
    double budget_gb = configured >= 0 ? configured : FBW_DEFAULT_GB;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const double room = ((double)free_b - 12e9) / 1.05 / 1e9;
        if (room < budget_gb) budget_gb = room;
    }
    return budget_gb;
}

// wide-table shape for a prepared handle: the (window, split) pair with the fewest additions per scalar whose table
// fits the HBM budget.  Without the split a scalar costs rows = ceil(256/c) additions from a table of
// rows * n * 2^(c-1) slots; with the GLV split (bases must be in the r-torsion subgroup) 2 * ceil(128/c) additions from
// a table of ceil(128/c) * n * 2^(c-1) slots.  Returns 0 when nothing fits.
int choose_wide_window(size_t n, bool glv_allowed, bool* use_glv, double configured_gb) {
    const double budget_gb = fbw_budget_gb(configured_gb);
    int best_c = 0, best_adds = 1 << 30;
    double best_gb = 0;
    bool best_glv = false;
    for (int g = 0; g <= (glv_allowed ? 1 : 0); ++g)
        for (int c = 10; c <= 18; ++c) {
            const int rows = g ? (127 + c) / c : 255 / c + 1, adds = g ? 2 * rows : rows;
            const double gb = (double)rows * (double)n * (double)((size_t)1 << (c - 1)) * 128.0 / 1e9;
            if (gb > budget_gb) continue;
            if (adds < best_adds || (adds == best_adds && gb < best_gb)) {
                best_adds = adds;
                best_c = c;
                best_gb = gb;
                best_glv = g != 0;
            }
        }
    *use_glv = best_glv;
    return best_c;
}

int choose_window(size_t n, bool prepared, bool glv, int forced) {
    // minimise adds: prepared  n*ceil(256/c) + 3*2^(c-1);  unprepared  ceil(256/c) * (n + 3*2^(c-1));
    // unprepared with the GLV split: (128/c + 1) bucket sets fed by 2n half-scalars
    // forced: tuning keys window / window_prepared as read when the handle was created (0 = by size)
    if (forced >= 2 && forced <= 22) return forced;
    if (glv) {
        // measured on MI355X (tools/sweep_window.py, device-resident inputs): c = 16 wins from n = 2^15 up to at
        // least 2^22 — eight windows with a full top window after the balanced split; smaller n are latency-bound
        // and prefer fewer buckets
        if (n >= ((size_t)1 << 15)) return 16;
        if (n >= ((size_t)1 << 12)) return 13;  // 2^12: 0.81 ms against 0.91 with c = 10
        if (n >= ((size_t)1 << 10)) return 10;
    }
    int best = 2;
    double best_cost = 1e300;
    for (int c = 2; c <= 22; ++c) {
        double w = glv ? (127 + c) / c : 255 / c + 1, nb = (double)((size_t)1 << (c - 1));
        double cost = prepared ? w * (double)n + 3.0 * nb : w * ((glv ? 2.0 : 1.0) * (double)n + 3.0 * nb);
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    void ensure(size_t n) {
        if (n <= cap) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        HIP_TRY(hipMalloc(&p, n * sizeof(T)));
        cap = n;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Workspace {
    DevBuf<u32> counts, offsets, sorted, scalars, ranks, tmp, bins, digits, wghist;
    DevBuf<Xyzz> buckets, lvlA[2], lvlM[2], top, win, dense, wpart;
    DevBuf<u32> wcount;
    DevBuf<u32> bcount;  // k_blocksum_hybrid's counters: zeroed when allocated, left at zero by every launch
    DevBuf<Xyzz> bpart;
    DevBuf<unsigned char> heavy;
    DevBuf<u32> heavy_list, nheavy;
    DevBuf<ff::Fp> out;
    // the last enqueue that used this workspace: another stream that is handed the same workspace waits for it
    hipEvent_t last_done = nullptr;
    hipStream_t last_stream = nullptr;
    void release() {
        if (last_done) (void)hipEventDestroy(last_done);
        last_done = nullptr;
        counts.release();
        digits.release();
        wghist.release();
        offsets.release();
        sorted.release();
        ranks.release();
        tmp.release();
        bins.release();
        scalars.release();
        buckets.release();
        for (int k = 0; k < 2; ++k) {
            lvlA[k].release();
            lvlM[k].release();
        }
        top.release();
        dense.release();
        wpart.release();
        wcount.release();
        bcount.release();
        bpart.release();
        win.release();
        heavy.release();
        heavy_list.release();
        nheavy.release();
        out.release();
    }
};

}  // namespace

// The tuning keys of config.h this engine reads, copied ONCE when a handle is created (an enqueue never looks anything
// up); every variant computes the same result a different way.
struct MsmTuning {
    int spl = 0, hybrid_max = 0, wide_fold_max = 0, spl1_max = 0, blocksum_threads = 0, lgc = 0, groups = 0, fine_bits = 0;
    bool no_wide_tail = false, no_hybrid_fold = false, no_wide_tree = false;
    int quad_accum_max = 4;
    bool one_level_sort = false, tree_tail = false, flat_digits = false, direct_scatter = false, scatter_atomics = false;
    bool combine = true;
    int combine_lanes = 3, combine_gather_min = 6, combine_gather_us = 60;
    int tail_pieces = 0, sub_streams = 6, tile_rows = 0, tile_quad = 1, digit_min_log = 14, sub_prio = 1, sub_large = 0, sort_ahead = 0;
    static MsmTuning from(const kzgamd::Options& o) {
        using namespace kzgamd;
        MsmTuning t;
        t.spl = (int)o.t[T_SPL];
        t.no_wide_tail = o.t[T_NO_WIDE_TAIL] != 0;
        t.no_hybrid_fold = o.t[T_NO_HYBRID_FOLD] != 0;
        t.no_wide_tree = o.t[T_NO_WIDE_TREE] != 0;
        t.quad_accum_max = (int)o.t[T_QUAD_ACCUM_MAX];
        t.hybrid_max = (int)o.t[T_HYBRID_MAX];
        t.wide_fold_max = (int)o.t[T_WIDE_FOLD_MAX];
        t.spl1_max = (int)o.t[T_SPL1_MAX];
        t.blocksum_threads = (int)o.t[T_BLOCKSUM_THREADS];
        t.lgc = (int)o.t[T_LGC];
        t.groups = (int)o.t[T_GROUPS];
        t.fine_bits = (int)o.t[T_FINE_BITS];
        t.one_level_sort = o.t[T_ONE_LEVEL_SORT] != 0;
        t.tree_tail = o.t[T_TREE_TAIL] != 0;
        t.flat_digits = o.t[T_FLAT_DIGITS] != 0;
        t.direct_scatter = o.t[T_DIRECT_SCATTER] != 0;
        t.scatter_atomics = o.t[T_SCATTER_ATOMICS] != 0;
        t.combine = o.t[T_COMBINE] != 0;
        t.combine_lanes = (int)o.t[T_COMBINE_LANES];
        t.combine_gather_min = (int)o.t[T_COMBINE_GATHER_MIN];
        t.combine_gather_us = (int)o.t[T_COMBINE_GATHER_US];
        t.tail_pieces = (int)o.t[T_TAIL_PIECES];
        t.sub_streams = (int)o.t[T_SUB_STREAMS];
        t.tile_rows = (int)o.t[T_TILE_ROWS];
        t.tile_quad = (int)o.t[T_TILE_QUAD];
        t.digit_min_log = (int)o.t[T_DIGIT_MIN_LOG];
        t.sub_prio = (int)o.t[T_SUB_PRIO];
        t.sub_large = (int)o.t[T_SUB_LARGE];
        t.sort_ahead = (int)o.t[T_SORT_AHEAD];
        return t;
    }
};

// The scalars of a combined batch stay where their callers staged them (page-locked slots, one per caller): one kernel
// fetches them over PCIe into the batch's contiguous buffer instead of one copy operation per request on the stream.
struct ScalarSlots {
    const uint4* p[32];  // MsmContext::COMBINE_MAX
};
__global__ void __launch_bounds__(256) k_gather_scalars(uint4* __restrict__ dst, ScalarSlots slots, size_t per, size_t nreq) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nreq * per) return;
    dst[t] = slots.p[t / per][t % per];
}

struct kzgamd::MsmContext {
    std::mutex mu;
    kzgamd::Options opt;
    MsmTuning tune;
    int window_forced = 0;  // tuning keys window (variable base) / window_prepared at creation, 0 = by size
    size_t mat_rows = 0, mat_cols = 0;  // matrix handle (kzgamd_prepare_msm_matrix): rows base sets of cols points
    MsmContext* matrix = nullptr;       // kzgamd_msm_attach_matrix: a matrix handle owned by this one
    int device = 0;
    size_t n = 0;
    bool prepared = false;
    bool glv = false;  // variable-base engine: scalars split k = k1 + k2*x^2, table = P_i followed by [x^2]P_i
    int c = 0, rows = 0, nwin = 0;
    size_t nb = 0;
    DevBuf<AffPt> table;  // rows x n (prepared) or n
    DevBuf<WidePt> wide;  // wide fixed-base table: (rows x n) x 2^(c-1) 128-byte slots, when it fits the budget
    bool fbw = false;
    bool fbw_glv = false;  // the wide table covers 128-bit half-scalars (rows = ceil(128 / c)); scalars are GLV-split
    Workspace ws;
    // One workspace per stream the handle is used on (up to 24): independent batches enqueued on different streams
    // may overlap on the GPU — the low-occupancy tail of one batch under the accumulation of the next.  `ws` serves
    // the handle's own stream and the first caller stream; further streams get their own.
    hipStream_t ws_owner = nullptr;
    bool ws_owned = false;
    std::vector<std::pair<hipStream_t, Workspace*>> ws_extra;
    Workspace& workspace_for(hipStream_t st) {
        if (st == stream) return ws;  // host-buffer entry points (synchronised internally)
        if (!ws_owned) {
            ws_owned = true;
            ws_owner = st;
        }
        if (st == ws_owner) return ws;
        for (auto& e : ws_extra)
            if (e.first == st) return *e.second;
        if (ws_extra.size() >= 23) return ws;  // more streams than workspaces: shared, see WsUse in msm_enqueue
        ws_extra.emplace_back(st, new Workspace());
        return *ws_extra.back().second;
    }
    hipStream_t stream = nullptr;
    // A batch of large MSMs runs as sub-batches (tuning key sub_streams).  The accumulations — VALU throughput, the chip
    // full — stay on the caller's stream, one after the other; everything else of a sub-batch (its sort before, its
    // reduction chains and Horner after: atomics / latency, a fraction of the chip) runs on one of these side streams,
    // which rotate over the sub-batches, each with its own workspace.  The side streams have the highest priority the
    // device offers: at equal priority the accumulation's waves hold every register of every SIMD and a reduction
    // kernel launched beside it waits for slots (traced in round 6: k_tile_sums_loop 2.8 ms instead of 0.32, the next
    // sub-batch's accumulation waiting behind it).  accum_on / ev_sorted / ev_accd are what msm_enqueue's k_accum launch
    // site reads.
    static constexpr int MAXSUB = 6;
    hipStream_t sub_stream[MAXSUB] = {};
    hipEvent_t ev_sub_fork = nullptr, ev_sub_join[MAXSUB] = {}, ev_sorted[MAXSUB] = {}, ev_accd[MAXSUB] = {}, ev_gate[MAXSUB] = {};
    bool accum_redirect = false;     // set by the sub-batch loop: this enqueue's k_accum goes to accum_on ...
    bool tail_on_accum = false;      // ... and so does everything after it (sort-ahead form: only the SORT stays on the side stream)
    hipStream_t accum_on = nullptr;  // ... the caller's stream (which may be the null stream)
    hipEvent_t accum_ready = nullptr, accum_done = nullptr;
    hipEvent_t sort_gate = nullptr;  // sort-ahead form: recorded where the NEXT sub-batch's sort may start (after this one's tile sums)
    bool in_sub_batch = false;  // the enqueue in progress is a sub-batch: it is not cut again
    // window-group pipeline of the variable-base engine: one auxiliary stream per group, events to fork from /
    // join into the caller's stream and to order the digit and accumulation kernels across groups
    static constexpr int MAXG = 4;
    hipStream_t aux[MAXG] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_dig[MAXG] = {}, ev_acc[MAXG] = {}, ev_done[MAXG] = {};
    bool profile = false;
    // per enqueue: start, accum-begin, accum-end, end (events on the launch stream)
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    static constexpr size_t EV_MAX = 4 * 512;
    // Combining of concurrent host-buffer calls on ONE prepared handle.  The reference shares the handle between rayon
    // workers (SpparkPrecomputation is Send + Sync, kzg/src/msm/sppark.rs:24-44; verify_blob_kzg_proof_batch calls the
    // MSM from par_chunks, kzg/src/eip_4844.rs:781-805): callers of mult_pippenger_prepared queue a request; one of
    // them at a time (the leader) takes everything queued with the same length, up to COMBINE_MAX requests, and runs
    // it as ONE nbatch launch — the ~10 runtime operations of an invocation are paid per batch, not per call, and
    // the GPU sees a few thousand waves instead of a few hundred.  A caller returns as soon as its own request is
    // served; leadership passes to whoever is waiting.  Each caller copies its scalars into a page-locked slot on its
    // own thread before it queues (truly asynchronous H2D copies; a copy from pageable memory goes through the
    // runtime's staging path, which serialises concurrent callers).
    struct HostCall {
        void* out;
        const void* scalars;
        size_t npoints;
        unsigned char* slot = nullptr;
        bool done = false, failed = false;
        HipErr err{hipSuccess, ""};
    };
    static constexpr size_t COMBINE_MAX = 32;  // = the pointers of ScalarSlots
    static constexpr int COMBINE_SLOTS = 48;
    // Calls of more scalars than this take the plain path: their kernels run for a millisecond and more, next to which
    // the per-invocation cost the combiner removes is nothing — and 48 slots of the handle's full length would pin
    // 1.6 GB of host memory for a 2^20-point handle.
    static constexpr size_t COMBINE_NMAX = (size_t)1 << 16;
    // Up to COMBINE_LANES batches are in flight at once, each on a lane of its own (stream, staging, and — through
    // workspace_for(stream) — MSM workspace): the copies and launches of one batch are issued while the kernels of the
    // previous one run.  The handle's mutex is held while a batch is enqueued, not while it is awaited.
    static constexpr int COMBINE_LANES = 4;
    struct CombineLane {
        hipStream_t st = nullptr;
        DevBuf<u32> scalars;
        DevBuf<ff::Fp> out;
        unsigned char* h_out = nullptr;  // COMBINE_MAX x 144, page-locked
        bool busy = false;
    };
    struct Combine {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<HostCall*> pending;
        int leaders = 0;
        CombineLane lanes[COMBINE_LANES];
        // page-locked staging slots, allocated in chunks as callers need them (4, 4, 8, 16, 16): a handle that one thread
        // uses at a time pins four slots (a single caller keeps the fast staging path: a copy from a page-locked slot
        // instead of the runtime's staging of pageable memory), sixteen concurrent callers grow the pool to what they use
        std::vector<unsigned char*> slot_chunks;
        int slots_allocated = 0;
        size_t slot_bytes = 0;
        bool pinned_failed = false;
        std::vector<unsigned char*> free_slots;
    } comb;
    ~MsmContext() {
        delete matrix;
        for (auto* c : comb.slot_chunks) (void)hipHostFree(c);
        for (auto& l : comb.lanes) {
            if (l.h_out) (void)hipHostFree(l.h_out);
            l.scalars.release();
            l.out.release();
            if (l.st) (void)hipStreamDestroy(l.st);
        }
        table.release();
        wide.release();
        ws.release();
        for (auto& e : ws_extra) {
            e.second->release();
            delete e.second;
        }
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_sub_fork) (void)hipEventDestroy(ev_sub_fork);
        for (int k = 0; k < MAXSUB; ++k) {
            if (ev_sub_join[k]) (void)hipEventDestroy(ev_sub_join[k]);
            if (ev_sorted[k]) (void)hipEventDestroy(ev_sorted[k]);
            if (ev_accd[k]) (void)hipEventDestroy(ev_accd[k]);
            if (ev_gate[k]) (void)hipEventDestroy(ev_gate[k]);
            if (sub_stream[k]) (void)hipStreamDestroy(sub_stream[k]);
        }
        for (int g = 0; g < MAXG; ++g) {
            if (ev_dig[g]) (void)hipEventDestroy(ev_dig[g]);
            if (ev_acc[g]) (void)hipEventDestroy(ev_acc[g]);
            if (ev_done[g]) (void)hipEventDestroy(ev_done[g]);
            if (aux[g]) (void)hipStreamDestroy(aux[g]);
        }
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace kzgamd {

static void require_device() {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0) throw HipErr{e == hipSuccess ? hipErrorNoDevice : e, "no gfx950 device visible"};
}

// Wide table: built tile by tile (chain of multiples as XYZZ -> batch-inverted affine slots).
static void build_wide_table(MsmContext* ctx) {
    const double budget_gb = fbw_budget_gb(ctx->opt.table_budget_gb);
    const size_t mults = ctx->nb;  // 2^(c-1)
    const size_t nslots = (size_t)ctx->rows * ctx->n;
    const double gb = (double)nslots * (double)mults * sizeof(WidePt) / 1e9;
    if (budget_gb <= 0 || gb > budget_gb || mults < (size_t)FBW_SEG) return;
    ctx->wide.ensure(nslots * mults);
    // tile: up to 2^22 table entries at a time (XYZZ 224 B + prefix 56 B of scratch each)
    size_t tile_slots = ((size_t)1 << 22) / mults;
    if (tile_slots == 0) tile_slots = 1;
    if (tile_slots > nslots) tile_slots = nslots;
    DevBuf<Xyzz> tmp;
    DevBuf<fp28::Fe> pref;
    tmp.ensure(tile_slots * mults);
    pref.ensure(tile_slots * mults);
    for (size_t s0 = 0; s0 < nslots; s0 += tile_slots) {
        const size_t ns = s0 + tile_slots <= nslots ? tile_slots : nslots - s0;
        const size_t chain_threads = ns * (mults / FBW_SEG), cnt = ns * mults;
        hipLaunchKernelGGL(k_fbw_chain, dim3((unsigned)((chain_threads + 127) / 128)), dim3(128), 0, ctx->stream, tmp.p,
                           (const AffPt*)ctx->table.p, s0, ns, mults);
        const size_t inv_threads = (cnt + FBW_INV - 1) / FBW_INV;
        hipLaunchKernelGGL(k_fbw_affine, dim3((unsigned)((inv_threads + 127) / 128)), dim3(128), 0, ctx->stream,
                           ctx->wide.p + s0 * mults, (const Xyzz*)tmp.p, pref.p, cnt);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    tmp.release();
    pref.release();
    ctx->fbw = true;
}

// uploads points (host or device pointer), builds the fixed-base rows when `prepare`
MsmContext* msm_create(const void* points, size_t n, bool points_on_device, bool prepare, bool points_are_affpt, int g1_policy,
                       const Options* opt) {
    require_device();
    auto* ctx = new MsmContext();
    static const bool verbose = getenv("KZGAMD_VERBOSE") != nullptr;
    try {
        if (opt) {
            ctx->opt = *opt;
        } else {
            std::string err;
            if (!Options::resolve(ctx->opt, nullptr, &err)) throw HipErr{hipErrorInvalidValue, "KZGAMD_TUNING does not parse"};
        }
        ctx->tune = MsmTuning::from(ctx->opt);
        // the handle lives on opt.device when one is named; the caller's current device is restored on the way out
        int cur_dev = 0;
        HIP_TRY(hipGetDevice(&cur_dev));
        DeviceGuard placed(ctx->opt.device >= 0 ? ctx->opt.device : cur_dev);
        HIP_TRY(placed.err);
        HIP_TRY(hipGetDevice(&ctx->device));
        HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->n = n;
        ctx->prepared = prepare;
        ctx->window_forced = (int)ctx->opt.t[prepare ? T_WINDOW_PREPARED : T_WINDOW];
        // The variable-base engine's GLV split (k = k1 + k2 x^2, second base psi(P) = [x^2]P) is an identity of the
        // r-torsion subgroup only, and the reference's G1::from_bytes accepts any curve point
        // (blst/src/types/g1.rs:65-87): the split is used when the caller vouches for the bases (internal callers
        // that have just subgroup-checked them), or after every base has passed the membership test here; otherwise
        // the engine runs on the 255-bit scalars.
        // A prepared handle too large for any wide table (n >= 2^19 with the default budget) would run the bucket engine
        // over table rows 2^(c j) P with one bucket set.  Measured (tools/time_prepared.py, 2^20 / 2^21 / 2^22 points):
        // 4.57 / 9.7 / 16.8 ms against 3.51 / 6.5 / 13.1 ms for the GLV-split engine on the plain bases — the rows make
        // the gathers of the accumulation miss every cache (2 GB of table) and one set of 2^19 buckets costs the reduction
        // more than eight sets of 2^15 (profiles/NOTES.md §9) — so such a handle takes the variable-base shape when its bases pass
        // the subgroup test.  Tuning key fixed_as_variable_min: log2 of the smallest such n (0 = never).
        bool as_variable = false;
        if (prepare && !ctx->window_forced && g1_policy != G1_NO_SPLIT) {
            const int lg = (int)ctx->opt.t[T_FIXED_AS_VARIABLE_MIN];
            bool dummy;
            as_variable = lg > 0 && lg < 40 && n >= ((size_t)1 << lg) && choose_wide_window(n, true, &dummy, ctx->opt.table_budget_gb) == 0;
        }
        ctx->glv = (!prepare || as_variable) && g1_policy != G1_NO_SPLIT;
        ctx->glv = ctx->glv && ctx->opt.t[T_GLV] != 0;
        if (!ctx->glv) as_variable = false;
        // the bases first (row 0 of the table; the variable-base engine appends the [x^2]P images)
        DevBuf<AffPt> row0;
        row0.ensure(ctx->glv ? 2 * n : n);
        DevBuf<ff::Fp> staging;
        const ff::Fp* src = (const ff::Fp*)points;
        if (!points_are_affpt && !points_on_device) {
            staging.ensure(2 * n);
            HIP_TRY(hipMemcpyAsync(staging.p, points, n * 96, hipMemcpyHostToDevice, ctx->stream));
            src = staging.p;
        }
        if (points_are_affpt && !points_on_device) throw HipErr{hipErrorInvalidValue, "AffPt input must be device-resident"};
        auto bases_in = [&](int with_images) {
            if (points_are_affpt)
                hipLaunchKernelGGL(k_copy_affpt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, row0.p,
                                   (const AffPt*)points, n, with_images);
            else
                hipLaunchKernelGGL(k_points_in, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, row0.p, src, n,
                                   with_images);
        };
        if (ctx->glv && g1_policy == G1_CHECK) {
            bases_in(0);
            DevBuf<int> bad;
            bad.ensure(1);
            HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream));
            hipLaunchKernelGGL(k_bases_in_g1, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const AffPt*)row0.p, n,
                               bad.p);
            int nbad = 1;
            HIP_TRY(hipMemcpyAsync(&nbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            bad.release();
            if (nbad != 0) ctx->glv = false;  // a base outside G1: no split (row0 keeps its 2n slots, n are used)
        }
        if (as_variable && ctx->glv) {
            prepare = false;
            ctx->prepared = false;
            if (verbose)
                fprintf(stderr, "kzg_mi355x: prepared handle over %zu points on GPU %d: no wide table fits, GLV-split bucket engine on the plain bases\n",
                        n, ctx->device);
        }
        bases_in(ctx->glv ? 1 : 0);
        // shape of the engine
        bool wide_glv = false;
        int cw = 0;
        if (prepare && !ctx->window_forced) {
            bool dummy;
            if (choose_wide_window(n, true, &dummy, ctx->opt.table_budget_gb) != 0) {
                // a wide table fits: its GLV form needs every base in the r-torsion subgroup (psi(P) = [x^2]P holds only
                // there; FsG1::from_bytes does not check it, blst/src/types/g1.rs:65-87) — test the bases, once
                bool in_g1 = false;
                if (ctx->opt.t[T_FBW_GLV] != 0) {
                    DevBuf<int> bad;
                    bad.ensure(1);
                    HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream));
                    hipLaunchKernelGGL(k_bases_in_g1, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream,
                                       (const AffPt*)row0.p, n, bad.p);
                    int nbad = 1;
                    HIP_TRY(hipMemcpyAsync(&nbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                    HIP_TRY(hipStreamSynchronize(ctx->stream));
                    bad.release();
                    in_g1 = nbad == 0;
                }
                cw = choose_wide_window(n, in_g1, &wide_glv, ctx->opt.table_budget_gb);
            }
        }
        if (cw) {
            ctx->c = cw;
            ctx->fbw_glv = wide_glv;
            ctx->rows = wide_glv ? (127 + cw) / cw : 255 / cw + 1;
            ctx->nwin = ctx->rows;
        } else {
            ctx->c = choose_window(n, prepare, ctx->glv, ctx->window_forced);
            ctx->rows = prepare ? (255 / ctx->c + 1) : 1;
            ctx->nwin = ctx->glv ? (127 + ctx->c) / ctx->c : 255 / ctx->c + 1;
        }
        ctx->nb = (size_t)1 << (ctx->c - 1);
        ctx->table.ensure(ctx->glv ? 2 * n : (size_t)ctx->rows * n);
        HIP_TRY(hipMemcpyAsync(ctx->table.p, row0.p, (ctx->glv ? 2 * n : n) * sizeof(AffPt), hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        row0.release();
        if (prepare && ctx->rows > 1)
            hipLaunchKernelGGL(k_table_rows, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, ctx->table.p, n,
                               ctx->rows, ctx->c);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        staging.release();
        if (prepare) build_wide_table(ctx);
        if (ctx->fbw_glv && !ctx->fbw) throw HipErr{hipErrorOutOfMemory, "wide GLV table did not fit the HBM budget it was sized for"};
        if (prepare && verbose) {
            // the shape a handle ended up with depends on the HBM that was free: say so when asked
            const double gb = ctx->fbw ? (double)ctx->rows * (double)n * (double)ctx->nb * sizeof(WidePt) / 1e9 : 0.0;
            fprintf(stderr, "kzg_mi355x: prepared handle over %zu points on GPU %d: %s, %d-bit windows, %d rows, %s, %.1f GB, %d additions per scalar\n",
                    n, ctx->device, ctx->fbw ? "wide table" : "bucket engine (no room for a wide table)", ctx->c, ctx->rows,
                    ctx->fbw_glv ? "GLV split" : "no split", gb, ctx->fbw ? (ctx->fbw_glv ? 2 * ctx->rows : ctx->rows) : 0);
        }
    } catch (...) {
        delete ctx;
        throw;
    }
    return ctx;
}

void msm_reset_points(MsmContext* ctx, const void* d_affpts, size_t n) {
    if (!ctx || ctx->prepared) throw HipErr{hipErrorInvalidValue, "msm_reset_points: not a variable-base handle"};
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard on_device(ctx->device);
    HIP_TRY(on_device.err);
    ctx->n = n;
    ctx->c = choose_window(n, false, ctx->glv, ctx->window_forced);
    ctx->rows = 1;
    ctx->nwin = ctx->glv ? (127 + ctx->c) / ctx->c : 255 / ctx->c + 1;
    ctx->nb = (size_t)1 << (ctx->c - 1);
    ctx->table.ensure(ctx->glv ? 2 * n : n);
    hipLaunchKernelGGL(k_copy_affpt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->table.p,
                       (const AffPt*)d_affpts, n, ctx->glv ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the caller's buffer may be reused as soon as this returns
}

void msm_destroy(MsmContext* ctx) {
    if (!ctx) return;
    DeviceGuard on_device(ctx->device);
    delete ctx;
}
int msm_device(MsmContext* ctx) { return ctx->device; }
bool msm_has_wide_table(MsmContext* ctx) { return ctx->fbw; }
bool msm_private_workspace(MsmContext* ctx, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    return &ctx->workspace_for(stream) != &ctx->ws;
}

// 48-byte compressed form of `count` XYZZ points (device pointers), batched inversion per 64 points
void g1_compress_xyzz(void* d_out48, const void* d_xyzz, size_t count, hipStream_t stream) {
    if (!count) return;
    hipLaunchKernelGGL(k_final, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, stream, (const Xyzz*)nullptr,
                       (const Xyzz*)d_xyzz, d_out48, count, 1, 1, 1, OUT_COMPRESSED);
    HIP_TRY(hipGetLastError());
}

// enqueue nbatch MSMs over the first npoints bases; d_scalars / d_out device pointers
// reserve_only: size and allocate the workspace `stream` will use for this shape, launch nothing (hipMalloc
// synchronises the device, so callers that must not stall a running pipeline reserve during set-up)
void msm_enqueue(MsmContext* ctx, void* d_out, const void* d_scalars, size_t npoints, size_t nbatch, int mont,
                 hipStream_t stream, int out_mode, bool reserve_only, size_t nseg) {
    if (npoints * (nseg ? nseg : 1) > ctx->n) throw HipErr{hipErrorInvalidValue, "npoints exceeds the prepared size"};
    if ((nseg || out_mode == OUT_XYZZ) && !ctx->fbw)
        throw HipErr{hipErrorInvalidValue, "segmented / XYZZ-output MSMs need the wide-table path"};
    if (nbatch == 0) return;
    const int c = ctx->c;
    const int nwin = ctx->nwin;
    const size_t nb = ctx->nb;
    const size_t nsets = ctx->prepared ? nbatch : nbatch * (size_t)nwin;
    const size_t set_cap = ctx->prepared ? npoints * (size_t)nwin : (ctx->glv ? 2 * npoints : npoints);
    if (npoints == 0) {
        if (reserve_only) return;
        if (out_mode == OUT_COMPRESSED) throw HipErr{hipErrorInvalidValue, "empty MSM in compressed mode"};
        HIP_TRY(hipMemsetAsync(d_out, 0, nbatch * 144, stream));
        return;
    }
    if (set_cap >= ((size_t)1 << 31) || nsets * nb >= ((size_t)1 << 40)) throw HipErr{hipErrorInvalidValue, "MSM too large"};
    Workspace& ws = ctx->workspace_for(stream);
    // Workspaces are per stream, but streams can outnumber them (and the handle's own stream shares the first one):
    // whoever gets a workspace last used on another stream first waits for that use to finish on the GPU.
    // a workspace that only one stream ever uses (every entry of ws_extra) is ordered by that stream: no events —
    // two runtime calls less per enqueue, and such an enqueue can be captured into a graph
    struct WsUse {
        Workspace& w;
        hipStream_t st;
        bool shared;
        WsUse(Workspace& w_, hipStream_t st_, bool shared_) : w(w_), st(st_), shared(shared_) {
            if (shared && w.last_done && w.last_stream != st) (void)hipStreamWaitEvent(st, w.last_done, 0);
        }
        ~WsUse() {
            if (!shared) return;
            if (!w.last_done && hipEventCreateWithFlags(&w.last_done, hipEventDisableTiming) != hipSuccess) w.last_done = nullptr;
            if (w.last_done) (void)hipEventRecord(w.last_done, st);
            w.last_stream = st;
        }
    };
    if (ctx->fbw) {
        // wide-table path: gather + add, then one block-sum per MSM
        // scalars per lane: 4 for large batches — a quarter of the partial sums for k_blocksum to fold against one
        // more real addition per lane (the first addition into an empty accumulator is free): +1.0 % at 1024 blobs
        // (87.1 k -> 88.0 k commitments/s, three alternating runs on one box); 1 for small batches, where the lanes
        // are needed for latency
        int spl = nbatch >= 256 ? 4 : 1;
        // segment MSMs (FK20: 128 MSMs of 64 scalars per blob): from 128 blobs on, 16 scalars per lane — 320 additions
        // per lane instead of 80 and four partial sums per MSM instead of 16 for k_lane_sum (0.23 -> 0.05 ms at 256
        // blobs); the accumulation itself takes the same 6.2 - 6.4 ms either way (measured, profiles/NOTES.md §16)
        if (nseg && ctx->fbw_glv && npoints % 32 == 0) {
            if (nbatch * npoints >= ((size_t)1 << 21)) spl = 16;       // 256 blobs: 131 072 lanes, two waves per SIMD
            else if (nbatch * npoints >= ((size_t)1 << 20)) spl = 8;  // 128 blobs: the same
        }
        if (ctx->tune.spl) spl = ctx->tune.spl;
        if (spl == 3 || (spl > 4 && spl != 8 && spl != 16)) spl = 4;
        if (!ctx->fbw_glv && spl > 4) spl = 4;
        if (ctx->fbw_glv) spl *= 2;  // two lanes (k1 / k2 digits) per scalar group: the same number of partial sums
        // a few MSMs over the 4096-point setup: a lane per (scalar, half) — 8 additions per lane instead of 16, twice the
        // partial sums for the fold (limb-parallel up to WIDE_FOLD_MAX MSMs, k_blocksum above)
        // (up to 8 MSMs: 8 commitments 0.79 -> 0.71 ms; 16 the same either way, 32 and 64 slower)
        const size_t spl1_max = ctx->tune.spl1_max > 0 ? (size_t)ctx->tune.spl1_max : 8;
        if (ctx->fbw_glv && nbatch <= spl1_max && npoints == 4096 && !ctx->tune.spl && !ctx->tune.no_wide_tail) spl = 1;
        const size_t lanes = (npoints + spl - 1) / spl * (ctx->fbw_glv ? 2 : 1);
        ws.buckets.ensure(nbatch * lanes);
        ws.lvlM[0].ensure(nbatch);
        if (nbatch <= 16 && lanes % 16 == 0 && lanes >= 1024) ws.lvlA[0].ensure(nbatch * 128);
        const size_t hybrid_max = ctx->tune.hybrid_max > 0 ? (size_t)ctx->tune.hybrid_max : BSH_MAX;
        const bool hybrid_fold = nbatch <= hybrid_max && nbatch <= BSH_CAP && lanes % (16 * 256) == 0 &&
                                 !ctx->tune.no_wide_tail && !ctx->tune.no_hybrid_fold;
        bool bcount_new = false;
        if (hybrid_fold) {
            ws.bpart.ensure(BSH_CAP * (size_t)BSH_PARTS);
            if (ws.bcount.cap < BSH_CAP) {
                ws.bcount.ensure(BSH_CAP);
                bcount_new = true;
            }
        }
        const size_t wf_max = ctx->tune.wide_fold_max > 0 ? (size_t)ctx->tune.wide_fold_max : WIDE_FOLD_MAX;
        const bool wide_fold = nbatch <= wf_max && (lanes == 4096 || lanes == 8192) && !ctx->tune.no_wide_tail;
        bool wcount_new = false;
        if (wide_fold) {
            // k_wide_tree: lanes / 32 workgroup sums + lanes / 512 group sums per MSM, a counter per group and one per MSM
            // (zeroed when allocated, left at zero by every launch); k_wide_fold64 (tuning key no_wide_tree): 129 cells x
            // WFOLD parts, a counter per cell — which is more of both
            const size_t need_p = (nbatch * 128 + nbatch) * (size_t)WFOLD, need_c = nbatch * 128 + nbatch;
            ws.wpart.ensure(need_p);
            if (ws.wcount.cap < need_c) {
                ws.wcount.ensure(need_c);
                wcount_new = true;
            }
        }
        if (ctx->fbw_glv) ws.digits.ensure(nbatch * npoints * 2 * (size_t)((nwin + 3) & ~3));
        if (bcount_new) HIP_TRY(hipMemsetAsync(ws.bcount.p, 0, BSH_CAP * sizeof(u32), stream));
        if (wcount_new) HIP_TRY(hipMemsetAsync(ws.wcount.p, 0, ws.wcount.cap * sizeof(u32), stream));
        if (reserve_only) return;
        WsUse ws_use(ws, stream, &ws == &ctx->ws);
        DigitParams P{npoints, nbatch, c, nwin, 1, mont, nb, ctx->n, 0, 0, nwin, (u32)nseg};
        hipEvent_t* pev = nullptr;
        if (ctx->profile && ctx->ev_used + 4 <= MsmContext::EV_MAX) {
            while (ctx->ev.size() < ctx->ev_used + 4) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                ctx->ev.push_back(e);
            }
            pev = &ctx->ev[ctx->ev_used];
            HIP_TRY(hipEventRecord(pev[0], stream));
        }
        const u32* acc_in = (const u32*)d_scalars;
        if (ctx->fbw_glv) {
            hipLaunchKernelGGL(k_fbw_digits, dim3((unsigned)((npoints * nbatch + 255) / 256)), dim3(256), 0, stream, P,
                               (const u32*)d_scalars, ws.digits.p);
            acc_in = ws.digits.p;
        }
        if (pev) HIP_TRY(hipEventRecord(pev[1], stream));
        const dim3 grid((unsigned)((lanes * nbatch + 255) / 256));
#define KZG_FBW_LAUNCH(SPL_, GLV_)                                                                           \
    hipLaunchKernelGGL((k_fbw_accum<SPL_, GLV_>), grid, dim3(256), 0, stream, P, acc_in,                         \
                       (const WidePt*)ctx->wide.p, ws.buckets.p, lanes)
        if (ctx->fbw_glv && spl == 1 && nbatch <= (size_t)ctx->tune.quad_accum_max) {
            // a lane per (scalar, half) AND four lanes per chain: the few-commitments form
            hipLaunchKernelGGL(k_fbw_accum_quad, dim3((unsigned)((lanes * nbatch * 4 + 255) / 256)), dim3(256), 0, stream, P, acc_in,
                               (const WidePt*)ctx->wide.p, ws.buckets.p, lanes);
        } else if (ctx->fbw_glv) {
            if (spl == 1) KZG_FBW_LAUNCH(1, true);
            else if (spl == 2) KZG_FBW_LAUNCH(2, true);
            else if (spl == 4) KZG_FBW_LAUNCH(4, true);
            else if (spl == 8) KZG_FBW_LAUNCH(8, true);
            else if (spl == 16) KZG_FBW_LAUNCH(16, true);
            else KZG_FBW_LAUNCH(32, true);
        } else {
            if (spl == 1) KZG_FBW_LAUNCH(1, false);
            else if (spl == 2) KZG_FBW_LAUNCH(2, false);
            else KZG_FBW_LAUNCH(4, false);
        }
#undef KZG_FBW_LAUNCH
        if (pev) HIP_TRY(hipEventRecord(pev[2], stream));
        Xyzz* sums = out_mode == OUT_XYZZ ? (Xyzz*)d_out : ws.lvlM[0].p;
        if (lanes <= 64) {
            // many small MSMs (segments of a table): one lane adds the few partial sums of an MSM
            hipLaunchKernelGGL(k_lane_sum, dim3((unsigned)((nbatch + 63) / 64)), dim3(64), 0, stream, (const Xyzz*)ws.buckets.p,
                               sums, lanes, nbatch);
        } else if (wide_fold && !ctx->tune.no_wide_tree) {
            // a few commitments (wide_fold_max): one launch of limb-parallel additions, 32 partial sums per workgroup
            const u32 nwg = (u32)(lanes / WTREE_PER_WG);
            hipLaunchKernelGGL(k_wide_tree, dim3((unsigned)(nbatch * nwg)), dim3(256), 0, stream, (const Xyzz*)ws.buckets.p, sums,
                               ws.wpart.p, ws.wcount.p, lanes, nwg);
        } else if (wide_fold) {
            // one or two commitments: the partial sums folded 64 : 1, then 64 : 1 or 128 : 1, with limb-parallel additions
            const int pw2 = (int)(lanes / 64 / WFOLD);  // 8 or 16 points per wave in the second launch
            const size_t c1 = nbatch * (lanes / 64), c2 = nbatch;
            HIP_TRY(hipMemsetAsync(ws.wcount.p, 0, (c1 + c2) * sizeof(u32), stream));
            hipLaunchKernelGGL(k_wide_fold64, dim3((unsigned)(c1 * WFOLD)), dim3(64), 0, stream, (const Xyzz*)ws.buckets.p,
                               ws.lvlA[0].p, ws.wpart.p, ws.wcount.p, 8);
            hipLaunchKernelGGL(k_wide_fold64, dim3((unsigned)(c2 * WFOLD)), dim3(64), 0, stream, (const Xyzz*)ws.lvlA[0].p, sums,
                               ws.wpart.p + c1 * WFOLD, ws.wcount.p + c1, pw2);
        } else if (hybrid_fold) {
            hipLaunchKernelGGL(k_blocksum_hybrid, dim3((unsigned)(nbatch * BSH_PARTS)), dim3(256), 256 * sizeof(Xyzz), stream,
                               (const Xyzz*)ws.buckets.p, sums, ws.bpart.p, ws.bcount.p, lanes);
        } else if (nbatch <= 16 && lanes % 16 == 0 && lanes >= 1024) {
            // a few MSMs: the one-workgroup-per-MSM fold is a latency chain (16 strided additions + 8 tree rounds);
            // 16 workgroups per MSM and a second small fold take 9 + 6 rounds instead (single commitment call
            // 0.83 -> 0.71 ms).  Splitting the windows of a scalar over 3 lanes as well was measured: no gain.
            hipLaunchKernelGGL(k_blocksum, dim3((unsigned)(nbatch * 16)), dim3(256), 256 * sizeof(Xyzz), stream,
                               (const Xyzz*)ws.buckets.p, ws.lvlA[0].p, lanes / 16);
            hipLaunchKernelGGL(k_blocksum, dim3((unsigned)nbatch), dim3(64), 64 * sizeof(Xyzz), stream,
                               (const Xyzz*)ws.lvlA[0].p, sums, (size_t)16);
        } else {
            // many MSMs: two waves per MSM instead of four — fewer mostly-idle tree rounds per partial sum
            // (256 / 128 / 64 threads: 87.7 k / 88.9 k / 87.6 k commitments/s at 1024 blobs, same box)
            int bs = 256;
            if (nbatch >= 256) bs = 128;
            if (const int v = ctx->tune.blocksum_threads) bs = v == 64 || v == 128 || v == 256 ? v : bs;
            hipLaunchKernelGGL(k_blocksum, dim3((unsigned)nbatch), dim3(bs), bs * sizeof(Xyzz), stream,
                               (const Xyzz*)ws.buckets.p, sums, lanes);
        }
        if (out_mode != OUT_XYZZ)
            hipLaunchKernelGGL(k_final, dim3((unsigned)((nbatch + 63) / 64)), dim3(64), 0, stream, (const Xyzz*)nullptr,
                               (const Xyzz*)ws.lvlM[0].p, d_out, nbatch, nwin, c, 1, out_mode);
        if (pev) {
            HIP_TRY(hipEventRecord(pev[3], stream));
            ctx->ev_used += 4;
        }
        HIP_TRY(hipGetLastError());
        return;
    }
    // A batch of large MSMs whose sets together have more coarse bins than the two-level sort's LDS counters hold
    // (MAX_BINS) would take the one-level sort — 1.2 ms instead of 0.3 per 2^20-point MSM (measured: 4 x 2^20 in one call
    // 4.2 ms per MSM against 3.4 for one).  Such a batch runs as sub-batches that fit.  Round 6: ANY batch of large MSMs is
    // cut into sub-batches, and only their accumulations stay on the caller's stream; the sort and the reduction of a
    // sub-batch run on a high-priority side stream beside another sub-batch's accumulation (MsmContext::sub_stream;
    // tuning key sub_streams = how many rotate, 0 = everything on the caller's stream and workspace as in round 5).
    if (nbatch > 1 && !ctx->in_sub_batch && !ctx->tune.one_level_sort && npoints >= ((size_t)1 << 15)) {
        const int fb0 = ctx->tune.fine_bits >= FINE_BITS_MIN && ctx->tune.fine_bits <= FINE_BITS_MAX ? ctx->tune.fine_bits : FINE_BITS_MIN;
        const size_t bins_per_msm = (nb >> fb0) * (ctx->prepared ? (size_t)1 : (size_t)nwin);
        size_t per = bins_per_msm ? MAX_BINS / bins_per_msm : 0;
        // Measured (tools/ab_batched.py, profiles/NOTES.md §20): 4 or 8 x 2^16 in one call 0.70 -> 0.645 ms per MSM with six side
        // streams (three: 0.70), 4 x 2^17 0.87 -> 0.82 — there the reduction chains of the sub-batches (0.7 ms each, a
        // fraction of the chip) overlap each other.  From
        // 2^18 points on an accumulation fills every register of every SIMD for milliseconds and whatever is launched
        // beside it, at any stream priority, crawls (k_tile_sums_loop 2.9 ms instead of 0.32, k_digit_sums_wide 1.6 instead
        // of 0.09) and delays the next sub-batch: 4 x 2^20 3.26 ms per MSM on one stream, 3.3 - 3.5 with side streams.
        // So the side streams serve batches of MSMs below 2^18 points; tuning key sub_large = 1 lifts the limit.
        int nside = ctx->tune.sub_streams < 0 ? 0 : ctx->tune.sub_streams > MsmContext::MAXSUB ? MsmContext::MAXSUB : ctx->tune.sub_streams;
        // Sort ahead (round 6, second form; key sort_ahead): from 2^18 points on NOTHING runs beside an accumulation, but
        // the sort of sub-batch k + 1 — scattered 4-byte traffic and LDS histograms, 40-odd registers — runs on a side
        // stream beside the REDUCTION of sub-batch k, whose chains leave most of the chip idle:
        //     caller's stream:  accumulate(0)  reduce(0)  accumulate(1)  reduce(1) ...
        //     side streams:     sort(0)        sort(1) [after accumulate(0)]       sort(2) [after accumulate(1)] ...
        // Two side streams / workspaces alternate (sort(k + 2) starts after accumulate(k + 1), i.e. after reduce(k) is done
        // with the workspace they share).
        // Measured (tools/ab_batched.py, one box, ms per MSM; off / on with high-priority side streams / on with default
        // priority): 4 x 2^20 3.296 / 3.312 / 3.281, 8 x 2^20 3.318 / 3.376 / 3.283, 8 x 2^18 1.194 / 1.283 / 1.179.  A first
        // form that let the sort start right after the accumulation — beside the tile sums — lost 2.5 % (3.42 vs 3.34): the
        // tile sums of two MSMs are 512 workgroups on every register of the chip, and k_part_scan (ONE workgroup) waited
        // 250 us for a slot.  +-1 % is not a reason for two more streams in a call: off by default.
        const bool sort_ahead = npoints >= ((size_t)1 << 18) && !ctx->tune.sub_large && ctx->tune.sort_ahead && nside >= 2;
        if (npoints >= ((size_t)1 << 18) && !ctx->tune.sub_large) nside = sort_ahead ? 2 : 0;
        if (nside > 0 && per >= 1 && !sort_ahead) {
            // at least two sub-batches; from 2^18 points an MSM's accumulation alone fills the chip for a millisecond
            const size_t cap = npoints >= ((size_t)1 << 18) ? (size_t)1 : (nbatch + 1) / 2;
            if (per > cap) per = cap;
        }
        if (nb >= ((size_t)1 << fb0) && per >= 1 && per < nbatch) {
            const size_t out_stride = out_mode == OUT_COMPRESSED ? 48 : out_mode == OUT_WINDOWS ? (size_t)nwin * 144 : 144;
            const size_t nsub = (nbatch + per - 1) / per;
            const int lanes = nside > 0 ? (int)(nsub < (size_t)nside ? nsub : (size_t)nside) : 0;  // side streams in use
            if (lanes > 0 && !ctx->sub_stream[lanes - 1]) {
                int least = 0, greatest = 0;
                HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
                const int prio = ctx->tune.sub_prio ? greatest : 0;
                for (int k = 0; k < lanes; ++k) {
                    if (!ctx->sub_stream[k]) HIP_TRY(hipStreamCreateWithPriority(&ctx->sub_stream[k], hipStreamNonBlocking, prio));
                    if (!ctx->ev_sub_join[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_sub_join[k], hipEventDisableTiming));
                    if (!ctx->ev_sorted[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_sorted[k], hipEventDisableTiming));
                    if (!ctx->ev_accd[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_accd[k], hipEventDisableTiming));
                    if (!ctx->ev_gate[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_gate[k], hipEventDisableTiming));
                }
            }
            struct SubReset {  // an exception below must not leave the next enqueue in sub-batch mode
                MsmContext* c;
                ~SubReset() {
                    c->accum_redirect = false;
                    c->tail_on_accum = false;
                    c->accum_on = nullptr;
                    c->accum_ready = c->accum_done = c->sort_gate = nullptr;
                    c->in_sub_batch = false;
                }
            } sub_reset{ctx};
            ctx->in_sub_batch = true;
            if (reserve_only) {  // the first sub-batch is the largest; every stream of the rotation gets its workspace
                if (lanes == 0) msm_enqueue(ctx, nullptr, nullptr, npoints, per, mont, stream, out_mode, true, nseg);
                for (int k = 0; k < lanes; ++k)
                    msm_enqueue(ctx, nullptr, nullptr, npoints, per, mont, ctx->sub_stream[k], out_mode, true, nseg);
                return;
            }
            if (lanes > 0) {
                if (!ctx->ev_sub_fork) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_sub_fork, hipEventDisableTiming));
                // whatever produced the scalars (and last used the outputs) on the caller's stream comes first
                HIP_TRY(hipEventRecord(ctx->ev_sub_fork, stream));
                for (int k = 0; k < lanes; ++k) HIP_TRY(hipStreamWaitEvent(ctx->sub_stream[k], ctx->ev_sub_fork, 0));
            }
            size_t k = 0;
            for (size_t b0 = 0; b0 < nbatch; b0 += per, ++k) {
                const size_t nbp = b0 + per <= nbatch ? per : nbatch - b0;
                hipStream_t st = stream;
                if (lanes > 0) {
                    const int lane = (int)(k % (size_t)lanes);
                    st = ctx->sub_stream[lane];
                    ctx->accum_redirect = true;
                    ctx->tail_on_accum = sort_ahead;
                    ctx->accum_on = stream;
                    ctx->accum_ready = ctx->ev_sorted[lane];
                    ctx->accum_done = ctx->ev_accd[lane];
                    // sort ahead: this sub-batch's sort waits for the previous sub-batch's accumulation (recorded on the
                    // caller's stream by that enqueue) and for its tile sums — throughput work on every register of the chip, like the
                    // accumulation — so that it runs beside the limb-parallel chains of the reduction only
                    ctx->sort_gate = sort_ahead ? ctx->ev_gate[lane] : nullptr;
                    if (sort_ahead && k > 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_gate[(int)((k - 1) % (size_t)lanes)], 0));
                }
                msm_enqueue(ctx, d_out ? (unsigned char*)d_out + b0 * out_stride : nullptr,
                            d_scalars ? (const unsigned char*)d_scalars + b0 * npoints * 32 : nullptr, npoints, nbp, mont, st,
                            out_mode, false, nseg);
            }
            ctx->accum_redirect = false;
            ctx->tail_on_accum = false;
            for (int j = 0; j < lanes; ++j) {
                HIP_TRY(hipEventRecord(ctx->ev_sub_join[j], ctx->sub_stream[j]));
                HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_sub_join[j], 0));
            }
            return;
        }
    }
    ws.offsets.ensure(nsets * (nb + 1));
    ws.sorted.ensure(nsets * set_cap);
    // entries per accumulation lane, 2^lgc: longer chunks leave fewer pieces per bucket to fold, shorter ones more lanes
    // (A/B in one process with the tiled reduction, median ms, lgc -> time:  2^18: 5 -> 1.36, 6 -> 1.40;  2^19: 5 -> 2.13,
    //  6 -> 2.10;  2^20: 5 -> 3.70, 6 -> 3.57, 7 -> 3.49;  2^21: 6 -> 6.50, 7 -> 6.35, 8 -> 6.52;  2^22: 6 -> 13.28,
    //  7 -> 12.84, 8 -> 12.69)
    int lgc = npoints >= ((size_t)1 << 22) ? 8 : npoints >= ((size_t)1 << 20) ? 7 : npoints >= ((size_t)1 << 19) ? 6
              : npoints >= ((size_t)1 << 18) ? 5 : 4;
    // the device may pick chunks up to 4x smaller for sets with fewer entries (eff_lgc): the piece slots are sized for that
    int lgc_lo = lgc > 6 ? lgc - 2 : (lgc > 4 ? 4 : lgc);
    if (const int v = ctx->tune.lgc) {
        if (v >= 2 && v <= 8) lgc = lgc_lo = v;
    }
    const size_t nchunk = (set_cap + ((size_t)1 << lgc_lo) - 1) >> lgc_lo;
    ws.buckets.ensure(nsets * (nb + nchunk));
    const size_t n1 = (nb + 1) / 2, n2 = (n1 + GRP - 1) / GRP;  // level 0 folds at least 2, later levels GRP
    ws.lvlA[0].ensure(nsets * n1);
    ws.lvlM[0].ensure(nsets * n1);
    ws.lvlA[1].ensure(nsets * n2);
    ws.lvlM[1].ensure(nsets * n2);
    ws.heavy.ensure(nsets * nb);
    // Window groups (tuning key groups = 2..4, off by default).  The sort is bound by atomics / scattered writes, the
    // accumulation by the integer VALUs and the reduction tail by latency, so a single large MSM can be cut into
    // groups of windows that run as a pipeline on their own streams: the sort of group g+1 overlaps the accumulation
    // of group g, whose tail overlaps the accumulation of g+1; only the Horner over the window sums joins them.
    // With the one-level sort (1.2 ms at n = 2^20) two groups gained 6 % (5.83 -> 5.47 ms; four were slower: every
    // group pays the scalar load + split again); with the two-level sort (0.5 ms) there is nothing left to hide:
    // 6.17 vs 6.22 ms on the same box.
    int G = 1;
    if (!ctx->prepared && nbatch == 1 && !ctx->profile && npoints >= 4096 && nwin >= 4) {
        if (const int v = ctx->tune.groups) {
            if (v >= 1 && v <= MsmContext::MAXG && v <= nwin) G = v;
        }
    }
    const size_t sets_per_group = (nsets + G - 1) / G;
    // lanes the accumulation wants per set: 90 % of two waves per SIMD over the sets of a launch
    const ChunkSel csel{lgc_lo, lgc, (u32)(118000 / sets_per_group)};
    // two-level sort: coarse bins must fit the LDS counters and a point index 24 bits
    const size_t max_pidx = ctx->prepared ? (size_t)ctx->rows * ctx->n : (ctx->glv ? 2 * ctx->n : ctx->n);
    int pbits = 1;  // bits of a point index
    while (((size_t)1 << pbits) < max_pidx) ++pbits;
    int fb = FINE_BITS_MIN;
    if (const int v = ctx->tune.fine_bits) {
        if (v >= FINE_BITS_MIN && v <= FINE_BITS_MAX && v <= 31 - pbits) fb = v;
    }
    const bool two_level = !ctx->tune.one_level_sort && nb >= ((size_t)1 << fb) && fb + 1 + pbits <= 32 &&
                           (nb >> fb) * sets_per_group <= MAX_BINS &&
                           npoints * nbatch >= ((size_t)1 << 15);  // below: four more launches than they save
    if (two_level) {
        ws.tmp.ensure(nsets * set_cap);
        ws.bins.ensure((size_t)G * (3 * MAX_BINS + 8));
        if (G == 1)  // per-workgroup histograms and run starts of the partition pass (1024 scalars per workgroup)
            ws.wghist.ensure(2 * ((npoints * nbatch + 1023) / 1024) * ((nb >> fb) * sets_per_group));
    } else {
        ws.counts.ensure(nsets * nb);
        ws.ranks.ensure((size_t)(ctx->glv ? 2 : 1) * nwin * nbatch * npoints);
    }
    const size_t heavy_cap = sets_per_group * set_cap / HEAVY + 1;  // a heavy bucket holds > HEAVY entries
    ws.heavy_list.ensure(2 * heavy_cap * G);
    ws.nheavy.ensure(G);
    const bool use_top = nsets <= 64;  // many independent sets (batched MSMs) keep plain tree levels busy on their own
    // few chains: run the serial tails limb-parallel, one point operation per wave
    const bool wide_tail = use_top && !ctx->tune.no_wide_tail;
    // few sets of many buckets: the digit-decomposed reduction instead of the (A, M) tree (tuning key tree_tail=1: the tree)
    // measured (same box, tree vs digits): n = 2^14 (4096 buckets) 1.20 vs 1.35 ms, 2^16 1.53 vs 1.48, 2^20 4.53 vs 4.35, 2^22 14.67 vs 14.26
    const bool digit_tail = use_top && nb >= ((size_t)1 << ctx->tune.digit_min_log) && !ctx->tune.tree_tail;
    // the tiled form of the digit sums (k_tile_sums_loop); tuning key flat_digits: one pass over the buckets per digit
    // tile rows: 32 (tiles of 1024 buckets, a CU per workgroup); 16 (512 buckets, a wave per SIMD) by tuning key
    const int tile_rows = ctx->tune.tile_rows ? ctx->tune.tile_rows : 32;  // 16: measured slower alone (3.52 vs 3.42 ms at 2^20) and no help beside an accumulation
    const bool tiled_digits = digit_tail && nb % ((size_t)tile_rows * 32) == 0 && !ctx->tune.flat_digits;
    const size_t ntiles = nb / ((size_t)tile_rows * 32);
    // shape of the tree (the same for every group: level 0 folds by the group size): k_top stride B + 2
    size_t top_stride = 0;
    {
        size_t nin = nb;
        int lvl = 0;
        do {
            int grp = GRP;
            if (lvl == 0)
                while (grp > 2 && sets_per_group * ((nin + grp - 1) / grp) < ((size_t)1 << 17)) grp >>= 1;
            nin = (nin + grp - 1) / grp;
            ++lvl;
        } while (nin > (use_top ? TOP_MAX : (size_t)1));
        int B = 0;
        while (((size_t)1 << B) < nin) ++B;
        top_stride = (size_t)B + 2;
        if (digit_tail) {
            int logNb = 0;
            while (((size_t)1 << logNb) < nb) ++logNb;
            top_stride = (size_t)logNb + 2;
            ws.lvlA[0].ensure(nsets * (size_t)(((logNb + DIGIT_BITS - 1) / DIGIT_BITS) * 32));
            ws.dense.ensure(nsets * nb);
            if (tiled_digits) {
                ws.lvlA[1].ensure(nsets * (nb >> 5));  // group sums
                ws.lvlM[1].ensure(nsets * (nb >> 4));  // per-tile column sums: ntiles x 32 (16-row tiles: nb / 16)
                const size_t cells = nsets * (size_t)(((logNb + DIGIT_BITS - 1) / DIGIT_BITS) * 32 + logNb + 2);
                ws.wpart.ensure(cells * WSPLIT);
                ws.wcount.ensure(cells);
            }
        }
        if (use_top) {
            ws.top.ensure(nsets * top_stride);
            ws.win.ensure(nsets);
        }
    }
    if (reserve_only) return;
    // sub-batch of a batch of large MSMs: the accumulation goes to the caller's stream (see MsmContext::sub_stream)
    const bool accum_redirect = ctx->accum_redirect;
    const hipEvent_t sort_gate = ctx->sort_gate;
    bool gate_recorded = false;
    const bool tail_on_accum = accum_redirect && ctx->tail_on_accum;  // sort-ahead form: the reduction follows its accumulation on the caller's stream
    const hipStream_t accum_on = ctx->accum_on;
    const hipEvent_t accum_ready = ctx->accum_ready, accum_done = ctx->accum_done;
    WsUse ws_use(ws, stream, &ws == &ctx->ws);
    hipEvent_t* pev = nullptr;
    if (ctx->profile && ctx->ev_used + 4 <= MsmContext::EV_MAX) {
        while (ctx->ev.size() < ctx->ev_used + 4) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            ctx->ev.push_back(e);
        }
        pev = &ctx->ev[ctx->ev_used];
        HIP_TRY(hipEventRecord(pev[0], stream));
    }
    if (G > 1) {
        if (!ctx->ev_fork) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        for (int g = 0; g < G; ++g) {
            if (!ctx->aux[g]) HIP_TRY(hipStreamCreateWithFlags(&ctx->aux[g], hipStreamNonBlocking));
            if (!ctx->ev_dig[g]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_dig[g], hipEventDisableTiming));
            if (!ctx->ev_acc[g]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_acc[g], hipEventDisableTiming));
            if (!ctx->ev_done[g]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_done[g], hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(ctx->ev_fork, stream));
    }
    const Xyzz *finA = nullptr, *finM = nullptr;
    for (int g = 0; g < G; ++g) {
        const size_t set0 = (size_t)g * sets_per_group;
        if (set0 >= nsets) break;
        const size_t ns = set0 + sets_per_group <= nsets ? sets_per_group : nsets - set0;
        hipStream_t st = G > 1 ? ctx->aux[g] : stream;
        if (G > 1) {
            HIP_TRY(hipStreamWaitEvent(st, ctx->ev_fork, 0));
            if (g > 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_dig[g - 1], 0));
        }
        // with groups, sets are windows (nbatch == 1): group g emits windows [set0, set0 + ns)
        DigitParams P{npoints, nbatch, c, nwin, ctx->prepared ? 1 : 0, mont, nb, ctx->n, ctx->glv ? 1 : 0,
                      G > 1 ? (int)set0 : 0, G > 1 ? (int)(set0 + ns) : nwin, 0u};
        u32* counts = two_level ? nullptr : ws.counts.p + set0 * nb;
        u32* offsets = ws.offsets.p + set0 * (nb + 1);
        unsigned char* heavy = ws.heavy.p + set0 * nb;
        u32* heavy_list = ws.heavy_list.p + 2 * heavy_cap * g;
        u32* nheavy = ws.nheavy.p + g;
        Xyzz* buckets = ws.buckets.p + set0 * (nb + nchunk);
        if (two_level) {
            const u32 cb = (u32)(nb >> fb), nbins = (u32)ns * cb;
            u32* bin_count = ws.bins.p + (size_t)g * (3 * MAX_BINS + 8);
            u32* bin_start = bin_count + MAX_BINS;
            u32* bin_cursor = bin_start + MAX_BINS + 8;
            u32* tmp = ws.tmp.p + set0 * set_cap;
            const unsigned gpart = (unsigned)((npoints * nbatch + 256 * PART_SCALARS - 1) / (256 * PART_SCALARS));
            HIP_TRY(hipMemsetAsync(bin_count, 0, nbins * sizeof(u32), st));
            // entries per scalar (windows of this launch's group x halves): the staged scatter holds 16 per scalar
            const size_t per_scalar = (size_t)(G > 1 ? ns : (size_t)nwin) * (ctx->glv ? 2 : 1);
            const bool staged = per_scalar * STAGE_T * STAGE_SCALARS <= STAGE_CAP && !ctx->tune.direct_scatter;
            // per-workgroup histograms kept by the count pass (tuning key scatter_atomics=1: recount + one atomic per run)
            const bool keep_hist = staged && G == 1 && !ctx->tune.scatter_atomics;
            u32 *wg_hist = nullptr, *wg_off = nullptr;
            if (keep_hist) {
                ws.wghist.ensure(2 * (size_t)gpart * nbins);
                wg_hist = ws.wghist.p;
                wg_off = wg_hist + (size_t)gpart * nbins;
            }
            hipLaunchKernelGGL(k_part_count, dim3(gpart), dim3(256), nbins * sizeof(u32), st, P, (const u32*)d_scalars,
                               (const AffPt*)ctx->table.p, bin_count, (u32)set0, cb, nbins, fb, wg_hist);
            hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(1024), 0, st, (const u32*)bin_count, bin_start, bin_cursor, nbins);
            if (staged) {
                const unsigned gst = (unsigned)((npoints * nbatch + STAGE_T * STAGE_SCALARS - 1) / (STAGE_T * STAGE_SCALARS));
                const size_t lds = (3 * (size_t)nbins + STAGE_CAP) * sizeof(u32) + STAGE_CAP * sizeof(unsigned short);
                if (keep_hist)
                    hipLaunchKernelGGL(k_part_offsets, dim3((nbins + 63) / 64), dim3(64 * OFF_SEGS), 0, st, (const u32*)wg_hist,
                                       wg_off, (const u32*)bin_start, gpart, nbins);
                hipLaunchKernelGGL(k_part_scatter_staged, dim3(gst), dim3(STAGE_T), lds, st, P, (const u32*)d_scalars,
                                   (const AffPt*)ctx->table.p, (const u32*)bin_start, bin_cursor, tmp, (u32)set0, cb, nbins, fb,
                                   (const u32*)wg_hist, (const u32*)wg_off);
            } else {
                hipLaunchKernelGGL(k_part_scatter, dim3(gpart), dim3(256), 2 * nbins * sizeof(u32), st, P,
                                   (const u32*)d_scalars, (const AffPt*)ctx->table.p, (const u32*)bin_start, bin_cursor, tmp,
                                   (u32)set0, cb, nbins, fb);
            }
            HIP_TRY(hipMemsetAsync(nheavy, 0, sizeof(u32), st));
            hipLaunchKernelGGL(k_bin_sort, dim3(nbins), dim3(256), 0, st, (const u32*)tmp, (const u32*)bin_start, offsets,
                               ws.sorted.p + set0 * set_cap, heavy, heavy_list, nheavy, (u32)heavy_cap, cb, nb, set_cap, fb);
        } else {
        HIP_TRY(hipMemsetAsync(counts, 0, ns * nb * sizeof(u32), st));
        const unsigned gdig = (unsigned)((npoints * nbatch + 255) / 256);
        // k_digits indexes sets globally (b * nwin + w): it gets the unshifted arrays
        hipLaunchKernelGGL(k_digits<0>, dim3(gdig), dim3(256), 0, st, P, (const u32*)d_scalars, ctx->table.p, ws.counts.p,
                           (const u32*)nullptr, (u32*)nullptr, set_cap, ws.ranks.p);
        HIP_TRY(hipMemsetAsync(nheavy, 0, sizeof(u32), st));
        hipLaunchKernelGGL(k_scan, dim3((unsigned)ns), dim3(1024), 0, st, counts, offsets, nb, heavy, heavy_list, nheavy,
                           (u32)heavy_cap);
        hipLaunchKernelGGL(k_digits<1>, dim3(gdig), dim3(256), 0, st, P, (const u32*)d_scalars, ctx->table.p, ws.counts.p,
                           (const u32*)ws.offsets.p, ws.sorted.p, set_cap, ws.ranks.p);
        }
        if (G > 1) {
            HIP_TRY(hipEventRecord(ctx->ev_dig[g], st));
            if (g > 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_acc[g - 1], 0));
        }
        // One PIECE of this group's sets: accumulation on `ast`, everything after it (heavy buckets, bucket reduction,
        // window sums) on `tst`.  With tst != ast the tail of a piece runs beside the accumulation of the next one: the
        // reductions are latency chains on a fraction of the chip (0.7 ms of 3.4 at n = 2^20), the accumulation is VALU
        // throughput — tuning key tail_pieces.
        auto run_piece = [&](size_t ps0, size_t pns, int piece, hipStream_t ast, hipStream_t tst) {
        // lanes per set so that THIS launch's sets fill the chip (every kernel below derives the chunk length from it)
        const ChunkSel pcs{csel.lo, csel.hi, pns == ns ? csel.target : (u32)(118000 / pns)};
        // (offsets, buckets, heavy are this GROUP's arrays: k_heavy's list indexes sets within the group)
        const u32* p_off = offsets + (ps0 - set0) * (nb + 1);
        Xyzz* p_buckets = buckets + (ps0 - set0) * (nb + nchunk);
        const unsigned char* p_heavy = heavy + (ps0 - set0) * nb;
        hipStream_t kst = ast;
        if (accum_redirect) {  // sorted on ast; accumulated on accum_on, behind the previous sub-batch's accumulation
            HIP_TRY(hipEventRecord(accum_ready, ast));
            HIP_TRY(hipStreamWaitEvent(accum_on, accum_ready, 0));
            kst = accum_on;
        }
        hipLaunchKernelGGL(k_accum, dim3((unsigned)((pns * nchunk + 255) / 256)), dim3(256), 0, kst, p_off,
                           (const u32*)(ws.sorted.p + ps0 * set_cap), (const AffPt*)ctx->table.p, p_buckets, nb, pns, set_cap,
                           nchunk, pcs);
        if (accum_redirect) {
            HIP_TRY(hipEventRecord(accum_done, accum_on));
            if (!tail_on_accum) HIP_TRY(hipStreamWaitEvent(ast, accum_done, 0));
        }
        if (tail_on_accum) tst = accum_on;  // (pieces and groups are forms of a lone MSM: never together with sub-batches)
        else if (tst != ast) {
            HIP_TRY(hipEventRecord(ctx->ev_acc[piece], ast));
            HIP_TRY(hipStreamWaitEvent(tst, ctx->ev_acc[piece], 0));
        } else if (G > 1) {
            HIP_TRY(hipEventRecord(ctx->ev_acc[g], ast));  // the next group's accumulation may start
        }
        hipStream_t st = tst;
        hipLaunchKernelGGL(k_heavy, dim3(256, 64), dim3(64), 0, st, buckets, (const u32*)offsets, (const u32*)heavy_list,
                           (const u32*)nheavy, (u32)heavy_cap, nb, nchunk, 1u, HSEG, pcs, (u32)(ps0 - set0), (u32)(ps0 - set0 + pns));
        hipLaunchKernelGGL(k_heavy, dim3(1024, 1), dim3(64), 0, st, buckets, (const u32*)offsets, (const u32*)heavy_list,
                           (const u32*)nheavy, (u32)heavy_cap, nb, nchunk, HSEG, 0u, pcs, (u32)(ps0 - set0), (u32)(ps0 - set0 + pns));
        if (digit_tail) {
            // few sets: digit-decomposed reduction (k_digit_sums / k_digit_bits), then the Horner over the bit sums
            int logNb = 0;
            while (((size_t)1 << logNb) < nb) ++logNb;
            const int J = (logNb + DIGIT_BITS - 1) / DIGIT_BITS;
            Xyzz* S = ws.lvlA[0].p + ps0 * (size_t)(J * 32);
            Xyzz* top = ws.top.p + ps0 * top_stride;
            Xyzz* dense = ws.dense.p + ps0 * nb;
            if (tiled_digits) {
                Xyzz* Gs = ws.lvlA[1].p + ps0 * (nb >> 5);
                Xyzz* Cp = ws.lvlM[1].p + ps0 * (ntiles * 32);
#define KZG_TILE_SUMS(R, Q)                                                                                                          \
    hipLaunchKernelGGL((k_tile_sums_loop<R, Q>), dim3((unsigned)(pns * ntiles)), dim3(R * 16), (R * 16) * sizeof(Xyzz), st, \
                       (const Xyzz*)p_buckets, p_off, p_heavy, dense, Gs, Cp, nb, nchunk, pcs)
                if (tile_rows == 16) {
                    if (ctx->tune.tile_quad) KZG_TILE_SUMS(16, true);
                    else KZG_TILE_SUMS(16, false);
                } else {
                    if (ctx->tune.tile_quad) KZG_TILE_SUMS(32, true);
                    else KZG_TILE_SUMS(32, false);
                }
#undef KZG_TILE_SUMS
                if (tail_on_accum && sort_gate && !gate_recorded) {
                    HIP_TRY(hipEventRecord(sort_gate, st));
                    gate_recorded = true;
                }
                if (wide_tail) {
                    // cells of this group: [0, pns * J * 32) digit sums, then pns * (logNb + 2) bit sums
                    const size_t c1 = pns * (size_t)(J * 32), c2 = pns * (size_t)(logNb + 2);
                    const size_t cbase = ps0 * (size_t)(J * 32 + logNb + 2);
                    u32* cnt = ws.wcount.p + cbase;
                    Xyzz* part = ws.wpart.p + cbase * WSPLIT;  // WSPLIT >= WSPLIT_S slots per cell
                    HIP_TRY(hipMemsetAsync(cnt, 0, (c1 + c2) * sizeof(u32), st));
                    hipLaunchKernelGGL(k_digit_sums_wide, dim3((unsigned)(c1 * WSPLIT_S)), dim3(64), 0, st, (const Xyzz*)Gs,
                                       (const Xyzz*)Cp, S, part, cnt, nb, logNb, J, ntiles);
                    hipLaunchKernelGGL(k_digit_bits_wide, dim3((unsigned)(c2 * WSPLIT)), dim3(64), 0, st, (const Xyzz*)S, top,
                                       part + c1 * WSPLIT, cnt + c1, logNb, J);
                } else {
                    hipLaunchKernelGGL(k_digit_sums2, dim3((unsigned)(pns * (size_t)(J * 32))), dim3(64), 0, st, (const Xyzz*)Gs,
                                       (const Xyzz*)Cp, S, nb, logNb, J, ntiles);
                }
            } else {
                hipLaunchKernelGGL(k_fold_buckets, dim3((unsigned)((pns * nb + 127) / 128)), dim3(128), 0, st,
                                   (const Xyzz*)p_buckets, p_off, p_heavy, dense, nb, pns, nchunk,
                                   pcs);
                hipLaunchKernelGGL(k_digit_sums, dim3((unsigned)(pns * (size_t)(J * 32))), dim3(DIGIT_T), DIGIT_T * sizeof(Xyzz),
                                   st, (const Xyzz*)dense, S, nb, logNb, J);
            }
            if (!(tiled_digits && wide_tail))
                hipLaunchKernelGGL(k_digit_bits, dim3((unsigned)(pns * (size_t)(logNb + 2))), dim3(64), 0, st, (const Xyzz*)S,
                                   top, logNb, J);
            if (wide_tail)
                hipLaunchKernelGGL(k_winsum_wide, dim3((unsigned)pns), dim3(64), 0, st, (const Xyzz*)top, ws.win.p + ps0,
                                   logNb, 0);
            else
                hipLaunchKernelGGL(k_winsum, dim3((unsigned)((pns + 63) / 64)), dim3(64), 0, st, (const Xyzz*)top,
                                   ws.win.p + ps0, pns, logNb, 0);
            finA = nullptr;
            finM = ws.win.p;
            return;
        }
        // bucket-reduction tree: GRP-ary levels while they are throughput work (nb -> nb/GRP -> ...), then the
        // B + 2 concurrent plain sums of k_top and the short per-window Horner of k_winsum
        size_t nin = nb;
        int logS = 0, lvl = 0;
        const Xyzz *inA = p_buckets, *inM = nullptr;
        do {
            // level 0 also folds each bucket's pieces (entries/chunk + 1 of them): keep >= ~128k lanes in flight
            int grp = GRP, lg = 3;
            if (lvl == 0)
                while (grp > 2 && sets_per_group * ((nin + grp - 1) / grp) < ((size_t)1 << 17)) {
                    grp >>= 1;
                    --lg;
                }
            const size_t nout = (nin + grp - 1) / grp;
            Xyzz *oA = ws.lvlA[lvl & 1].p + ps0 * nout, *oM = ws.lvlM[lvl & 1].p + ps0 * nout;
            const unsigned grid = (unsigned)((pns * nout + 127) / 128);
            if (lvl == 0)
                hipLaunchKernelGGL(k_level<true>, dim3(grid), dim3(128), 0, st, inA, inM, oA, oM, nin, pns, logS,
                                   (const u32*)offsets, p_heavy, nb, nchunk, grp, pcs);
            else
                hipLaunchKernelGGL(k_level<false>, dim3(grid), dim3(128), 0, st, inA, inM, oA, oM, nin, pns, logS,
                                   (const u32*)nullptr, (const unsigned char*)nullptr, nb, nchunk, grp, pcs);
            inA = oA;
            inM = oM;
            nin = nout;
            logS += lg;
            ++lvl;
        } while (nin > (use_top ? TOP_MAX : (size_t)1));
        if (use_top) {
            int B = 0;
            while (((size_t)1 << B) < nin) ++B;
            Xyzz* top = ws.top.p + ps0 * top_stride;
            hipLaunchKernelGGL(k_top, dim3((unsigned)(pns * (size_t)(B + 2))), dim3(TOPT), TOPT * sizeof(Xyzz), st, inA, inM,
                               top, nin, B);
            if (wide_tail)
                hipLaunchKernelGGL(k_winsum_wide, dim3((unsigned)pns), dim3(64), 0, st, (const Xyzz*)top, ws.win.p + ps0, B,
                                   logS);
            else
                hipLaunchKernelGGL(k_winsum, dim3((unsigned)((pns + 63) / 64)), dim3(64), 0, st, (const Xyzz*)top,
                                   ws.win.p + ps0, pns, B, logS);
            finA = nullptr;
            finM = ws.win.p;
        } else {
            // plain tree: one (A, M) node per set, in the level buffers written last
            finA = ws.lvlA[(lvl - 1) & 1].p;
            finM = ws.lvlM[(lvl - 1) & 1].p;
        }
            };
        {
            // pieces: only for one group of few large sets (a single large MSM) in the digit-tail form
            int pieces = 1;
            // Measured (tools/time_pieces.py, profiles/NOTES.md): n = 2^20 in 1 / 2 / 3 / 4 pieces 3.50 / 3.54 / 4.00 / 3.92 ms.
            // The side tail does run inside the next accumulation (the trace shows it there, three times slower than
            // alone, the accumulation none the slower), but two accumulation launches with the shorter chunks a piece needs
            // to fill the chip cost 0.25 ms more than one, and the LAST piece's tail — latency chains that do not shrink
            // with the number of sets — stays exposed.  One piece is the default; the key stays for the measurement.
            if (G == 1 && nbatch == 1 && digit_tail && tiled_digits && wide_tail && ns >= 4 && ctx->tune.tail_pieces > 1) {
                pieces = ctx->tune.tail_pieces;
                if (pieces > MsmContext::MAXG) pieces = MsmContext::MAXG;
                if ((size_t)pieces > ns) pieces = (int)ns;
            }
            if (pev) HIP_TRY(hipEventRecord(pev[1], st));
            size_t done = 0;
            for (int p = 0; p < pieces; ++p) {
                const size_t pns = (ns - done) / (size_t)(pieces - p);
                hipStream_t tst = st;
                if (p + 1 < pieces) {  // the last piece's tail stays on the launch stream
                    if (!ctx->aux[p]) HIP_TRY(hipStreamCreateWithFlags(&ctx->aux[p], hipStreamNonBlocking));
                    if (!ctx->ev_acc[p]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_acc[p], hipEventDisableTiming));
                    if (!ctx->ev_done[p]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_done[p], hipEventDisableTiming));
                    tst = ctx->aux[p];
                }
                run_piece(set0 + done, pns, p, st, tst);
                if (tst != st) HIP_TRY(hipEventRecord(ctx->ev_done[p], tst));
                done += pns;
            }
            // the launch stream joins the side tails only now: a wait placed earlier would hold the next piece's accumulation
            for (int p = 0; p + 1 < pieces; ++p) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_done[p], 0));
            if (pev) HIP_TRY(hipEventRecord(pev[2], st));
            if (G > 1) {
                HIP_TRY(hipEventRecord(ctx->ev_done[g], st));
                HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_done[g], 0));
            }
        }
    }
    const hipStream_t fin_stream = tail_on_accum ? accum_on : stream;
    if (tail_on_accum && sort_gate && !gate_recorded) HIP_TRY(hipEventRecord(sort_gate, fin_stream));  // reduction forms without tiles
    if (wide_tail && !ctx->prepared && out_mode == OUT_JACOBIAN && nwin > 1) {
        // few independent Horner chains: one wave each, limb-parallel doublings
        hipLaunchKernelGGL(k_final_wide, dim3((unsigned)nbatch), dim3(64), 0, fin_stream, finM, d_out, nwin, c);
    } else {
        const size_t nfinal = out_mode == OUT_WINDOWS ? nbatch * (size_t)nwin : nbatch;
        hipLaunchKernelGGL(k_final, dim3((unsigned)((nfinal + 63) / 64)), dim3(64), 0, fin_stream, finA, finM, d_out, nbatch, nwin,
                           c, ctx->prepared ? 1 : 0, out_mode);
    }
    if (pev) {
        HIP_TRY(hipEventRecord(pev[3], stream));
        ctx->ev_used += 4;
    }
    HIP_TRY(hipGetLastError());
}

void msm_lock(MsmContext* ctx) { ctx->mu.lock(); }
void msm_unlock(MsmContext* ctx) { ctx->mu.unlock(); }
void msm_set_profile(MsmContext* ctx, bool on) {
    ctx->profile = on;
    ctx->ev_used = 0;
}
// averages over every enqueue recorded since msm_set_profile(true)
int msm_get_profile(MsmContext* ctx, float* accum_ms, float* total_ms) {
    if (!ctx->profile || ctx->ev_used == 0) return 0;
    double a = 0, t = 0;
    int cnt = 0;
    for (size_t k = 0; k + 4 <= ctx->ev_used; k += 4) {
        float fa, ft;
        if (hipEventSynchronize(ctx->ev[k + 3]) != hipSuccess) return 0;
        if (hipEventElapsedTime(&fa, ctx->ev[k + 1], ctx->ev[k + 2]) != hipSuccess) return 0;
        if (hipEventElapsedTime(&ft, ctx->ev[k], ctx->ev[k + 3]) != hipSuccess) return 0;
        a += fa;
        t += ft;
        ++cnt;
    }
    *accum_ms = (float)(a / cnt);
    *total_ms = (float)(t / cnt);
    return cnt;
}

// one batch of the combiner on `lane`: every request has the same length; fills failed / err of each
static void msm_run_combined_batch(MsmContext* ctx, MsmContext::CombineLane& lane, const std::vector<MsmContext::HostCall*>& batch) {
    const size_t nb = batch.size(), np = batch[0]->npoints;
    try {
        DeviceGuard on_device(ctx->device);
        HIP_TRY(on_device.err);
        {
            std::lock_guard<std::mutex> lk(ctx->mu);  // enqueue only: the wait below runs beside the other lane's enqueue
            if (!lane.st) HIP_TRY(hipStreamCreateWithFlags(&lane.st, hipStreamNonBlocking));
            if (!lane.h_out) HIP_TRY(hipHostMalloc((void**)&lane.h_out, MsmContext::COMBINE_MAX * 144, hipHostMallocDefault));
            lane.scalars.ensure(nb * np * 8 + 8);
            lane.out.ensure(nb * 3 + 3);
            try {
                bool all_slots = true;
                for (size_t j = 0; j < nb; ++j) all_slots = all_slots && batch[j]->slot != nullptr;
                if (all_slots) {
                    ScalarSlots ss;
                    for (size_t j = 0; j < MsmContext::COMBINE_MAX; ++j) ss.p[j] = j < nb ? (const uint4*)batch[j]->slot : nullptr;
                    const size_t per = np * 2, total = nb * per;  // 16-byte words per request
                    hipLaunchKernelGGL(k_gather_scalars, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, lane.st,
                                       (uint4*)lane.scalars.p, ss, per, nb);
                } else {  // a caller found the slot pool empty: its scalars are in pageable memory
                    for (size_t j = 0; j < nb; ++j)
                        HIP_TRY(hipMemcpyAsync(lane.scalars.p + j * np * 8, batch[j]->slot ? (const void*)batch[j]->slot : batch[j]->scalars,
                                               np * 32, hipMemcpyHostToDevice, lane.st));
                }
                msm_enqueue(ctx, lane.out.p, lane.scalars.p, np, nb, 1, lane.st, OUT_JACOBIAN);
                HIP_TRY(hipMemcpyAsync(lane.h_out, lane.out.p, nb * 144, hipMemcpyDeviceToHost, lane.st));
            } catch (...) {
                (void)hipStreamSynchronize(lane.st);  // whatever was enqueued still reads the callers' slots
                throw;
            }
        }
        HIP_TRY(hipStreamSynchronize(lane.st));
        for (size_t j = 0; j < nb; ++j) memcpy(batch[j]->out, lane.h_out + j * 144, 144);
    } catch (const HipErr& e) {
        for (auto* r : batch) {
            r->failed = true;
            r->err = e;
        }
    } catch (...) {
        for (auto* r : batch) {
            r->failed = true;
            r->err = HipErr{hipErrorUnknown, "combined MSM batch failed"};
        }
    }
}

static void msm_run_host_combined(MsmContext* ctx, void* out, const void* scalars, size_t npoints) {
    MsmContext::HostCall me{out, scalars, npoints};
    auto& q = ctx->comb;
    std::vector<MsmContext::HostCall*> batch;
    batch.reserve(MsmContext::COMBINE_MAX);  // everything that can throw happens before the request is visible
    std::unique_lock<std::mutex> lk(q.mu);
    if (q.free_slots.empty() && !q.pinned_failed && q.slots_allocated < MsmContext::COMBINE_SLOTS) {
        DeviceGuard on_device(ctx->device);
        q.slot_bytes = (ctx->n < MsmContext::COMBINE_NMAX ? ctx->n : MsmContext::COMBINE_NMAX) * 32;
        int grow = q.slots_allocated < 8 ? 4 : q.slots_allocated < 16 ? 8 : 16;
        if (grow > MsmContext::COMBINE_SLOTS - q.slots_allocated) grow = MsmContext::COMBINE_SLOTS - q.slots_allocated;
        unsigned char* chunk = nullptr;
        if (on_device.err != hipSuccess ||
            hipHostMalloc((void**)&chunk, (size_t)grow * q.slot_bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
            q.pinned_failed = true;  // callers without a slot hand their own (pageable) buffer to the batch
            (void)hipGetLastError();
        } else {
            q.slot_chunks.push_back(chunk);
            q.slots_allocated += grow;
            for (int i = grow; i-- > 0;) q.free_slots.push_back(chunk + (size_t)i * q.slot_bytes);
        }
    }
    if (!q.free_slots.empty()) {
        me.slot = q.free_slots.back();
        q.free_slots.pop_back();
        lk.unlock();
        memcpy(me.slot, scalars, npoints * 32);
        lk.lock();
    }
    q.pending.push_back(&me);
    q.cv.notify_one();  // a leader gathering requests may have enough now
    while (!me.done) {
        if (q.leaders < ctx->tune.combine_lanes && !q.pending.empty()) {
            ++q.leaders;
            MsmContext::CombineLane* lane = nullptr;
            for (auto& l : q.lanes)
                if (!l.busy) {
                    lane = &l;
                    break;
                }
            lane->busy = true;  // leaders <= lanes: one is free
            while (!q.pending.empty() && !me.done) {
                // another batch is in flight: a short wait lets the callers it is about to release come back with their
                // next request — larger batches, fewer invocations
                if (q.leaders > 1 && q.pending.size() < (size_t)ctx->tune.combine_gather_min && ctx->tune.combine_gather_us > 0) {
                    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(ctx->tune.combine_gather_us);
                    while (q.pending.size() < (size_t)ctx->tune.combine_gather_min && !me.done &&
                           q.cv.wait_until(lk, until) != std::cv_status::timeout) {
                    }
                    if (me.done) break;
                    if (q.pending.empty()) continue;
                }
                batch.clear();
                const size_t np = q.pending.front()->npoints;
                for (auto it = q.pending.begin(); it != q.pending.end() && batch.size() < MsmContext::COMBINE_MAX;) {
                    if ((*it)->npoints == np) {
                        batch.push_back(*it);
                        it = q.pending.erase(it);
                    } else {
                        ++it;
                    }
                }
                lk.unlock();
                msm_run_combined_batch(ctx, *lane, batch);
                lk.lock();
                for (auto* r : batch) r->done = true;
                q.cv.notify_all();
            }
            lane->busy = false;
            --q.leaders;
            q.cv.notify_all();  // whoever still waits leads what is left
        } else {
            q.cv.wait(lk);
        }
    }
    if (me.slot) q.free_slots.push_back(me.slot);
    lk.unlock();
    if (me.failed) throw me.err;
}

// host buffers in, host buffers out
void msm_run_host(MsmContext* ctx, void* out, const void* scalars, size_t npoints, size_t nbatch, size_t nseg) {
    if (npoints * (nseg ? nseg : 1) > ctx->n) throw HipErr{hipErrorInvalidValue, "npoints exceeds the prepared size"};
    if (ctx->prepared && nbatch == 1 && npoints > 0 && npoints <= MsmContext::COMBINE_NMAX && ctx->tune.combine && !nseg) {
        msm_run_host_combined(ctx, out, scalars, npoints);
        return;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard on_device(ctx->device);
    HIP_TRY(on_device.err);
    ctx->ws.scalars.ensure(nbatch * npoints * 8 + 8);
    ctx->ws.out.ensure(nbatch * 3 + 3);
    if (npoints * nbatch)
        HIP_TRY(hipMemcpyAsync(ctx->ws.scalars.p, scalars, nbatch * npoints * 32, hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->prepared && npoints > 0) {
        // variable-base engine: the ~255 Horner doublings are one serial chain; a CPU core runs that chain
        // several times faster than a single GPU lane, and the result goes to the host anyway
        const int nwin = ctx->nwin;
        ctx->ws.out.ensure(nbatch * (size_t)nwin * 3);
        msm_enqueue(ctx, ctx->ws.out.p, ctx->ws.scalars.p, npoints, nbatch, 1, ctx->stream, OUT_WINDOWS);
        std::vector<blst_p1> win(nbatch * (size_t)nwin);
        HIP_TRY(hipMemcpyAsync(win.data(), ctx->ws.out.p, win.size() * 144, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (size_t b = 0; b < nbatch; ++b) {
            HostJac acc;
            acc.x = acc.y = acc.z = ff::Fp::zero();
            for (int w = nwin - 1; w >= 0; --w) {
                for (int k = 0; k < ctx->c; ++k) acc = host_jac_dbl(acc);
                const ff::Fp* P = reinterpret_cast<const ff::Fp*>(&win[b * nwin + w]);
                acc = host_jac_add(acc, HostJac{P[0], P[1], P[2]});
            }
            ff::Fp* O = reinterpret_cast<ff::Fp*>((blst_p1*)out + b);
            O[0] = acc.x;
            O[1] = acc.y;
            O[2] = acc.z;
        }
        return;
    }
    msm_enqueue(ctx, ctx->ws.out.p, ctx->ws.scalars.p, npoints, nbatch, 1, ctx->stream, OUT_JACOBIAN, false, nseg);
    HIP_TRY(hipMemcpyAsync(out, ctx->ws.out.p, nbatch * 144, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
}

}  // namespace kzgamd

// ---------------------------------------------------------------- C ABI (B1)
using kzgamd::MsmContext;

template <class F>
static RustError guarded(F&& f) {
    try {
        f();
        return ok_error();
    } catch (const HipErr& e) {
        return make_error((int)e.e ? (int)e.e : 1, std::string(e.what) + ": " + hipGetErrorString(e.e));
    } catch (const std::exception& e) {
        return make_error(1, e.what());
    }
}

// a handle from the C ABI: NULL (with a line on stderr) on any failure, a malformed configuration included
template <class F>
static void* create_guarded(const char* who, const KzgAmdConfig* cfg, F&& make) {
    try {
        kzgamd::Options opt;
        std::string err;
        if (!kzgamd::Options::resolve(opt, cfg, &err)) {
            fprintf(stderr, "kzg_mi355x: %s: %s\n", who, err.c_str());
            return nullptr;
        }
        return make(opt);
    } catch (const HipErr& e) {
        fprintf(stderr, "kzg_mi355x: %s failed: %s: %s\n", who, e.what, hipGetErrorString(e.e));
        return nullptr;
    } catch (...) {
        return nullptr;
    }
}

extern "C" void kzgamd_config_init(KzgAmdConfig* cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = (uint32_t)sizeof *cfg;
    cfg->device = -1;
}

extern "C" const char* kzgamd_tuning_keys(void) {
    static const std::string text = [] {
        std::string s;
        const kzgamd::TuneKey* k = kzgamd::tune_keys();
        for (int i = 0; i < kzgamd::T_COUNT; ++i)
            s += std::string(k[i].name) + " " + std::to_string(k[i].dflt) + " " + std::to_string(k[i].lo) + " " +
                 std::to_string(k[i].hi) + " " + k[i].what + "\n";
        return s;
    }();
    return text.c_str();
}

extern "C" void* kzgamd_prepare_msm_ex(const blst_p1_affine points[], size_t npoints, const KzgAmdConfig* cfg) {
    if (!points || npoints == 0) return nullptr;
    return create_guarded("prepare_msm", cfg, [&](const kzgamd::Options& opt) {
        return kzgamd::msm_create(points, npoints, false, true, false, kzgamd::G1_CHECK, &opt);
    });
}
extern "C" void* prepare_msm(const blst_p1_affine points[], size_t npoints) { return kzgamd_prepare_msm_ex(points, npoints, nullptr); }

// rows base sets of cols points in one wide table; see include/kzg_mi355x.h
extern "C" void* kzgamd_prepare_msm_matrix(const blst_p1_affine points[], size_t rows, size_t cols, const KzgAmdConfig* cfg) {
    if (!points || rows == 0 || cols == 0) return nullptr;
    return create_guarded("kzgamd_prepare_msm_matrix", cfg, [&](const kzgamd::Options& opt) -> void* {
        MsmContext* ctx = kzgamd::msm_create(points, rows * cols, false, true, false, kzgamd::G1_CHECK, &opt);
        if (!ctx->fbw) {
            kzgamd::msm_destroy(ctx);
            fprintf(stderr, "kzg_mi355x: kzgamd_prepare_msm_matrix: no wide table fits the budget (%zu x %zu points)\n", rows, cols);
            return nullptr;
        }
        ctx->mat_rows = rows;
        ctx->mat_cols = cols;
        return ctx;
    });
}

// the same table as a part of an existing handle: what one PrecomputationTable of the reference holds (points AND
// matrix, kzg/src/msm/bgmw.rs:206-304), behind the single pointer SpparkPrecomputation has room for
extern "C" RustError kzgamd_msm_attach_matrix(void* msm, const blst_p1_affine points[], size_t rows, size_t cols,
                                              const KzgAmdConfig* cfg) {
    if (!msm || !points || rows == 0 || cols == 0) return make_error(1, "kzgamd_msm_attach_matrix: bad arguments");
    MsmContext* ctx = (MsmContext*)msm;
    KzgAmdConfig inherited;
    kzgamd_config_init(&inherited);
    if (cfg) inherited = *cfg;
    inherited.device = ctx->device;  // the matrix lives where its handle lives
    {
        // a handle takes ONE matrix, once: matrix calls read ctx->matrix without the handle's lock (they run on the
        // matrix's own), so replacing it under them would free a table in use
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (ctx->matrix) return make_error(1, "kzgamd_msm_attach_matrix: the handle already has a matrix attached");
    }
    void* m = kzgamd_prepare_msm_matrix(points, rows, cols, &inherited);
    if (!m) return make_error(1, "kzgamd_msm_attach_matrix: the matrix handle could not be built (see stderr)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->matrix) {  // another thread attached meanwhile
        free_msm(m);
        return make_error(1, "kzgamd_msm_attach_matrix: the handle already has a matrix attached");
    }
    ctx->matrix = (MsmContext*)m;
    return ok_error();
}

extern "C" int kzgamd_msm_matrix_shape(void* msm, size_t* rows, size_t* cols) {
    MsmContext* ctx = (MsmContext*)msm;
    if (ctx && !ctx->mat_rows && ctx->matrix) ctx = ctx->matrix;
    if (!ctx || !ctx->mat_rows) return 1;
    if (rows) *rows = ctx->mat_rows;
    if (cols) *cols = ctx->mat_cols;
    return 0;
}

extern "C" RustError kzgamd_mult_pippenger_matrix(void* msm, blst_p1 out[], const blst_fr scalars[], size_t nmat) {
    if (!msm || !out || !scalars) return make_error(1, "kzgamd_mult_pippenger_matrix: null handle, output or scalars");
    MsmContext* ctx = (MsmContext*)msm;
    if (!ctx->mat_rows && ctx->matrix) ctx = ctx->matrix;
    if (!ctx->mat_rows) return make_error(1, "kzgamd_mult_pippenger_matrix: not a matrix handle");
    return guarded([&] { kzgamd::msm_run_host(ctx, out, scalars, ctx->mat_cols, nmat * ctx->mat_rows, ctx->mat_rows); });
}

extern "C" void free_msm(void* msm) {
    if (msm) kzgamd::msm_destroy((MsmContext*)msm);
}

extern "C" RustError mult_pippenger_prepared(void* msm, blst_p1* out, size_t npoints, const blst_fr scalars[]) {
    if (!msm || !out) return make_error(1, "mult_pippenger_prepared: null handle or output");
    return guarded([&] { kzgamd::msm_run_host((MsmContext*)msm, out, scalars, npoints, 1); });
}

extern "C" RustError mult_pippenger_prepared_batch(void* msm, blst_p1 out[], size_t npoints, size_t nbatch,
                                                   const blst_fr scalars[]) {
    if (!msm || !out) return make_error(1, "mult_pippenger_prepared_batch: null handle or output");
    return guarded([&] { kzgamd::msm_run_host((MsmContext*)msm, out, scalars, npoints, nbatch); });
}

extern "C" RustError mult_pippenger(blst_p1* out, const blst_p1_affine points[], size_t npoints, const blst_fr scalars[]) {
    if (!out) return make_error(1, "mult_pippenger: null output");
    return guarded([&] {
        if (npoints == 0) {
            memset(out, 0, sizeof *out);
            return;
        }
        // bases outside G1 are legal input here: the membership test where it costs less than the split saves
        // (a 1.8 ms latency chain up to ~2^16 points, 22 ms at 2^20), the unsplit engine beyond
        MsmContext* ctx = kzgamd::msm_create(points, npoints, false, false, false,
                                             npoints <= ((size_t)1 << 15) ? kzgamd::G1_CHECK : kzgamd::G1_NO_SPLIT);
        try {
            kzgamd::msm_run_host(ctx, out, scalars, npoints, 1);
        } catch (...) {
            kzgamd::msm_destroy(ctx);
            throw;
        }
        kzgamd::msm_destroy(ctx);
    });
}

extern "C" RustError kzgamd_msm_prepared_batch_device(void* msm, void* d_out, const void* d_scalars, size_t npoints,
                                                      size_t nbatch, int scalars_mont, void* stream) {
    if (!msm) return make_error(1, "null handle");
    return guarded([&] {
        MsmContext* ctx = (MsmContext*)msm;
        std::lock_guard<std::mutex> lk(ctx->mu);
        kzgamd::DeviceGuard on_device(ctx->device);
        HIP_TRY(on_device.err);
        kzgamd::msm_enqueue(ctx, d_out, d_scalars, npoints, nbatch, scalars_mont, (hipStream_t)stream, kzgamd::OUT_JACOBIAN);
    });
}

// Allocates, now, the workspace `stream` will use for nbatch MSMs of npoints scalars on this handle, so that the
// enqueue calls that follow never call hipMalloc (which synchronises the device).  Without it the first enqueue on
// a new stream, or with a larger shape, allocates lazily.
extern "C" RustError kzgamd_msm_reserve(void* msm, size_t npoints, size_t nbatch, void* stream) {
    if (!msm) return make_error(1, "null handle");
    return guarded([&] {
        MsmContext* ctx = (MsmContext*)msm;
        std::lock_guard<std::mutex> lk(ctx->mu);
        kzgamd::DeviceGuard on_device(ctx->device);
        HIP_TRY(on_device.err);
        kzgamd::msm_enqueue(ctx, nullptr, nullptr, npoints, nbatch, 0, (hipStream_t)stream, kzgamd::OUT_JACOBIAN, true);
    });
}

extern "C" int kzgamd_msm_device(void* msm) { return msm ? ((MsmContext*)msm)->device : -1; }

// Device selection for the calling thread (hipSetDevice / hipGetDevice): handles created afterwards — prepare_msm,
// kzgamd_msm_create_device, kzgamd_ntt_new, load_trusted_setup(_file) — live on that GPU; every later call on a
// handle switches to the handle's GPU by itself and restores the caller's device on return.
extern "C" int kzgamd_set_device(int device) { return hipSetDevice(device) == hipSuccess ? 0 : 1; }
extern "C" int kzgamd_get_device(void) {
    int d = -1;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}

extern "C" int kzgamd_msm_info(void* msm, int* window_bits, int* rows, size_t* nbuckets, size_t* npoints) {
    if (!msm) return 1;
    MsmContext* ctx = (MsmContext*)msm;
    if (window_bits) *window_bits = ctx->c;
    if (rows) *rows = ctx->rows;
    if (nbuckets) *nbuckets = ctx->nb;
    if (npoints) *npoints = ctx->n;
    return 0;
}

extern "C" int kzgamd_msm_uses_wide_table(void* msm) {
    if (!msm || !((MsmContext*)msm)->fbw) return 0;
    return ((MsmContext*)msm)->fbw_glv ? 2 : 1;
}

extern "C" int kzgamd_msm_set_profile(void* msm, int on) {
    if (!msm) return 1;
    kzgamd::msm_set_profile((MsmContext*)msm, on != 0);
    return 0;
}
extern "C" int kzgamd_msm_get_profile(void* msm, float* accum_ms, float* total_ms) {
    if (!msm || !accum_ms || !total_ms) return 1;
    int cnt = kzgamd::msm_get_profile((MsmContext*)msm, accum_ms, total_ms);
    return cnt > 0 ? cnt : -1;
}

// bench/test utility: n distinct G1 points P_i = h_i * G, blst affine layout, on the device
extern "C" RustError kzgamd_generate_points(void* d_out_affine, size_t n, uint64_t seed, void* stream) {
    if (!d_out_affine) return make_error(1, "null output");
    return guarded([&] {
        hipLaunchKernelGGL(k_gen_points, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                           (ff::Fp*)d_out_affine, n, seed);
        HIP_TRY(hipGetLastError());
    });
}

// device-resident bases: prepare != 0 builds the fixed-base rows (like prepare_msm)
extern "C" void* kzgamd_msm_create_device_ex(const void* d_points_affine, size_t npoints, int prepare, const KzgAmdConfig* cfg) {
    if (!d_points_affine || npoints == 0) return nullptr;
    return create_guarded("kzgamd_msm_create_device", cfg, [&](const kzgamd::Options& opt) {
        return kzgamd::msm_create(d_points_affine, npoints, true, prepare != 0, false, kzgamd::G1_CHECK, &opt);
    });
}
extern "C" void* kzgamd_msm_create_device(const void* d_points_affine, size_t npoints, int prepare) {
    return kzgamd_msm_create_device_ex(d_points_affine, npoints, prepare, nullptr);
}

// sum of n Jacobian points on the host: the combine step of a large MSM sharded over several GPUs (each rank's
// partial result is one 144-byte point; group addition is not a collective reduction op, so the partials are
// all-gathered and added locally).  Pure host arithmetic, no device needed.
extern "C" void kzgamd_g1_sum(blst_p1* out, const blst_p1* in, size_t n) {
    kzgamd::HostJac acc;
    acc.x = acc.y = acc.z = ff::Fp::zero();
    for (size_t i = 0; i < n; ++i) {
        kzgamd::HostJac p;
        memcpy(&p, &in[i], sizeof p);
        acc = kzgamd::host_jac_add(acc, p);
    }
    memcpy(out, &acc, sizeof acc);
}

extern "C" int kzgamd_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

extern "C" const char* kzgamd_version(void) { return "kzg_mi355x 0.1 (gfx950)"; }
