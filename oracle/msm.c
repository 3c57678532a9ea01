/*
 * TEST INFRASTRUCTURE (see oracle.h).  The reference's CPU MSM for the hot path:
 * Booth-signed window Pippenger over XYZZ buckets, restated from
 * kzg/src/msm/tiling_pippenger_ops.rs:21-138 and kzg/src/msm/pippenger_utils.rs:231-317,
 * plus the g1_linear_combination wrapper (blst/src/kzg_proofs.rs:25-72,
 * kzg/src/msm/msm_impls.rs:40-61,114-148).
 */
#include "oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* pippenger_utils.rs:231-244.  Little-endian load of the (at most 4) bytes the
 * window [off, off+bits) touches, shifted down to bit `off`; bits above the
 * window are left in place ("trash") and masked by the caller.  The reference's
 * branch-free byte walk contributes zero for bytes past the window's last one. */
static uint64_t get_wval_limb(const uint8_t *d, size_t off, size_t bits) {
    size_t first = off / 8, last = (off + bits - 1) / 8;
    uint64_t ret = 0;
    for (size_t i = 0; i < 4 && first + i <= last; ++i) ret |= (uint64_t)d[first + i] << (8 * i);
    return ret >> (off % 8);
}

/* pippenger_utils.rs:251-256 */
static uint64_t booth_encode(uint64_t wval, size_t sz) {
    uint64_t mask = 0 - (wval >> sz);
    wval = (wval + 1) >> 1;
    return (wval ^ mask) - mask;
}

/* pippenger_utils.rs:270-281 */
static void booth_decode(og1_xyzz_t *buckets, uint64_t booth_idx, size_t wbits, const og1_affine_t *p) {
    int booth_sign = (booth_idx >> wbits) & 1;
    booth_idx &= ((uint64_t)1 << wbits) - 1;
    if (booth_idx != 0) og1_xyzz_dadd_affine(&buckets[booth_idx - 1], p, booth_sign);
}

/* pippenger_utils.rs:300-317 */
size_t opippenger_window_size(size_t npoints) {
    size_t wbits = 0;
    for (size_t v = npoints; v; v >>= 1) ++wbits;
    if (wbits > 13) return wbits - 4;
    if (wbits > 5) return wbits - 3;
    return 2;
}

static int xyzz_is_zero(const og1_xyzz_t *p) {
    const uint64_t *w = (const uint64_t *)p;
    uint64_t acc = 0;
    for (size_t i = 0; i < sizeof *p / 8; ++i) acc |= w[i];
    return acc == 0;
}

/* tiling_pippenger_ops.rs:21-45 */
static void integrate_buckets(og1_t *out, og1_xyzz_t *buckets, size_t wbits) {
    size_t n = ((size_t)1 << wbits) - 1;
    og1_xyzz_t ret = buckets[n], acc = buckets[n];
    memset(&buckets[n], 0, sizeof buckets[n]);
    while (n--) {
        if (!xyzz_is_zero(&buckets[n])) {
            og1_xyzz_dadd(&acc, &buckets[n]);
            memset(&buckets[n], 0, sizeof buckets[n]);
        }
        og1_xyzz_dadd(&ret, &acc);
    }
    og1_xyzz_to_jacobian(out, &ret);
}

/* tiling_pippenger_ops.rs:68-104 */
static void tile_pippenger(og1_t *ret, const og1_affine_t *points, const uint8_t *scalars, size_t n,
                           og1_xyzz_t *buckets, size_t bit0, size_t wbits, size_t cbits) {
    uint64_t wmask = ((uint64_t)1 << (wbits + 1)) - 1;
    uint64_t z = (bit0 == 0);
    bit0 -= (size_t)(z ^ 1);
    wbits += (size_t)(z ^ 1);
    for (size_t i = 0; i < n; ++i) {
        uint64_t wval = (get_wval_limb(scalars + 32 * i, bit0, wbits) << z) & wmask;
        wval = booth_encode(wval, cbits);
        booth_decode(buckets, wval, cbits, &points[i]);
    }
    integrate_buckets(ret, buckets, cbits - 1);
}

/* tiling_pippenger_ops.rs:106-138 */
void omsm_tiling_pippenger(og1_t *out, const og1_affine_t *points, const uint8_t *scalars, size_t n) {
    size_t window = opippenger_window_size(n);
    og1_xyzz_t *buckets = calloc((size_t)1 << (window - 1), sizeof *buckets);
    size_t wbits = 255 % window, cbits = wbits + 1, bit0 = 255;
    og1_t tile, ret;
    og1_set_inf(&ret);
    for (;;) {
        bit0 -= wbits;
        if (bit0 == 0) break;
        tile_pippenger(&tile, points, scalars, n, buckets, bit0, wbits, cbits);
        og1_add_or_dbl(&ret, &ret, &tile);
        for (size_t i = 0; i < window; ++i) og1_dbl(&ret, &ret);
        cbits = window;
        wbits = window;
    }
    tile_pippenger(&tile, points, scalars, n, buckets, 0, wbits, cbits);
    og1_add_or_dbl(&ret, &ret, &tile);
    free(buckets);
    *out = ret;
}

void omsm_naive(og1_t *out, const og1_affine_t *points, const ofr_t *scalars, size_t n) {
    og1_t acc, p, t;
    og1_set_inf(&acc);
    for (size_t i = 0; i < n; ++i) {
        og1_from_affine(&p, &points[i]);
        og1_mul(&t, &p, &scalars[i]);
        og1_add_or_dbl(&acc, &acc, &t);
    }
    *out = acc;
}

/* The sppark-boundary shape: affine points (infinity = (0,0)), Montgomery scalars.
 * len < 8 -> naive (kzg_proofs.rs:37-45); else to_scalar + tiling_pippenger. */
void omsm_affine(og1_t *out, const og1_affine_t *points, const ofr_t *scalars, size_t n) {
    if (n < 8) {
        omsm_naive(out, points, scalars, n);
        return;
    }
    uint8_t *le = malloc(32 * n);
    for (size_t i = 0; i < n; ++i) ofr_to_scalar_le(le + 32 * i, &scalars[i]);
    omsm_tiling_pippenger(out, points, le, n);
    free(le);
}

/* g1_linear_combination without precomputation: filter infinities
 * (msm_impls.rs:50-55), batch-convert to affine (:103-111), Pippenger. */
void og1_lincomb(og1_t *out, const og1_t *points, const ofr_t *scalars, size_t n) {
    if (n < 8) {
        og1_t acc, t;
        og1_set_inf(&acc);
        for (size_t i = 0; i < n; ++i) {
            og1_mul(&t, &points[i], &scalars[i]);
            og1_add_or_dbl(&acc, &acc, &t);
        }
        *out = acc;
        return;
    }
    og1_affine_t *aff = malloc(n * sizeof *aff);
    ofr_t *sc = malloc(n * sizeof *sc);
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        if (og1_is_inf(&points[i])) continue;
        og1_to_affine(&aff[m], &points[i]);
        sc[m++] = scalars[i];
    }
    if (m == 0) og1_set_inf(out);
    else {
        uint8_t *le = malloc(32 * m);
        for (size_t i = 0; i < m; ++i) ofr_to_scalar_le(le + 32 * i, &sc[i]);
        omsm_tiling_pippenger(out, aff, le, m);
        free(le);
    }
    free(aff);
    free(sc);
}

/* CPU baseline helper: split the point range over threads, each runs the
 * sequential Pippenger on its slice, partial sums added.  (The reference's
 * parallel variant tiles points x windows, tiling_parallel_pippenger.rs:70-186;
 * this is the simpler decomposition with the same per-add cost.) */
typedef struct {
    og1_t out;
    const og1_affine_t *points;
    const ofr_t *scalars;
    size_t n;
} mt_job_t;

static void *mt_worker(void *arg) {
    mt_job_t *j = arg;
    omsm_affine(&j->out, j->points, j->scalars, j->n);
    return NULL;
}

void omsm_affine_mt(og1_t *out, const og1_affine_t *points, const ofr_t *scalars, size_t n, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n / 64) nthreads = (int)(n / 64 ? n / 64 : 1);
    mt_job_t *jobs = calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *th = calloc((size_t)nthreads, sizeof *th);
    size_t per = (n + (size_t)nthreads - 1) / (size_t)nthreads;
    for (int t = 0; t < nthreads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per > n ? n : lo + per;
        if (lo > n) lo = n;
        jobs[t].points = points + lo;
        jobs[t].scalars = scalars + lo;
        jobs[t].n = hi - lo;
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    og1_t acc;
    og1_set_inf(&acc);
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        og1_add_or_dbl(&acc, &acc, &jobs[t].out);
    }
    free(jobs);
    free(th);
    *out = acc;
}

/* ---- BGMW fixed-base MSM: the reference's DEFAULT algorithm for the 4096-point setup when built with the `bgmw`
 * feature (kzg/src/msm/bgmw.rs; precomputation :206-232, evaluation multiply_sequential :381-439).  Restated for the
 * CPU baseline and pinned on the same commitment vectors as the tiling Pippenger above. */

/* bgmw.rs:89-129 */
size_t obgmw_window_size(size_t npoints) {
    size_t wbits = 0;
    for (size_t v = npoints; v; v >>= 1) ++wbits;
    static const unsigned char tab[38] = {0, 4, 5, 5, 6, 7, 8, 8, 9, 10, 10, 11, 12, 13, 13, 15, 15, 16, 17, 17, 19, 20, 20, 22,
                                          22, 24, 24, 26, 26, 26, 29, 29, 29, 32, 32, 32, 32, 32};
    return wbits <= 37 ? (wbits == 0 ? 4 : tab[wbits]) : 37;
}

/* get_table_dimensions (bgmw.rs:49-69): rows = ceil(255 / w) + (255 % w == 0) */
static size_t bgmw_rows(size_t w) { return (255 + w - 1) / w + (255 % w == 0 ? 1 : 0); }

/* BgmwTable::new (bgmw.rs:206-232): table[j * n + i] = affine(2^(w j) * P_i), j < rows */
int obgmw_table_new(obgmw_table_t *t, const og1_affine_t *points, size_t n) {
    t->n = n;
    t->window = obgmw_window_size(n);
    t->h = bgmw_rows(t->window);
    t->table = malloc(n * t->h * sizeof(og1_affine_t));
    if (!t->table) return 1;
    for (size_t i = 0; i < n; ++i) {
        og1_t p;
        og1_from_affine(&p, &points[i]);
        for (size_t j = 0; j < t->h; ++j) {
            og1_to_affine(&t->table[j * n + i], &p);
            for (size_t k = 0; k < t->window; ++k) og1_dbl(&p, &p); /* tmp_point.mul(q), q = 2^w */
        }
    }
    return 0;
}

void obgmw_table_free(obgmw_table_t *t) {
    free(t->table);
    memset(t, 0, sizeof *t);
}

/* p1_tile_bgmw (bgmw.rs:601-680): the digit of window [bit0, bit0 + wbits) of every scalar selects a bucket of ONE
 * shared set; the point comes from the table row of that window */
static void tile_bgmw(const og1_affine_t *points, const uint8_t *scalars, size_t n, og1_xyzz_t *buckets, size_t bit0,
                      size_t wbits, size_t cbits) {
    uint64_t wmask = ((uint64_t)1 << (wbits + 1)) - 1;
    uint64_t z = (bit0 == 0);
    bit0 -= (size_t)(z ^ 1);
    wbits += (size_t)(z ^ 1);
    for (size_t i = 0; i < n; ++i) {
        uint64_t wval = (get_wval_limb(scalars + 32 * i, bit0, wbits) << z) & wmask;
        wval = booth_encode(wval, cbits);
        booth_decode(buckets, wval, cbits, &points[i]);
    }
}

/* multiply_sequential_raw (bgmw.rs:381-425): scalars are canonical 32-byte little-endian, n of them (n <= t->n) */
void obgmw_multiply(og1_t *out, const obgmw_table_t *t, const uint8_t *scalars_le, size_t n) {
    size_t window = t->window;
    og1_xyzz_t *buckets = calloc((size_t)1 << (window - 1), sizeof *buckets);
    size_t wbits = 255 % window, cbits = wbits + 1, bit0 = 255, q_idx = t->h;
    for (;;) {
        bit0 -= wbits;
        q_idx -= 1;
        if (bit0 == 0) break;
        tile_bgmw(t->table + q_idx * t->n, scalars_le, n, buckets, bit0, wbits, cbits);
        cbits = window;
        wbits = window;
    }
    tile_bgmw(t->table, scalars_le, n, buckets, 0, wbits, cbits);
    integrate_buckets(out, buckets, wbits - 1);
    free(buckets);
}
