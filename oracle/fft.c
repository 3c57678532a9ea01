/*
 * TEST INFRASTRUCTURE (see oracle.h).  Radix-2 NTT over Fr and the DAS extension,
 * restated from blst/src/fft_fr.rs:14-186, blst/src/data_availability_sampling.rs:14-100,
 * blst/src/types/fft_settings.rs:28-106 and kzg/src/common_utils.rs:6-34.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* SCALE2_ROOT_OF_UNITY[k] (blst/src/consts.rs:17-50) = 7^((r-1)/2^k) mod r, canonical limbs.
 * Generated here instead of tabulated; tests/test_oracle_golden.py checks entries
 * against the values the reference tabulates. */
void oscale2_root_of_unity(uint64_t out[4], unsigned scale) {
    /* (r-1)/2^32 */
    static const uint64_t E32[4] = {0xfffe5bfeffffffffull, 0x09a1d80553bda402ull, 0x299d7d483339d808ull,
                                    0x0000000073eda753ull};
    ofr_t g, acc, base;
    ofr_from_u64(&g, 7);
    ofr_one(&acc);
    base = g;
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 64; ++b) {
            if ((E32[i] >> b) & 1) ofr_mul(&acc, &acc, &base);
            ofr_sqr(&base, &base);
        }
    /* acc = primitive 2^32-th root; square down to order 2^scale */
    for (unsigned k = 32; k > scale; --k) ofr_sqr(&acc, &acc);
    ofr_to_u64_arr(out, &acc);
}

/* kzg/src/common_utils.rs:6-34 */
void oreverse_bit_order(void *data, size_t elem_size, size_t n) {
    if (n < 2) return;
    unsigned bits = 0;
    while (((size_t)1 << bits) < n) ++bits;
    uint8_t *p = data, *tmp = malloc(elem_size);
    for (size_t i = 0; i < n; ++i) {
        size_t r = 0;
        for (unsigned b = 0; b < bits; ++b)
            if (i & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
        if (r > i) {
            memcpy(tmp, p + i * elem_size, elem_size);
            memcpy(p + i * elem_size, p + r * elem_size, elem_size);
            memcpy(p + r * elem_size, tmp, elem_size);
        }
    }
    free(tmp);
}

/* FsFFTSettings::new + expand_root_of_unity, fft_settings.rs:28-106 */
int offt_settings_new(offt_settings_t *fs, unsigned scale) {
    if (scale >= 32) return 1;
    size_t w = (size_t)1 << scale;
    uint64_t root_raw[4];
    ofr_t root;
    oscale2_root_of_unity(root_raw, scale);
    ofr_from_u64_arr(&root, root_raw);
    fs->max_width = w;
    fs->roots_of_unity = malloc((w + 1) * sizeof(ofr_t));
    fs->reverse_roots_of_unity = malloc((w + 1) * sizeof(ofr_t));
    fs->brp_roots_of_unity = malloc(w * sizeof(ofr_t));
    ofr_one(&fs->roots_of_unity[0]);
    for (size_t i = 1; i <= w; ++i) ofr_mul(&fs->roots_of_unity[i], &fs->roots_of_unity[i - 1], &root);
    if (!ofr_is_one(&fs->roots_of_unity[w])) return 2;
    for (size_t i = 0; i <= w; ++i) fs->reverse_roots_of_unity[i] = fs->roots_of_unity[w - i];
    memcpy(fs->brp_roots_of_unity, fs->roots_of_unity, w * sizeof(ofr_t));
    oreverse_bit_order(fs->brp_roots_of_unity, sizeof(ofr_t), w);
    return 0;
}

void offt_settings_free(offt_settings_t *fs) {
    free(fs->roots_of_unity);
    free(fs->reverse_roots_of_unity);
    free(fs->brp_roots_of_unity);
    memset(fs, 0, sizeof *fs);
}

/* fft_fr_fast_inner, fft_fr.rs:49-108: recursive out-of-place DIT */
static void fft_fast(ofr_t *ret, size_t n, const ofr_t *data, size_t data_start, size_t stride,
                     const ofr_t *roots, size_t roots_stride) {
    size_t half = n / 2;
    if (half > 0) {
        fft_fast(ret, half, data, data_start, stride * 2, roots, roots_stride * 2);
        fft_fast(ret + half, half, data, data_start + stride, stride * 2, roots, roots_stride * 2);
        for (size_t i = 0; i < half; ++i) {
            ofr_t y_times_root;
            ofr_mul(&y_times_root, &ret[i + half], &roots[i * roots_stride]);
            ofr_sub(&ret[i + half], &ret[i], &y_times_root);
            ofr_add(&ret[i], &ret[i], &y_times_root);
        }
    } else {
        ret[0] = data[data_start];
    }
}

/* fft_fr_output, fft_fr.rs:112-153 */
int offt_fr(const offt_settings_t *fs, ofr_t *out, const ofr_t *in, size_t n, int inverse) {
    if (n > fs->max_width) return 1;
    if (n == 0 || (n & (n - 1))) return 2;
    size_t stride = fs->max_width / n;
    const ofr_t *roots = inverse ? fs->reverse_roots_of_unity : fs->roots_of_unity;
    fft_fast(out, n, in, 0, 1, roots, stride);
    if (inverse) {
        ofr_t inv_len;
        ofr_from_u64(&inv_len, (uint64_t)n);
        ofr_inv(&inv_len, &inv_len);
        for (size_t i = 0; i < n; ++i) ofr_mul(&out[i], &out[i], &inv_len);
    }
    return 0;
}

/* fft_fr_slow, fft_fr.rs:168-186 (forward, stride 1) */
void offt_fr_slow(const offt_settings_t *fs, ofr_t *out, const ofr_t *in, size_t n) {
    size_t roots_stride = fs->max_width / n;
    for (size_t i = 0; i < n; ++i) {
        ofr_mul(&out[i], &in[0], &fs->roots_of_unity[0]);
        for (size_t j = 1; j < n; ++j) {
            ofr_t v;
            ofr_mul(&v, &in[j], &fs->roots_of_unity[((i * j) % n) * roots_stride]);
            ofr_add(&out[i], &out[i], &v);
        }
    }
}

/* das_fft_extension_stride, data_availability_sampling.rs:14-72 */
static void das_stride(const offt_settings_t *fs, ofr_t *ab, size_t n, size_t stride) {
    if (n < 2) return;
    if (n == 2) {
        ofr_t x, y, yr;
        ofr_add(&x, &ab[0], &ab[1]);
        ofr_sub(&y, &ab[0], &ab[1]);
        ofr_mul(&yr, &y, &fs->roots_of_unity[stride]);
        ofr_add(&ab[0], &x, &yr);
        ofr_sub(&ab[1], &x, &yr);
        return;
    }
    size_t half = n / 2;
    for (size_t i = 0; i < half; ++i) {
        ofr_t t1, t2;
        ofr_add(&t1, &ab[i], &ab[half + i]);
        ofr_sub(&t2, &ab[i], &ab[half + i]);
        ofr_mul(&ab[half + i], &t2, &fs->reverse_roots_of_unity[i * 2 * stride]);
        ab[i] = t1;
    }
    das_stride(fs, ab, half, stride * 2);
    das_stride(fs, ab + half, half, stride * 2);
    for (size_t i = 0; i < half; ++i) {
        ofr_t x = ab[i], y = ab[half + i], yr;
        ofr_mul(&yr, &y, &fs->roots_of_unity[(1 + 2 * i) * stride]);
        ofr_add(&ab[i], &x, &yr);
        ofr_sub(&ab[half + i], &x, &yr);
    }
}

/* das_fft_extension, data_availability_sampling.rs:78-100 */
int odas_fft_extension(const offt_settings_t *fs, ofr_t *odds, const ofr_t *evens, size_t n) {
    if (n == 0) return 1;
    if (n & (n - 1)) return 2;
    if (n * 2 > fs->max_width) return 3;
    size_t stride = fs->max_width / (n * 2);
    memmove(odds, evens, n * sizeof(ofr_t));
    das_stride(fs, odds, n, stride);
    ofr_t inv_len;
    ofr_from_u64(&inv_len, (uint64_t)n);
    ofr_inv(&inv_len, &inv_len);
    for (size_t i = 0; i < n; ++i) ofr_mul(&odds[i], &odds[i], &inv_len);
    return 0;
}

/* fft_g1_fast, blst/src/fft_g1.rs:13-52: the same recursive DIT network over G1 points; the butterfly
 * multiplies by the root with a full scalar multiplication (FsG1::mul) */
static void fft_g1_fast(og1_t *ret, size_t n, const og1_t *data, size_t stride, const ofr_t *roots, size_t roots_stride) {
    size_t half = n / 2;
    if (half > 0) {
        fft_g1_fast(ret, half, data, stride * 2, roots, roots_stride * 2);
        fft_g1_fast(ret + half, half, data + stride, stride * 2, roots, roots_stride * 2);
        for (size_t i = 0; i < half; ++i) {
            og1_t y_times_root, neg;
            og1_mul(&y_times_root, &ret[i + half], &roots[i * roots_stride]);
            og1_neg(&neg, &y_times_root);
            og1_add_or_dbl(&ret[i + half], &ret[i], &neg);
            og1_add_or_dbl(&ret[i], &ret[i], &y_times_root);
        }
    } else {
        ret[0] = data[0];
    }
}

/* FFTG1::fft_g1 for FsFFTSettings, blst/src/fft_g1.rs:54-83.  0 ok; 1 too long; 2 not a power of two */
int offt_g1(const offt_settings_t *fs, og1_t *out, const og1_t *in, size_t n, int inverse) {
    if (n > fs->max_width) return 1;
    if (n == 0 || (n & (n - 1))) return 2;
    size_t stride = fs->max_width / n;
    fft_g1_fast(out, n, in, 1, inverse ? fs->reverse_roots_of_unity : fs->roots_of_unity, stride);
    if (inverse) {
        ofr_t inv_len;
        ofr_from_u64(&inv_len, (uint64_t)n);
        ofr_inv(&inv_len, &inv_len);
        for (size_t i = 0; i < n; ++i) og1_mul(&out[i], &out[i], &inv_len);
    }
    return 0;
}

/* fft_g1_slow, blst/src/fft_g1.rs:86-104 (forward, stride 1) */
void offt_g1_slow(const offt_settings_t *fs, og1_t *out, const og1_t *in, size_t n) {
    size_t roots_stride = fs->max_width / n;
    for (size_t i = 0; i < n; ++i) {
        og1_mul(&out[i], &in[0], &fs->roots_of_unity[0]);
        for (size_t j = 1; j < n; ++j) {
            og1_t v;
            og1_mul(&v, &in[j], &fs->roots_of_unity[((i * j) % n) * roots_stride]);
            og1_add_or_dbl(&out[i], &out[i], &v);
        }
    }
}
