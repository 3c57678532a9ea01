// SHA-256 (FIPS 180-4) for the Fiat-Shamir challenge (kzg/src/eip_4844.rs:236-238, :920-945).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kzgamd {
class Sha256 {
  public:
    Sha256() { reset(); }
    void reset() {
        static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                       0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        memcpy(h_, iv, sizeof iv);
        fill_ = 0;
        total_ = 0;
    }
    void update(const uint8_t* p, size_t n) {
        total_ += n;
        if (fill_) {
            size_t take = 64 - fill_ < n ? 64 - fill_ : n;
            memcpy(buf_ + fill_, p, take);
            fill_ += take;
            p += take;
            n -= take;
            if (fill_ == 64) {
                block(buf_);
                fill_ = 0;
            }
        }
        while (n >= 64) {
            block(p);
            p += 64;
            n -= 64;
        }
        if (n) {
            memcpy(buf_, p, n);
            fill_ = n;
        }
    }
    void finish(uint8_t out[32]) {
        uint64_t bits = total_ * 8;
        uint8_t pad[72] = {0x80};
        size_t padlen = (fill_ < 56) ? 56 - fill_ : 120 - fill_;
        uint8_t len[8];
        for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(pad, padlen);
        update(len, 8);
        for (int i = 0; i < 8; ++i) {
            out[4 * i] = (uint8_t)(h_[i] >> 24);
            out[4 * i + 1] = (uint8_t)(h_[i] >> 16);
            out[4 * i + 2] = (uint8_t)(h_[i] >> 8);
            out[4 * i + 3] = (uint8_t)h_[i];
        }
    }

  private:
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#if defined(__x86_64__)
    static bool have_shani() {
        static const bool ok = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1");
        return ok;
    }
    // one 64-byte block with the SHA extensions (the challenge hash is 131 120 bytes per blob and sits on
    // the critical path of every blob proof)
    __attribute__((target("sha,sse4.1,ssse3"))) void block_ni(const uint8_t* data) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
            0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
            0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
            0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
            0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
            0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
            0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
            0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
        __m128i tmp = _mm_loadu_si128((const __m128i*)&h_[0]);     // a b c d
        __m128i st1 = _mm_loadu_si128((const __m128i*)&h_[4]);     // e f g h
        tmp = _mm_shuffle_epi32(tmp, 0xB1);                        // b a d c
        st1 = _mm_shuffle_epi32(st1, 0x1B);                        // h g f e
        __m128i st0 = _mm_alignr_epi8(tmp, st1, 8);                // a b e f
        st1 = _mm_blend_epi16(st1, tmp, 0xF0);                     // c d g h
        const __m128i save0 = st0, save1 = st1;
        __m128i m[4];
        for (int i = 0; i < 4; ++i) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 16 * i)), bswap);
        for (int r = 0; r < 16; ++r) {
            __m128i msg = _mm_add_epi32(m[r & 3], _mm_loadu_si128((const __m128i*)&K[4 * r]));
            st1 = _mm_sha256rnds2_epu32(st1, st0, msg);
            msg = _mm_shuffle_epi32(msg, 0x0E);
            st0 = _mm_sha256rnds2_epu32(st0, st1, msg);
            if (r < 12) {  // schedule words for round group r + 4
                __m128i w = _mm_sha256msg1_epu32(m[r & 3], m[(r + 1) & 3]);
                w = _mm_add_epi32(w, _mm_alignr_epi8(m[(r + 3) & 3], m[(r + 2) & 3], 4));
                m[r & 3] = _mm_sha256msg2_epu32(w, m[(r + 3) & 3]);
            }
        }
        st0 = _mm_add_epi32(st0, save0);
        st1 = _mm_add_epi32(st1, save1);
        tmp = _mm_shuffle_epi32(st0, 0x1B);                        // f e b a
        st1 = _mm_shuffle_epi32(st1, 0xB1);                        // d c h g
        st0 = _mm_blend_epi16(tmp, st1, 0xF0);                     // d c b a
        st1 = _mm_alignr_epi8(st1, tmp, 8);                        // h g f e
        _mm_storeu_si128((__m128i*)&h_[0], st0);
        _mm_storeu_si128((__m128i*)&h_[4], st1);
    }
#endif
    void block(const uint8_t* b) {
#if defined(__x86_64__)
        if (have_shani()) {
            block_ni(b);
            return;
        }
#endif
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
            0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
            0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
            0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
            0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
            0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
            0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
            0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; ++i)
            w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h_[0], bb = h_[1], c = h_[2], d = h_[3], e = h_[4], f = h_[5], g = h_[6], hh = h_[7];
        for (int i = 0; i < 64; ++i) {
            uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h_[0] += a; h_[1] += bb; h_[2] += c; h_[3] += d; h_[4] += e; h_[5] += f; h_[6] += g; h_[7] += hh;
    }
    uint32_t h_[8];
    uint8_t buf_[64];
    size_t fill_;
    uint64_t total_;
};
}  // namespace kzgamd
