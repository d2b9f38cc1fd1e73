// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of yuki/src/sampling/{mod,uniform,stratified}.rs.
//
// Third-party arithmetic that is NOT under /root/reference (SURVEY.md §8(c),
// Appendix A) is restated from the published algorithms and pinned by our own
// known-answer tests (tests/test_oracle_sampler.py):
//   * rand_pcg 0.3 `Pcg32` = PCG-XSH-RR 64/32 (pcg32 C reference: seed 42/54)
//   * rand 0.8 `Standard` for f32: (next_u32() >> 8) * 2^-24
//   * std `DefaultHasher::default()` = SipHash-1-3, zero keys, fields fed as
//     native-endian ints without length prefixes (reference SipHash vectors
//     are for 2-4; our KAT is the 1-3 variant cross-checked against a
//     bit-level Python restatement)
// Parity for these three is therefore "unpinned by the reference".
#pragma once
#include <cstdint>

#include "omath.h"

namespace orc {

struct SipHasher13 {
    uint64_t v0, v1, v2, v3;
    uint64_t tail;   // unprocessed bytes, little endian
    unsigned ntail;  // number of bytes in tail
    uint64_t length;
    SipHasher13() : tail(0), ntail(0), length(0) {
        v0 = 0x736f6d6570736575ULL;
        v1 = 0x646f72616e646f6dULL;
        v2 = 0x6c7967656e657261ULL;
        v3 = 0x7465646279746573ULL;
    }
    static inline uint64_t rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
    inline void round() {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    }
    void write(const uint8_t* p, unsigned n) {
        for (unsigned i = 0; i < n; ++i) {
            tail |= (uint64_t)p[i] << (8 * ntail);
            ++ntail;
            ++length;
            if (ntail == 8) {
                v3 ^= tail;
                round();
                v0 ^= tail;
                tail = 0;
                ntail = 0;
            }
        }
    }
    void write_u16(uint16_t v) { uint8_t b[2] = {(uint8_t)v, (uint8_t)(v >> 8)}; write(b, 2); }
    void write_u32(uint32_t v) {
        uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
        write(b, 4);
    }
    void write_u64(uint64_t v) {
        uint8_t b[8];
        for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(v >> (8 * i));
        write(b, 8);
    }
    uint64_t finish() const {
        SipHasher13 s = *this;
        uint64_t b = (s.length << 56) | s.tail;
        s.v3 ^= b;
        s.round();
        s.v0 ^= b;
        s.v2 ^= 0xff;
        s.round();
        s.round();
        s.round();
        return s.v0 ^ s.v1 ^ s.v2 ^ s.v3;
    }
};

// hash_values!(pixel)  — sampling/mod.rs:90-103 with Point2<u16> deriving Hash
inline uint64_t hash_pixel(uint16_t x, uint16_t y) {
    SipHasher13 h;
    h.write_u16(x);
    h.write_u16(y);
    return h.finish();
}
// hash_values!(pixel, dimension:u32, rng_seed:u64) — stratified.rs:105,122
inline uint64_t hash_pixel_dim_seed(uint16_t x, uint16_t y, uint32_t dimension, uint64_t seed) {
    SipHasher13 h;
    h.write_u16(x);
    h.write_u16(y);
    h.write_u32(dimension);
    h.write_u64(seed);
    return h.finish();
}

struct Pcg32 {
    uint64_t state, inc;
    static constexpr uint64_t MULT = 6364136223846793005ULL;
    Pcg32() : state(0), inc(1) {}
    Pcg32(uint64_t st, uint64_t stream) {
        inc = (stream << 1) | 1;
        state = st + inc;
        state = state * MULT + inc;
    }
    uint32_t next_u32() {
        uint64_t old = state;
        state = old * MULT + inc;
        uint32_t xsh = (uint32_t)(((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t)(old >> 59);
        return (xsh >> rot) | (xsh << ((32 - rot) & 31));
    }
    void advance(uint64_t delta) {
        uint64_t acc_mult = 1, acc_plus = 0, cur_mult = MULT, cur_plus = inc;
        while (delta > 0) {
            if (delta & 1) {
                acc_mult *= cur_mult;
                acc_plus = acc_plus * cur_mult + cur_plus;
            }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta >>= 1;
        }
        state = acc_mult * state + acc_plus;
    }
    // rand 0.8 Standard f32
    float next_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
};

// stratified.rs:147-178
inline uint32_t permutation_element(uint32_t i, uint32_t l, uint32_t p) {
    uint32_t w = l - 1;
    w |= w >> 1;
    w |= w >> 2;
    w |= w >> 4;
    w |= w >> 8;
    w |= w >> 16;
    do {
        i ^= p;
        i *= 0xe170893d;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8;
        i *= 0x0929eb3f;
        i ^= p >> 23;
        i ^= (i & w) >> 1;
        i *= 1 | p >> 27;
        i *= 0x6935fa69;
        i ^= (i & w) >> 11;
        i *= 0x74dcb303;
        i ^= (i & w) >> 2;
        i *= 0x9e501cc3;
        i ^= (i & w) >> 2;
        i *= 0xc860a3df;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}

enum SamplerKind { SAMPLER_UNIFORM = 0, SAMPLER_STRATIFIED = 1 };

// One struct for both samplers (trait Sampler, sampling/mod.rs:46-57).
// The seed is explicit (quirk 20: the reference draws it from thread_rng).
struct Sampler {
    int kind;
    uint32_t nx, ny;  // Uniform: nx = pixel_samples, ny = 1
    bool jitter;
    uint64_t rng_seed;
    // per pixel-sample state
    uint16_t px, py;
    uint32_t sample_index, dimension;
    Pcg32 rng;

    uint32_t samples_per_pixel() const { return kind == SAMPLER_UNIFORM ? nx : nx * ny; }

    // uniform.rs:72-84 / stratified.rs:90-102 (the latter zeroes `dimension`
    // but still advances the PCG by the argument — quirk 8)
    void start_pixel_sample(uint16_t x, uint16_t y, uint32_t index, uint32_t dim) {
        px = x;
        py = y;
        sample_index = index;
        dimension = kind == SAMPLER_UNIFORM ? dim : 0;
        uint64_t hashed = hash_pixel(x, y);
        rng = Pcg32(rng_seed, hashed);
        rng.advance((uint64_t)index * 65536ULL + (uint64_t)dim);
    }
    float get_1d() {
        if (kind == SAMPLER_UNIFORM) {
            dimension += 1;
            return rng.next_f32();
        }
        uint64_t hashed = hash_pixel_dim_seed(px, py, dimension, rng_seed);
        uint32_t stratum = permutation_element(sample_index, samples_per_pixel(), (uint32_t)hashed);
        dimension += 1;
        float delta = jitter ? rng.next_f32() : 0.5f;
        return ((float)stratum + delta) / (float)samples_per_pixel();
    }
    Point2f get_2d() {
        if (kind == SAMPLER_UNIFORM) {
            dimension += 2;
            float a = rng.next_f32();
            float b = rng.next_f32();
            return Point2f(a, b);
        }
        uint64_t hashed = hash_pixel_dim_seed(px, py, dimension, rng_seed);
        uint32_t stratum = permutation_element(sample_index, samples_per_pixel(), (uint32_t)hashed);
        dimension += 2;
        uint32_t x = stratum % nx;
        uint32_t y = stratum / ny;  // sic, stratified.rs:128
        float dx = jitter ? rng.next_f32() : 0.5f;
        float dy = jitter ? rng.next_f32() : 0.5f;
        return Point2f(((float)x + dx) / (float)nx, ((float)y + dy) / (float)ny);
    }
};

}  // namespace orc
