// yk_rng.h — the reference's samplers as inline integer code for the kernels.
//   trait Sampler                      yuki/src/sampling/mod.rs:46-57
//   UniformSampler                     sampling/uniform.rs:54-95
//   StratifiedSampler                  sampling/stratified.rs:70-144
//   permutation_element                sampling/stratified.rs:147-178
//   hash_values! = std DefaultHasher   sampling/mod.rs:90-103 (SipHash-1-3, zero keys)
//   Pcg32 (rand_pcg 0.3), Standard f32 (rand 0.8): published algorithms,
//   SURVEY.md Appendix A — not under /root/reference.
#pragma once
#include "yk_math.h"

namespace yk {

typedef unsigned long long u64;

YK_HD u64 rotl64(u64 x, int b) { return (x << b) | (x >> (64 - b)); }

#define YK_SIPROUND(v0, v1, v2, v3) \
    do {                            \
        v0 += v1;                   \
        v1 = rotl64(v1, 13);        \
        v1 ^= v0;                   \
        v0 = rotl64(v0, 32);        \
        v2 += v3;                   \
        v3 = rotl64(v3, 16);        \
        v3 ^= v2;                   \
        v0 += v3;                   \
        v3 = rotl64(v3, 21);        \
        v3 ^= v0;                   \
        v2 += v1;                   \
        v1 = rotl64(v1, 17);        \
        v1 ^= v2;                   \
        v2 = rotl64(v2, 32);        \
    } while (0)

// SipHash-1-3, k0=k1=0, message = up to two full 8-byte words + length byte.
// n_words full words m0,m1 followed by `tail` (remaining bytes, little endian)
// for a message of total_len bytes.
YK_HD u64 siphash13_words(int n_words, u64 m0, u64 m1, u64 tail, unsigned total_len) {
#ifdef YK_ABLATE_HASH  // timing builds only (tools/gpu_shade_ablation.sh): what the sampler's SipHash-1-3 costs; the image changes
    return (m0 * 0x9E3779B97F4A7C15ull) ^ (m1 + tail + total_len + (unsigned)n_words);
#endif
    u64 v0 = 0x736f6d6570736575ULL, v1 = 0x646f72616e646f6dULL, v2 = 0x6c7967656e657261ULL, v3 = 0x7465646279746573ULL;
    if (n_words > 0) {
        v3 ^= m0;
        YK_SIPROUND(v0, v1, v2, v3);
        v0 ^= m0;
    }
    if (n_words > 1) {
        v3 ^= m1;
        YK_SIPROUND(v0, v1, v2, v3);
        v0 ^= m1;
    }
    u64 b = ((u64)total_len << 56) | tail;
    v3 ^= b;
    YK_SIPROUND(v0, v1, v2, v3);
    v0 ^= b;
    v2 ^= 0xff;
    YK_SIPROUND(v0, v1, v2, v3);
    YK_SIPROUND(v0, v1, v2, v3);
    YK_SIPROUND(v0, v1, v2, v3);
    return v0 ^ v1 ^ v2 ^ v3;
}

// hash_values!(pixel): Point2<u16>{x,y} -> 4 bytes x_lo x_hi y_lo y_hi
YK_HD u64 hash_pixel(unsigned px, unsigned py) { return siphash13_words(0, 0, 0, (u64)(px & 0xffff) | ((u64)(py & 0xffff) << 16), 4); }
// hash_values!(pixel, dimension: u32, rng_seed: u64): 16 bytes = two words
YK_HD u64 hash_pixel_dim_seed(unsigned px, unsigned py, unsigned dimension, u64 seed) {
    u64 m0 = (u64)(px & 0xffff) | ((u64)(py & 0xffff) << 16) | ((u64)dimension << 32);
    return siphash13_words(2, m0, seed, 0, 16);
}

#define YK_PCG_MULT 6364136223846793005ULL

struct Pcg {
    u64 state, inc;
};
YK_HD Pcg pcg_new(u64 st, u64 stream) {
    Pcg r;
    r.inc = (stream << 1) | 1;
    r.state = st + r.inc;
    r.state = r.state * YK_PCG_MULT + r.inc;
    return r;
}
YK_HD unsigned pcg_next(Pcg& r) {
    u64 old = r.state;
    r.state = old * YK_PCG_MULT + r.inc;
    unsigned xsh = (unsigned)(((old >> 18) ^ old) >> 27);
    unsigned rot = (unsigned)(old >> 59);
    return (xsh >> rot) | (xsh << ((32 - rot) & 31));
}
YK_HD void pcg_advance(Pcg& r, u64 delta) {
    u64 acc_mult = 1, acc_plus = 0, cur_mult = YK_PCG_MULT, cur_plus = r.inc;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    r.state = acc_mult * r.state + acc_plus;
}
// advance by index * 65536 + dim (Sampler::start_pixel_sample).  Jumping is additive, so the `dim` steps go through the loop above
// (none for a camera sample) and the index * 2^16 steps start sixteen squarings in: after k squarings the loop's cur_mult is
// MULT^(2^k) and its cur_plus is inc times the product of (MULT^(2^j) + 1), j < k — constants (mod 2^64), whatever the stream.
// The same state bit for bit, at a third of the instructions of the plain loop for index < 64.
#define YK_PCG_MULT_2_16 0x902da3ff53640001ULL  // MULT^(2^16)
#define YK_PCG_PLUS_2_16 0x39f376e3016b0000ULL  // prod_{j<16} (MULT^(2^j) + 1)
YK_HD void pcg_advance_sample(Pcg& r, unsigned index, unsigned dim) {
    pcg_advance(r, (u64)dim);
    u64 acc_mult = 1, acc_plus = 0, cur_mult = YK_PCG_MULT_2_16, cur_plus = YK_PCG_PLUS_2_16 * r.inc;
    for (unsigned delta = index; delta > 0; delta >>= 1) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
    }
    r.state = acc_mult * r.state + acc_plus;
}
// rand 0.8 Standard for f32
YK_HD float pcg_f32(Pcg& r) { return (float)(pcg_next(r) >> 8) * (1.0f / 16777216.0f); }

YK_HD unsigned permutation_element(unsigned i, unsigned l, unsigned p) {
    unsigned w = l - 1;
    w |= w >> 1;
    w |= w >> 2;
    w |= w >> 4;
    w |= w >> 8;
    w |= w >> 16;
    do {
        i ^= p;
        i *= 0xe170893du;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8;
        i *= 0x0929eb3fu;
        i ^= p >> 23;
        i ^= (i & w) >> 1;
        i *= 1u | p >> 27;
        i *= 0x6935fa69u;
        i ^= (i & w) >> 11;
        i *= 0x74dcb303u;
        i ^= (i & w) >> 2;
        i *= 0x9e501cc3u;
        i ^= (i & w) >> 2;
        i *= 0xc860a3dfu;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    unsigned s = i + p;
    return (l & w) == 0u ? (s & w) : s % l;  // l power of two <=> (l & (l-1)) == 0 ; w = l-1 then
}

// Sampler parameters, constant over a render
struct SamplerCfg {
    unsigned kind;  // 0 uniform, 1 stratified
    unsigned nx, ny;
    unsigned jitter;
    u64 seed;
    unsigned spp;
};

// Per pixel-sample sampler state carried by a path
struct SamplerState {
    Pcg rng;
    unsigned px, py;
    unsigned sample_index;
    unsigned dimension;
};

// start_pixel_sample(p, index, dimension): uniform.rs:72-84, stratified.rs:90-102
// (the stratified sampler resets its own dimension to 0 but advances the PCG by
// the argument — quirk 8)
YK_HD SamplerState sampler_start(const SamplerCfg& c, unsigned px, unsigned py, unsigned index, unsigned dim) {
    SamplerState s;
    s.px = px;
    s.py = py;
    s.sample_index = index;
    s.dimension = c.kind == 0 ? dim : 0;
    s.rng = pcg_new(c.seed, hash_pixel(px, py));
    pcg_advance_sample(s.rng, index, dim);  // == pcg_advance(index * 65536 + dim)
    return s;
}
// What the camera sample's sampler start and first 2-D draw derive from the PIXEL alone (every sample of the pixel shares it):
// the PCG stream = SipHash-1-3 of the pixel, and the stratified sampler's hash of (pixel, dimension 0, seed).  k_pixel_sampler
// computes them once per pixel; k_raygen then starts a sample with two SipHashes less.
struct PixelSampler {
    u64 stream;       // hash_pixel(px, py)
    unsigned hash0;   // low word of hash_pixel_dim_seed(px, py, 0, seed) (stratified only)
};
YK_HD PixelSampler pixel_sampler(const SamplerCfg& c, unsigned px, unsigned py) {
    PixelSampler p;
    p.stream = hash_pixel(px, py);
    p.hash0 = c.kind == 0 ? 0u : (unsigned)hash_pixel_dim_seed(px, py, 0u, c.seed);
    return p;
}
// sampler_start(c, px, py, index, 0) followed by sampler_get_2d, from the pixel's values: same state, same draws
YK_HD SamplerState sampler_start_camera(const SamplerCfg& c, const PixelSampler& ps, unsigned px, unsigned py, unsigned index, float& ux, float& uy) {
    SamplerState s;
    s.px = px;
    s.py = py;
    s.sample_index = index;
    s.dimension = 2;
    s.rng = pcg_new(c.seed, ps.stream);
    pcg_advance_sample(s.rng, index, 0u);
    if (c.kind == 0) {
        ux = pcg_f32(s.rng);
        uy = pcg_f32(s.rng);
        return s;
    }
    const unsigned stratum = permutation_element(index, c.spp, ps.hash0);
    unsigned x, y;
    if (((c.nx & (c.nx - 1u)) | (c.ny & (c.ny - 1u))) == 0u) {
        x = stratum & (c.nx - 1u);
        y = stratum >> (31 - __builtin_clz(c.ny));
    } else {
        x = stratum % c.nx;
        y = stratum / c.ny;
    }
    const float dx = c.jitter ? pcg_f32(s.rng) : 0.5f;
    const float dy = c.jitter ? pcg_f32(s.rng) : 0.5f;
    ux = ((float)x + dx) / (float)c.nx;
    uy = ((float)y + dy) / (float)c.ny;
    return s;
}
YK_HD float sampler_get_1d(const SamplerCfg& c, SamplerState& s) {
    if (c.kind == 0) {
        s.dimension += 1;
        return pcg_f32(s.rng);
    }
    u64 hashed = hash_pixel_dim_seed(s.px, s.py, s.dimension, c.seed);
    unsigned stratum = permutation_element(s.sample_index, c.spp, (unsigned)hashed);
    s.dimension += 1;
    float delta = c.jitter ? pcg_f32(s.rng) : 0.5f;
    return ((float)stratum + delta) / (float)c.spp;
}
YK_HD void sampler_get_2d(const SamplerCfg& c, SamplerState& s, float& ux, float& uy) {
    if (c.kind == 0) {
        s.dimension += 2;
        ux = pcg_f32(s.rng);
        uy = pcg_f32(s.rng);
        return;
    }
    u64 hashed = hash_pixel_dim_seed(s.px, s.py, s.dimension, c.seed);
    unsigned stratum = permutation_element(s.sample_index, c.spp, (unsigned)hashed);
    s.dimension += 2;
    // x = stratum % nx ; y = stratum / ny (sic, stratified.rs:128); shifts when both
    // are powers of two (the usual 8x8 / 16x16), same values
    unsigned x, y;
    if (((c.nx & (c.nx - 1u)) | (c.ny & (c.ny - 1u))) == 0u) {
        x = stratum & (c.nx - 1u);
        y = stratum >> (31 - __builtin_clz(c.ny));
    } else {
        x = stratum % c.nx;
        y = stratum / c.ny;
    }
    float dx = c.jitter ? pcg_f32(s.rng) : 0.5f;
    float dy = c.jitter ? pcg_f32(s.rng) : 0.5f;
    ux = ((float)x + dx) / (float)c.nx;
    uy = ((float)y + dy) / (float)c.ny;
}

}  // namespace yk
