// glibc_libm_check.cpp — the restatements of glibc's sinf / cosf / tanf / logf / expf / acosf / atanf / atan2f against the platform's functions:
// ALL 2^32 arguments of the one-argument functions; atan2f on every argument against 24 special partners (both positions) and on
// 2^32 random pairs (random bit patterns, and pairs whose ratio sweeps 2^-70 .. 2^70).  NaN results compared as a class.  CPU only.
//   the oracle's copy (oracle/olibm.h):
//     g++ -O2 -std=c++17 -ffp-contract=off -fno-builtin -I oracle tools/micro/glibc_libm_check.cpp -o /tmp/libm_check_oracle -lpthread -lm
//   the HOST instance of the product's copy (yuki_amd/csrc/yk_libm.h; its device instance is compared with the oracle on the GPU):
//     hipcc -x hip --cuda-host-only -DCHECK_PRODUCT -O2 -std=c++17 -ffp-contract=off -fno-builtin tools/micro/glibc_libm_check.cpp -o /tmp/libm_check_product -lpthread
//   run: /tmp/libm_check_oracle [stride [row mask]]      (stride 1 = exhaustive: 12 minutes on 8 cores, nearly all of it atan2f's partners)
#ifdef CHECK_PRODUCT
#include <hip/hip_runtime.h>
#endif
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <gnu/libc-version.h>
#include <thread>
#include <vector>

#ifdef CHECK_PRODUCT
#include "../../yuki_amd/csrc/yk_libm.h"
#define WHAT "host instance of yuki_amd/csrc/yk_libm.h"
static inline float r_sin(float x) { return yk::det_sinf(x); }
static inline float r_cos(float x) { return yk::det_cosf(x); }
static inline float r_tan(float x) { return yk::det_tanf(x); }
static inline float r_log(float x) { return yk::det_logf(x); }
static inline float r_acos(float x) { return yk::det_acosf(x); }
static inline float r_exp(float x) { return yk::det_expf(x); }
static inline float r_atan(float x) { return yk::gl_atanf(x); }
static inline float r_atan2(float y, float x) { return yk::det_atan2f(y, x); }
#define HAVE_PAIR 1
static inline void r_pair(float x, float& s, float& c) { yk::det_sincosf(x, s, c); }
#else
#include "olibm.h"
#define WHAT "oracle/olibm.h"
static inline float r_sin(float x) { return orc::lm::glibc::sinf(x); }
static inline float r_cos(float x) { return orc::lm::glibc::cosf(x); }
static inline float r_tan(float x) { return orc::lm::glibc::tanf(x); }
static inline float r_log(float x) { return orc::lm::glibc::logf(x); }
static inline float r_acos(float x) { return orc::lm::glibc::acosf(x); }
static inline float r_exp(float x) { return orc::lm::glibc::expf(x); }
static inline float r_atan(float x) { return orc::lm::glibc::atanf(x); }
static inline float r_atan2(float y, float x) { return orc::lm::glibc::atan2f(y, x); }
#endif

static inline uint32_t bits(float f) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return b;
}
static inline float fl(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline bool same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }

int main(int argc, char** argv) {
    const unsigned stride = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1u;
    const unsigned mask = argc > 2 ? (unsigned)std::strtoul(argv[2], nullptr, 0) : 0xffffffffu;  // bit per row of the table below
    const unsigned T = std::max(1u, std::thread::hardware_concurrency());
    enum { SIN, COS, TAN, LOG, EXP, ACOS, ATAN, ATAN2_SPECIAL, ATAN2_RANDOM, PAIR, NF };
    const char* names[NF] = {"sinf", "cosf", "tanf", "logf", "expf", "acosf", "atanf", "atan2f(special partners)", "atan2f(random pairs)", "sincos pair (shared reduction)"};
    std::atomic<uint64_t> bad[NF], done[NF];
    for (int i = 0; i < NF; ++i) bad[i] = 0, done[i] = 0;
    static const uint32_t special[24] = {0x00000000u, 0x80000000u, 0x3f800000u, 0xbf800000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x00000001u,
                                         0x80000001u, 0x007fffffu, 0x00800000u, 0x7f7fffffu, 0xff7fffffu, 0x3f000000u, 0x40000000u, 0x3f7fffffu,
                                         0x3f800001u, 0x5d800000u, 0x21800000u, 0xdd800000u, 0xa1800000u, 0x40490fdbu, 0x3eaaaaabu, 0xc0a00000u};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            uint64_t b[NF] = {0}, n[NF] = {0};
            uint64_t rng = 0x9E3779B97F4A7C15ull * (t + 1);
            auto next = [&rng]() {
                rng ^= rng << 13;
                rng ^= rng >> 7;
                rng ^= rng << 17;
                return rng;
            };
            auto report = [&](int f, uint32_t u, uint32_t v, float a, float c) {
                if (b[f] < 3) std::fprintf(stderr, "%s(%08x, %08x): glibc %08x restatement %08x\n", names[f], u, v, bits(a), bits(c));
                ++b[f];
            };
            for (uint64_t i = (uint64_t)t * stride; i < (1ull << 32); i += (uint64_t)T * stride) {
                const uint32_t u = (uint32_t)i;
                const float x = fl(u);
                float a, c;
#define ONE(F, HOST, MINE)                      \
    if (mask & (1u << F)) {                     \
        a = HOST(x);                            \
        c = MINE(x);                            \
        if (!same(a, c)) report(F, u, 0, a, c); \
        ++n[F];                                 \
    }
                ONE(SIN, ::sinf, r_sin)
                ONE(COS, ::cosf, r_cos)
                ONE(TAN, ::tanf, r_tan)
                ONE(LOG, ::logf, r_log)
                ONE(EXP, ::expf, r_exp)
                ONE(ACOS, ::acosf, r_acos)
                ONE(ATAN, ::atanf, r_atan)
#undef ONE
#ifdef HAVE_PAIR
                if (mask & (1u << PAIR)) {  // the pair the shading code calls: both results against the platform's sinf and cosf
                    float ps, pc;
                    r_pair(x, ps, pc);
                    a = ::sinf(x);
                    if (!same(a, ps)) report(PAIR, u, 0, a, ps);
                    a = ::cosf(x);
                    if (!same(a, pc)) report(PAIR, u, 1, a, pc);
                    ++n[PAIR];
                }
#endif
                const uint32_t sp = special[(i / stride) % 24];  // every argument meets every partner once per 24 strides; all of them when stride == 1 below
                for (int k = 0; (mask & (1u << ATAN2_SPECIAL)) && k < (stride == 1 ? 24 : 1); ++k) {
                    const uint32_t p = stride == 1 ? special[k] : sp;
                    a = ::atan2f(x, fl(p));
                    c = r_atan2(x, fl(p));
                    if (!same(a, c)) report(ATAN2_SPECIAL, u, p, a, c);
                    a = ::atan2f(fl(p), x);
                    c = r_atan2(fl(p), x);
                    if (!same(a, c)) report(ATAN2_SPECIAL, p, u, a, c);
                    n[ATAN2_SPECIAL] += 2;
                }
                if (!(mask & (1u << ATAN2_RANDOM))) continue;
                // a random pair: odd draws are two random bit patterns, even draws y = x * 2^e * m with e in -70..70
                const uint64_t r = next();
                uint32_t yu, xu;
                if (i & 1) {
                    yu = (uint32_t)r;
                    xu = (uint32_t)(r >> 32);
                } else {
                    xu = ((uint32_t)r & 0x807fffffu) | ((64u + (uint32_t)((r >> 40) % 128u)) << 23);  // finite, exponent 64..191
                    const int e = (int)((r >> 48) % 141u) - 70;
                    yu = ((uint32_t)(r >> 8) & 0x807fffffu) | ((((xu >> 23) & 0xffu) + (uint32_t)e) << 23);
                }
                a = ::atan2f(fl(yu), fl(xu));
                c = r_atan2(fl(yu), fl(xu));
                if (!same(a, c)) report(ATAN2_RANDOM, yu, xu, a, c);
                ++n[ATAN2_RANDOM];
            }
            for (int f = 0; f < NF; ++f) bad[f] += b[f], done[f] += n[f];
        });
    for (auto& x : th) x.join();
    uint64_t total = 0;
    std::printf("%s against glibc %s, stride %u:\n", WHAT, gnu_get_libc_version(), stride);
    for (int f = 0; f < NF; ++f) {
        std::printf("  %-26s %12llu arguments, %llu differ\n", names[f], (unsigned long long)done[f].load(), (unsigned long long)bad[f].load());
        total += bad[f].load();
    }
    return total ? 1 : 0;
}
