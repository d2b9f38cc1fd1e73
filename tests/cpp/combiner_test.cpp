// combiner_test.cpp — the host logic of yk_combiner.cpp (group commit of per-tile render calls) without a GPU: the file is compiled
// together with stand-ins for the three C-ABI functions it calls (yk_render_tiles, yk_render_tiles_accumulating, yk_last_error), which
// "render" a pattern that depends on (scene, pixel, sample) after a delay, poll the predicate like the library does, and record how
// they were called.  Built and run by tests/test_combiner.py (also under ThreadSanitizer: tools/asan/tsan_combiner.sh).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../yuki_amd/csrc/yk_internal.h"

static std::atomic<int> g_calls{0}, g_in_flight{0}, g_max_in_flight{0}, g_mixed{0}, g_cancelled_calls{0};
static std::atomic<long> g_tiles{0};
static std::atomic<int> g_delay_us{1500};

static float pattern(const yk_scene* scene, int x, int y, int c, int sample) {
    return (float)((reinterpret_cast<uintptr_t>(scene) & 0xff) * 1000003u % 977u) + (float)x * 0.25f + (float)y * 64.0f + (float)c * 0.125f + (float)sample * 4096.0f;
}

static yk_status fake_render(yk_context* ctx, const yk_scene* scene, const yk_tile* tiles, const uint16_t* samples, size_t n, float* out, yk_render_stats* stats,
                             yk_cancel_fn cancel, void* user) {
    (void)ctx;
    const int now = ++g_in_flight;
    int seen = g_max_in_flight.load();
    while (now > seen && !g_max_in_flight.compare_exchange_weak(seen, now)) {
    }
    ++g_calls;
    g_tiles += (long)n;
    yk_status st = YK_OK;
    for (int waited = 0; waited < g_delay_us.load(); waited += 100) {  // the library's wait: poll the predicate about every 100 us
        if (cancel && cancel(user)) {
            st = YK_ERR_CANCELLED;
            ++g_cancelled_calls;
            break;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    size_t off = 0;
    uint64_t rays = 0;
    for (size_t t = 0; t < n && st == YK_OK; ++t) {
        if (tiles[t].x0 == 9999) st = YK_ERR_DEVICE;  // a tile that "faults"
        for (int y = tiles[t].y0; y < tiles[t].y1; ++y)
            for (int x = tiles[t].x0; x < tiles[t].x1; ++x, ++off)
                for (int c = 0; c < 3; ++c) out[off * 3 + c] = pattern(scene, x, y, c, samples ? samples[t] : -1);
        rays += 7u * (uint64_t)(tiles[t].x1 - tiles[t].x0) * (uint64_t)(tiles[t].y1 - tiles[t].y0) + 3u;
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->rays = rays;
        stats->shadow_rays = 2 * rays;
        stats->samples = off;
        stats->seconds_total = 1e-3;
    }
    --g_in_flight;
    return st;
}

extern "C" {
yk_status yk_render_tiles(yk_context* ctx, const yk_scene* scene, const yk_camera*, const yk_sampler_desc*, const yk_integrator_desc*, const yk_tile* tiles, size_t n,
                          float* out, yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    return fake_render(ctx, scene, tiles, nullptr, n, out, stats, cancel, user);
}
yk_status yk_render_tiles_accumulating(yk_context* ctx, const yk_scene* scene, const yk_camera*, const yk_sampler_desc*, const yk_integrator_desc*, const yk_tile* tiles,
                                       const uint16_t* samples, size_t n, float* out, yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    return fake_render(ctx, scene, tiles, samples, n, out, stats, cancel, user);
}
yk_status yk_last_error(const yk_context*, char* buf, size_t cap) {
    std::snprintf(buf, cap, "stand-in error text");
    return YK_OK;
}
}

static int g_failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failures;                                                  \
        }                                                                  \
    } while (0)

struct Fire {
    std::atomic<int> polls{0};
    int after;  // answer non-zero from this poll on; < 0: never
    bool only_while_rendering = false;  // count polls only while a stand-in render call is in flight (the interruption must land inside a job)
};
static int fire_fn(void* u) {
    Fire* f = static_cast<Fire*>(u);
    if (f->only_while_rendering && g_in_flight.load() == 0) return 0;
    const int k = ++f->polls;
    return f->after >= 0 && k > f->after;
}

static bool tile_ok(const yk_scene* scene, const yk_tile& t, int sample, const std::vector<float>& px) {
    size_t off = 0;
    for (int y = t.y0; y < t.y1; ++y)
        for (int x = t.x0; x < t.x1; ++x, ++off)
            for (int c = 0; c < 3; ++c)
                if (px[off * 3 + c] != pattern(scene, x, y, c, sample)) return false;
    return true;
}

int main() {
    // contexts are only identities to the combiner (it reads `device`): never handed to HIP here
    std::vector<yk_context*> ctxs;
    for (int i = 0; i < 3; ++i) ctxs.push_back(new yk_context);
    const yk_scene* sceneA = reinterpret_cast<const yk_scene*>(uintptr_t(0x1010));
    const yk_scene* sceneB = reinterpret_cast<const yk_scene*>(uintptr_t(0x2027));
    yk_camera cam;
    yk_sampler_desc smp;
    yk_integrator_desc integ;
    std::memset(&cam, 0, sizeof cam);
    std::memset(&smp, 0, sizeof smp);
    std::memset(&integ, 0, sizeof integ);

    // ---- argument checks
    yk_combiner* c = nullptr;
    CHECK(yk_combiner_create(nullptr, 1, 0, 0, &c) == YK_ERR_INVALID_ARGUMENT);
    yk_context* twice[2] = {ctxs[0], ctxs[0]};
    CHECK(yk_combiner_create(twice, 2, 0, 0, &c) == YK_ERR_INVALID_ARGUMENT);
    CHECK(yk_combiner_create(ctxs.data(), 2, 0, 100, &c) == YK_OK && c);
    {
        yk_tile empty{4, 4, 4, 8};
        float px[3];
        CHECK(yk_combiner_render_tile(c, sceneA, &cam, &smp, &integ, &empty, -1, px, nullptr, nullptr, nullptr) == YK_ERR_INVALID_ARGUMENT);
        CHECK(yk_combiner_render_tile(c, nullptr, &cam, &smp, &integ, &empty, -1, px, nullptr, nullptr, nullptr) == YK_ERR_INVALID_ARGUMENT);
    }

    // ---- 12 workers x 30 tiles on 2 lanes, two scenes and both modes interleaved: right pixels, merged submissions, exact sums
    {
        const int T = 12, PER = 30;
        std::atomic<uint64_t> rays{0}, shadow{0}, samples{0}, expect_rays{0};
        std::atomic<int> bad{0};
        std::vector<std::thread> th;
        for (int k = 0; k < T; ++k)
            th.emplace_back([&, k] {
                for (int i = 0; i < PER; ++i) {
                    const yk_scene* sc = (k % 3 == 0) ? sceneB : sceneA;
                    const int sample = (k % 4 == 1) ? (i % 5) : -1;
                    yk_tile t{(uint16_t)(16 * i), (uint16_t)(16 * k), (uint16_t)(16 * i + 16 - (i % 3)), (uint16_t)(16 * k + 16 - (k % 2))};  // clipped tiles too
                    std::vector<float> px((size_t)(t.x1 - t.x0) * (t.y1 - t.y0) * 3, -1.0f);
                    yk_render_stats st;
                    if (yk_combiner_render_tile(c, sc, &cam, &smp, &integ, &t, sample, px.data(), &st, nullptr, nullptr) != YK_OK || !tile_ok(sc, t, sample, px)) ++bad;
                    rays += st.rays;
                    shadow += st.shadow_rays;
                    samples += st.samples;
                    expect_rays += 7u * (uint64_t)(t.x1 - t.x0) * (uint64_t)(t.y1 - t.y0);
                }
            });
        for (auto& t : th) t.join();
        yk_combiner_info info;
        CHECK(yk_combiner_get_info(c, &info) == YK_OK);
        CHECK(bad.load() == 0);
        CHECK(info.tiles == (uint64_t)T * PER && info.lanes == 2 && info.requeued == 0);
        CHECK(info.submissions == (uint64_t)g_calls.load() && info.submissions < info.tiles / 2);  // calls really were merged
        CHECK(info.largest_submission >= 3 && info.largest_submission <= 64);
        CHECK(g_max_in_flight.load() == 2);                                                        // both lanes used, never more
        CHECK(rays.load() == expect_rays.load() + 3u * info.tiles && shadow.load() == 2 * rays.load());  // counts exact in sum (the stand-in adds 3 per tile)
        std::printf("merge: %llu tiles in %llu submissions (largest %u), %d in flight at most\n", (unsigned long long)info.tiles, (unsigned long long)info.submissions,
                    info.largest_submission, g_max_in_flight.load());
    }
    yk_combiner_destroy(c);

    // ---- interruption: one worker's predicate fires inside a running submission; it alone returns CANCELLED, the others get their pixels
    {
        const int T = 6;
        // max_tiles = T with a long linger: the leader submits the moment all six have arrived — one submission holds them all, on any host
        CHECK(yk_combiner_create(ctxs.data(), 1, T, 5000000, &c) == YK_OK);
        g_delay_us = 300000;  // a job long enough for worker 2's polls to land inside it however slowly a loaded host schedules them
        std::vector<Fire> fires(T);
        for (int k = 0; k < T; ++k) fires[k].after = (k == 2) ? 5 : -1;  // worker 2 is told to stop at its sixth poll INSIDE the job
        fires[2].only_while_rendering = true;
        std::atomic<int> ready{0};
        std::vector<yk_status> res(T, YK_OK);
        std::vector<int> ok(T, 0);
        std::vector<std::thread> th;
        for (int k = 0; k < T; ++k)
            th.emplace_back([&, k] {
                yk_tile t{(uint16_t)(16 * k), 0, (uint16_t)(16 * k + 16), 16};
                std::vector<float> px(16 * 16 * 3, -1.0f);
                ++ready;
                while (ready.load() < T) std::this_thread::yield();  // all workers call together: one submission holds them all
                res[k] = yk_combiner_render_tile(c, sceneA, &cam, &smp, &integ, &t, -1, px.data(), nullptr, fire_fn, &fires[k]);
                ok[k] = tile_ok(sceneA, t, -1, px);
            });
        for (auto& t : th) t.join();
        yk_combiner_info info;
        (void)yk_combiner_get_info(c, &info);
        for (int k = 0; k < T; ++k) {
            CHECK(res[k] == (k == 2 ? YK_ERR_CANCELLED : YK_OK));
            if (k != 2) CHECK(ok[k] == 1);
            if (k == 2) CHECK(fires[k].polls.load() > 0);  // (the others' polls are counted too, but a leader that is never interrupted may finish first)
        }
        CHECK(g_cancelled_calls.load() >= 1 && info.requeued >= 1);
        std::printf("interrupt: %llu tiles queued again, %d submissions interrupted\n", (unsigned long long)info.requeued, g_cancelled_calls.load());
        // a predicate that is already true: the call comes back CANCELLED whether it leads or waits
        Fire now;
        now.after = 0;
        yk_tile t{0, 0, 16, 16};
        std::vector<float> px(16 * 16 * 3);
        CHECK(yk_combiner_render_tile(c, sceneA, &cam, &smp, &integ, &t, -1, px.data(), nullptr, fire_fn, &now) == YK_ERR_CANCELLED);
        yk_combiner_destroy(c);
        g_delay_us = 1500;
    }

    // ---- an error of the submission reaches every caller that was part of it, with the text
    {
        CHECK(yk_combiner_create(ctxs.data(), 1, 0, 3000, &c) == YK_OK);
        std::vector<yk_status> res(3, YK_OK);
        std::vector<std::thread> th;
        for (int k = 0; k < 3; ++k)
            th.emplace_back([&, k] {
                yk_tile t{(uint16_t)(k == 1 ? 9999 : 16 * k), 0, (uint16_t)(k == 1 ? 10015 : 16 * k + 16), 16};
                std::vector<float> px(16 * 16 * 3);
                res[k] = yk_combiner_render_tile(c, sceneA, &cam, &smp, &integ, &t, -1, px.data(), nullptr, nullptr, nullptr);
            });
        for (auto& t : th) t.join();
        int failed = 0;
        for (int k = 0; k < 3; ++k) failed += res[k] == YK_ERR_DEVICE;
        CHECK(res[1] == YK_ERR_DEVICE && failed >= 1);
        char buf[64];
        CHECK(yk_combiner_last_error(c, buf, sizeof buf) == YK_OK && std::strstr(buf, "stand-in") != nullptr);
        yk_combiner_destroy(c);
    }
    std::printf("%s\n", g_failures ? "combiner_test: FAILED" : "combiner_test: ok");
    return g_failures ? 1 : 0;
}
