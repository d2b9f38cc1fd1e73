// yk_combiner.cpp — many render workers, one device: the reference's calling pattern without changing its render manager.
//
// The reference renders with `num_cpus - 1` worker threads (render_manager.rs:78-97), each popping one 16x16 tile from the queue and
// calling Integrator::render for it (render_worker.rs:205-256).  A tile is 256 pixels; the device wants 10^5..10^7 camera samples in
// flight.  The combiner sits under that call: a worker's yk_combiner_render_tile blocks, the calls that are waiting at the same time
// are merged into ONE yk_render_tiles submission (group commit: the first waiter leads, the others follow), and every caller gets its
// own tile back — the same bits a single-tile call returns, because a pixel sample depends on (seed, pixel, sample index) only
// (uniform.rs:72-84) and a tile's slab is written by its own samples.  With L contexts ("lanes") up to L submissions are in flight.
//
// Interruption: every caller keeps polling ITS OWN early_termination_predicate from ITS OWN thread (about every 100 us, like a
// synchronous yk_render_tiles does) — the reference's predicate consumes a channel message (render_worker.rs:240-249), so it must
// fire in the worker it belongs to.  A caller whose predicate fires while its tile is still queued leaves at once; if the tile is
// part of a running submission, that submission is interrupted, callers whose predicate fired return YK_ERR_CANCELLED and the
// others' tiles are queued again (they are never handed pixels of an interrupted job).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "yk_internal.h"

namespace {

struct Key {
    const yk_scene* scene;
    yk_camera camera;
    yk_sampler_desc sampler;
    yk_integrator_desc integrator;
    bool accumulating;
    bool operator==(const Key& o) const {
        return scene == o.scene && accumulating == o.accumulating && std::memcmp(&camera, &o.camera, sizeof camera) == 0 &&
               std::memcmp(&sampler, &o.sampler, sizeof sampler) == 0 && std::memcmp(&integrator, &o.integrator, sizeof integrator) == 0;
    }
};

struct Batch {
    std::atomic<int> interrupt{0};  // raised by a member whose predicate fired
};

struct Request {
    Key key;
    yk_tile tile;
    uint16_t sample = 0;
    float* out = nullptr;
    yk_render_stats* stats = nullptr;
    yk_cancel_fn cancel = nullptr;
    void* user = nullptr;
    // state, under the combiner's mutex
    enum { QUEUED, RUNNING, DONE } state = QUEUED;
    Batch* batch = nullptr;
    bool fired = false;  // this caller's predicate has answered non-zero
    yk_status status = YK_OK;
    size_t area() const { return (size_t)(tile.x1 - tile.x0) * (size_t)(tile.y1 - tile.y0); }
};

}  // namespace

struct yk_combiner {
    std::vector<yk_context*> lanes;
    std::vector<char> lane_busy;
    uint32_t max_tiles = 64;
    uint32_t linger_us = 100;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Request*> queue;
    std::string last_error;
    // counters (yk_combiner_get_info)
    uint64_t submissions = 0, tiles = 0, requeued = 0;
    uint32_t largest = 0;
};

namespace {

struct LeaderPoll {
    Request* self;
    Batch* batch;
};

// the predicate the leader hands to yk_render_tiles: its own caller's, or a member's verdict
int leader_predicate(void* p) {
    LeaderPoll* lp = static_cast<LeaderPoll*>(p);
    if (lp->batch->interrupt.load(std::memory_order_acquire)) return 1;
    if (lp->self->cancel && !lp->self->fired && lp->self->cancel(lp->self->user)) {
        lp->self->fired = true;  // only this thread touches its own `fired` while the request is RUNNING
        lp->batch->interrupt.store(1, std::memory_order_release);
        return 1;
    }
    return 0;
}

void run_batch(yk_combiner* c, yk_context* ctx, const std::vector<Request*>& members, Request* self, Batch* batch, yk_status& st, std::string& err,
               yk_render_stats& stats, std::vector<float>& pixels) {
    const Key& k = members[0]->key;
    std::vector<yk_tile> tiles(members.size());
    std::vector<uint16_t> samples(members.size());
    size_t total = 0;
    for (size_t i = 0; i < members.size(); ++i) {
        tiles[i] = members[i]->tile;
        samples[i] = members[i]->sample;
        total += members[i]->area();
    }
    pixels.resize(total * 3);
    LeaderPoll lp{self, batch};
    std::memset(&stats, 0, sizeof stats);
    if (k.accumulating)
        st = yk_render_tiles_accumulating(ctx, k.scene, &k.camera, &k.sampler, &k.integrator, tiles.data(), samples.data(), tiles.size(), pixels.data(), &stats,
                                          leader_predicate, &lp);
    else
        st = yk_render_tiles(ctx, k.scene, &k.camera, &k.sampler, &k.integrator, tiles.data(), tiles.size(), pixels.data(), &stats, leader_predicate, &lp);
    if (st != YK_OK) {
        char buf[512] = {0};
        (void)yk_last_error(ctx, buf, sizeof buf);
        err = buf;
    }
    (void)c;
}

}  // namespace

extern "C" {

yk_status yk_combiner_create(yk_context* const* contexts, uint32_t n_contexts, uint32_t max_tiles, uint32_t linger_us, yk_combiner** out) try {
    if (!contexts || n_contexts == 0 || n_contexts > 16 || !out) return YK_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < n_contexts; ++i) {
        if (!contexts[i]) return YK_ERR_INVALID_ARGUMENT;
        for (uint32_t j = 0; j < i; ++j)
            if (contexts[j] == contexts[i]) return YK_ERR_INVALID_ARGUMENT;  // a lane per context: the same one twice would serialise on its lock
        if (contexts[i]->device != contexts[0]->device) return YK_ERR_INVALID_ARGUMENT;  // callers pass ONE scene, which lives on one device
    }
    yk_combiner* c = new yk_combiner;
    c->lanes.assign(contexts, contexts + n_contexts);
    c->lane_busy.assign(n_contexts, 0);
    c->max_tiles = max_tiles ? max_tiles : 64;
    c->linger_us = linger_us;
    *out = c;
    return YK_OK;
} catch (const std::bad_alloc&) {
    return YK_ERR_OUT_OF_MEMORY;
}

void yk_combiner_destroy(yk_combiner* c) { delete c; }

yk_status yk_combiner_last_error(const yk_combiner* c, char* buf, size_t cap) {
    if (!c || !buf || cap == 0) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(const_cast<yk_combiner*>(c)->m);
    std::snprintf(buf, cap, "%s", c->last_error.c_str());
    return YK_OK;
}

yk_status yk_combiner_get_info(const yk_combiner* c, yk_combiner_info* out) {
    if (!c || !out) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(const_cast<yk_combiner*>(c)->m);
    out->submissions = c->submissions;
    out->tiles = c->tiles;
    out->requeued = c->requeued;
    out->largest_submission = c->largest;
    out->lanes = (uint32_t)c->lanes.size();
    return YK_OK;
}

yk_status yk_combiner_render_tile(yk_combiner* c, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                  const yk_integrator_desc* integrator, const yk_tile* tile, int32_t accumulating_sample, float* tile_pixels,
                                  yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    if (!c) return YK_ERR_INVALID_ARGUMENT;
    if (!scene || !camera || !sampler || !integrator || !tile || !tile_pixels || accumulating_sample > 0xFFFF || tile->x1 <= tile->x0 || tile->y1 <= tile->y0) {
        std::lock_guard<std::mutex> lk(c->m);
        c->last_error = "null argument, empty tile or sample index above u16";
        return YK_ERR_INVALID_ARGUMENT;
    }
    Request r;
    std::memset(&r.key, 0, sizeof r.key);  // padding bytes take part in the comparison
    r.key.scene = scene;
    std::memcpy(&r.key.camera, camera, sizeof *camera);
    std::memcpy(&r.key.sampler, sampler, sizeof *sampler);
    std::memcpy(&r.key.integrator, integrator, sizeof *integrator);
    r.key.accumulating = accumulating_sample >= 0;
    r.tile = *tile;
    r.sample = (uint16_t)(accumulating_sample >= 0 ? accumulating_sample : 0);
    r.out = tile_pixels;
    r.stats = stats;
    r.cancel = cancel;
    r.user = user;

    std::unique_lock<std::mutex> lk(c->m);
    try {
        c->queue.push_back(&r);
    } catch (const std::bad_alloc&) {
        return YK_ERR_OUT_OF_MEMORY;
    }
    c->cv.notify_all();
    const auto poll = std::chrono::microseconds(100);
    for (;;) {
        if (r.state == Request::DONE) return r.status;
        // lead a submission when a lane is free and this request is still waiting
        int lane = -1;
        if (r.state == Request::QUEUED)
            for (size_t i = 0; i < c->lanes.size(); ++i)
                if (!c->lane_busy[i]) {
                    lane = (int)i;
                    break;
                }
        if (lane >= 0) {
            c->lane_busy[lane] = 1;  // claimed before lingering: two waiters must not both linger for the same lane
            if (c->queue.size() < 2 && c->linger_us) {
                // alone: give the other workers a moment to arrive (they are between update_tile and their next pop)
                c->cv.wait_for(lk, std::chrono::microseconds(c->linger_us), [&] { return c->queue.size() >= c->max_tiles; });
            }
            if (r.state != Request::QUEUED) {  // another leader took this request while this thread lingered
                c->lane_busy[lane] = 0;
                c->cv.notify_all();
                continue;
            }
            // the submission: this request and every queued one that asks for the same job, oldest first
            std::vector<Request*> members;
            Batch batch;
            try {
                members.reserve(std::min<size_t>(c->max_tiles, c->queue.size()));
            } catch (const std::bad_alloc&) {  // nothing has changed hands yet: leave the queue as a caller that never came
                c->lane_busy[lane] = 0;
                c->queue.erase(std::find(c->queue.begin(), c->queue.end(), &r));
                c->cv.notify_all();
                return YK_ERR_OUT_OF_MEMORY;
            }
            members.push_back(&r);
            for (Request* q : c->queue)
                if (q != &r && members.size() < c->max_tiles && q->state == Request::QUEUED && q->key == r.key) members.push_back(q);
            for (Request* q : members) {
                q->state = Request::RUNNING;
                q->batch = &batch;
                c->queue.erase(std::find(c->queue.begin(), c->queue.end(), q));
            }
            yk_context* ctx = c->lanes[lane];
            lk.unlock();
            yk_status st = YK_OK;
            std::string err;
            yk_render_stats bs;
            std::vector<float> pixels;
            try {
                run_batch(c, ctx, members, &r, &batch, st, err, bs, pixels);
            } catch (const std::bad_alloc&) {
                st = YK_ERR_OUT_OF_MEMORY;
                err = "host allocation failed";
            }
            if (st == YK_OK) {  // hand every caller its slab (tile-major, each tile row-major: the tile_pixels layout) before anyone is woken
                size_t off = 0;
                for (Request* q : members) {
                    std::memcpy(q->out, pixels.data() + off * 3, q->area() * 3 * sizeof(float));
                    off += q->area();
                }
            }
            lk.lock();
            c->lane_busy[lane] = 0;
            c->submissions += 1;
            c->tiles += members.size();
            c->largest = std::max<uint32_t>(c->largest, (uint32_t)members.size());
            size_t total_area = 0;
            for (Request* q : members) total_area += q->area();
            uint64_t rays_left = bs.rays, shadow_left = bs.shadow_rays, samples_left = bs.samples;
            for (size_t i = members.size(); i-- > 0;) {
                Request* q = members[i];
                q->batch = nullptr;
                if (st == YK_ERR_CANCELLED && !q->fired) {  // somebody else's interruption: this tile is rendered again
                    bool back = true;
                    try {
                        c->queue.push_front(q);
                    } catch (const std::bad_alloc&) {
                        back = false;
                    }
                    if (back) {
                        q->state = Request::QUEUED;
                        c->requeued += 1;
                        continue;
                    }
                    st = YK_ERR_OUT_OF_MEMORY;  // (this and the remaining members report it)
                    err = "host allocation failed";
                }
                q->state = Request::DONE;
                q->status = st;
                if (st == YK_OK && q->stats) {
                    // counts of a submission are not kept per tile: they are shared out by area, the remainder to the leader — exact in sum,
                    // which is what the reference does with them (render_manager.rs:277-281, window.rs:911-916)
                    *q->stats = bs;
                    const bool leader = (i == 0);
                    q->stats->rays = leader ? rays_left : bs.rays * q->area() / total_area;
                    q->stats->shadow_rays = leader ? shadow_left : bs.shadow_rays * q->area() / total_area;
                    q->stats->samples = leader ? samples_left : bs.samples * q->area() / total_area;
                    rays_left -= leader ? 0 : q->stats->rays;
                    shadow_left -= leader ? 0 : q->stats->shadow_rays;
                    samples_left -= leader ? 0 : q->stats->samples;
                }
            }
            if (st != YK_OK && st != YK_ERR_CANCELLED) c->last_error = err;
            c->cv.notify_all();
            continue;
        }
        // follow: wait for the leader (or for a lane), polling this caller's own predicate from this thread
        if (cancel && !r.fired) {
            lk.unlock();
            const bool now = cancel(user) != 0;
            lk.lock();
            if (now) {
                r.fired = true;
                if (r.state == Request::QUEUED) {
                    c->queue.erase(std::find(c->queue.begin(), c->queue.end(), &r));
                    return YK_ERR_CANCELLED;
                }
                if (r.state == Request::RUNNING && r.batch) r.batch->interrupt.store(1, std::memory_order_release);
            }
            if (r.state == Request::DONE) return r.status;
        } else if (r.fired && r.state == Request::QUEUED) {  // fired while running, and the submission ended some other way before seeing it
            c->queue.erase(std::find(c->queue.begin(), c->queue.end(), &r));
            return YK_ERR_CANCELLED;
        }
        if (cancel)
            c->cv.wait_for(lk, poll);
        else
            c->cv.wait(lk);
    }
}

}  // extern "C"
