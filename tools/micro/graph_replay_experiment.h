// graph_replay_experiment.h — TIMING EXPERIMENT, not part of the product build (tools/build_variant.sh graph -DYK_EXPERIMENT_GRAPH;
// tools/graph_tile_bench.py, profiles/r03_graph_tile.txt): what a hipGraph of a small job's ~40 dependent launches returns.
// YK_GRAPH_REPLAY=N (read per call: the tool sets it after a plain warm-up call): the job's launches are captured into a graph once and the
// graph is launched N times; the time of one replay goes to stderr.  Needs a job that enqueues without host synchronisation (one tile, or a
// prepared list) whose buffers exist already.  Expanded inside render_tiles_impl (yk_render.cpp): uses its locals ctx, st, stats, cancel, kt, ev0.
#pragma once
#define YK_GRAPH_BEGIN \
    const int graph_replay = std::getenv("YK_GRAPH_REPLAY") ? std::atoi(std::getenv("YK_GRAPH_REPLAY")) : 0; /* read per call: the tool sets it after a plain warm-up call */ \
    const bool capturing = graph_replay > 0 && stats != nullptr && !cancel; \
    if (capturing) { \
        kt.on = false; \
        HIP_TRY(ctx, hipStreamSynchronize(st)); \
        HIP_TRY(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); \
    }

#define YK_GRAPH_END \
    if (capturing) { \
        hipGraph_t graph = nullptr; \
        hipGraphExec_t exec = nullptr; \
        HIP_TRY(ctx, hipStreamEndCapture(st, &graph)); \
        HIP_TRY(ctx, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)); \
        HIP_TRY(ctx, hipGraphLaunch(exec, st)); /* warm */ \
        HIP_TRY(ctx, hipStreamSynchronize(st)); \
        hipEvent_t g0, g1; \
        HIP_TRY(ctx, hipEventCreate(&g0)); \
        HIP_TRY(ctx, hipEventCreate(&g1)); \
        const double w0 = now_seconds(); \
        HIP_TRY(ctx, hipEventRecord(g0, st)); \
        for (int k = 0; k < graph_replay; ++k) HIP_TRY(ctx, hipGraphLaunch(exec, st)); \
        HIP_TRY(ctx, hipEventRecord(g1, st)); \
        HIP_TRY(ctx, hipStreamSynchronize(st)); \
        const double w1 = now_seconds(); \
        float gms = 0.0f; \
        (void)hipEventElapsedTime(&gms, g0, g1); \
        std::fprintf(stderr, "graph replay: %d launches of the captured job, %.4f ms each on the device, %.4f ms each on the host clock\n", graph_replay, gms / graph_replay, \
                     (w1 - w0) * 1e3 / graph_replay); \
        (void)hipEventDestroy(g0); \
        (void)hipEventDestroy(g1); \
        (void)hipGraphExecDestroy(exec); \
        (void)hipGraphDestroy(graph); \
        HIP_TRY(ctx, hipEventRecord(ev0, st)); \
    }
