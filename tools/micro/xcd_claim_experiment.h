// xcd_claim_experiment.h — TIMING EXPERIMENT, not part of the product build (tools/build_variant.sh sortxcd -DYK_EXPERIMENT_SORT -DYK_EXPERIMENT_XCD;
// DESIGN.md §9, profiles/r03_ray_order_sweep.txt).  ChunkCursor::take's dynamic claim with the queue cut into eight ranges, one per group of
// blocks that share an XCD (blockIdx % 8, MI355X_MICROARCH.md): a wave claims from its own range (one atomic per claim, as the product does)
// and moves on to the next range when that one is drained.  `head` points at eight words (yk_render.cpp zeroes them before the launch).
// Expanded inside ChunkCursor::take (yk_trace.hip): uses its locals n, head, chunk and the members cur, end, exhausted, xcd_step.
#pragma once
#define YK_XCD_CLAIM \
                for (;;) { \
                    if (xcd_step >= 8u) { \
                        exhausted = true; \
                        return 0xffffffffu; \
                    } \
                    const unsigned y = ((blockIdx.x & 7u) + xcd_step) & 7u; \
                    const unsigned lo = (unsigned)((unsigned long long)n * y / 8ull), hi = (unsigned)((unsigned long long)n * (y + 1u) / 8ull); \
                    unsigned got = 0; \
                    if (lane_id() == 0) got = atomicAdd(head + y, chunk); \
                    got = (unsigned)__builtin_amdgcn_readfirstlane((int)got); \
                    if (got < hi - lo) { \
                        cur = lo + got; \
                        end = cur + chunk < hi ? cur + chunk : hi; \
                        break; \
                    } \
                    ++xcd_step; \
                } \
                goto claimed;
