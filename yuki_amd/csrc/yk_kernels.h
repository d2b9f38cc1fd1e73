// yk_kernels.h — host-callable launchers of the gfx950 kernels (yk_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "yk_device.h"

namespace yk {

unsigned trace_block_size();
unsigned trace_spill_depth();
unsigned trace_top_nodes();
unsigned trace_top_nodes_any();
// wave-packet traversal for coherent rays (yk_packet.hip); requires tree depth <= 64
unsigned packet_blocks_per_cu();
// rayO == nullptr: every ray starts at *lean_origin (the lean camera bounce, yk_device.h)
void launch_trace_closest_packet(hipStream_t s, unsigned grid, const DevScene& sc, const float4* rayO, const float4* rayD, const unsigned* count_ptr,
                                 unsigned* head, int* hit_tri, unsigned long long* ray_counter, const float4* lean_origin = nullptr, CancelRef cancel = CancelRef{nullptr, nullptr});
void launch_trace_any_packet(hipStream_t s, unsigned grid, const DevScene& sc, const float4* shO, const float4* shD, const unsigned* slot_of,
                             const unsigned* count_ptr, unsigned* head, unsigned char* vis, unsigned long long* shadow_counter, CancelRef cancel = CancelRef{nullptr, nullptr});
unsigned trace_blocks_per_cu();

void launch_pixel_table(hipStream_t s, const yk_tile* tiles, const uint32_t* tile_offset, uint32_t n_tiles, uint32_t n_pixels, uint32_t* pixel_xy,
                        const uint16_t* tile_sample = nullptr, uint32_t* pixel_sample = nullptr);
void launch_pixel_table_one(hipStream_t s, const yk_tile& tile, uint32_t n_pixels, uint32_t* pixel_xy, uint32_t tile_sample, uint32_t* pixel_sample);
void launch_raygen(hipStream_t s, const DevCamera& cam, const RenderParams& prm, const uint32_t* pixel_xy, const uint32_t* pixel_sample, uint64_t work0,
                   uint32_t n, PathBuffers out, float4* sample_buf, unsigned* count, float4* lean_origin = nullptr, const uint4* pixel_aux = nullptr);
void launch_pixel_sampler(hipStream_t s, const SamplerCfg& cfg, const uint32_t* pixel_xy, uint32_t n_pixels, uint4* pixel_aux);
void launch_raygen_user(hipStream_t s, const RenderParams& prm, const float* o, const float* d, const uint16_t* pixel, const uint32_t* sample_index,
                        uint32_t dimension, uint32_t n, PathBuffers out, float4* sample_buf, uint32_t* pixel_xy, unsigned* ctrl);
void launch_trace_closest(hipStream_t s, unsigned grid, const DevScene& sc, const float4* rayO, const float4* rayD, const float* t_max_opt,
                          const unsigned* count_ptr, unsigned* head, int* hit_tri, float4* hit_out, uint4* stats_out, uint2* spill,
                          unsigned spill_stride, unsigned* ctrl, unsigned long long* ray_counter, const unsigned* cancel_host = nullptr);
void launch_whitted(hipStream_t s, unsigned grid, const DevScene& sc, const RenderParams& prm, const uint32_t* pixel_xy, const uint32_t* sample_index_tab,
                    PathBuffers cur, uint32_t n, float4* sample_buf, uint2* spill, unsigned spill_stride, unsigned* ctrl, unsigned long long* counters);
unsigned whitted_max_depth();
void launch_trace_any(hipStream_t s, unsigned grid, const DevScene& sc, const float4* shO, const float4* shD, const unsigned* slot_of,
                      const unsigned* count_ptr, unsigned* head, unsigned char* vis, uint2* spill, unsigned spill_stride, unsigned* ctrl,
                      unsigned long long* shadow_counter, const unsigned* cancel_host = nullptr);
void launch_shade(hipStream_t s, unsigned grid, const DevScene& sc, const RenderParams& prm, const uint32_t* pixel_xy, const uint32_t* sample_index_tab,
                  PathBuffers cur, PathBuffers nxt, const int* hit_tri, float4* pend, float4* shO, float4* shD, float4* shC, unsigned char* vis,
                  unsigned* shq, float4* shO2, float4* shD2, unsigned* shq2, unsigned* bc, unsigned split_delta, unsigned reorder, unsigned block_slots, unsigned sid_base, const float4* lean_origin);
void launch_accumulate(hipStream_t s, unsigned grid, const RenderParams& prm, PathBuffers cur, const float4* pend, const float4* shC,
                       const unsigned char* vis, unsigned nl, float4* sample_buf, const unsigned* bc, unsigned first, unsigned sid_base);
void launch_resolve(hipStream_t s, const float4* sample_buf, uint32_t n_pixels, uint32_t spp, float* out_rgb);
void launch_resolve_passes(hipStream_t s, const float4* sample_buf, uint32_t n_pixels, uint32_t n_passes, float* out_rgb, size_t pass_stride);
void launch_film_scatter(hipStream_t s, const uint32_t* pixel_xy, uint32_t n_pixels, const float* tile_rgb, uint32_t res_x, float* film_rgb, int accumulate,
                         uint32_t n_passes = 1, size_t pass_stride = 0);
void launch_debug_shade(hipStream_t s, const DevScene& sc, uint32_t integrator, PathBuffers cur, const int* hit_tri, const uint4* stats, uint32_t n,
                        float4* sample_buf);
void launch_device_math(hipStream_t s, int fn, size_t n, const float* a, const float* b, float* out);
void launch_sampler_sequence(hipStream_t s, const SamplerCfg& cfg, uint32_t px, uint32_t py, uint32_t sample_index, const uint8_t* dims, size_t n_draws,
                             float* out);
void launch_light_test(hipStream_t s, const DevLight& L, int index, size_t n, const float* p, const float* ng, const float* u, float* out);
void launch_bsdf_test(hipStream_t s, const Material& m, size_t n, const float* ng, const float* ns, const float* dpdu, const float* wo,
                      const float* wi_or_u, int sample, float* out);
void launch_pack_rays(hipStream_t s, size_t n, const float* o, const float* d, float4* rayO, float4* rayD);
void launch_pack_shadow_rays(hipStream_t s, size_t n, const float* o, const float* d, const float* t_max, const int* area_light, float4* shO,
                             float4* shD);
void launch_unpack_rays(hipStream_t s, size_t n, const float4* rayO, const float4* rayD, float* o, float* d);

}  // namespace yk
