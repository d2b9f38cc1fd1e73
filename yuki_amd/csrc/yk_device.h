// yk_device.h — device-resident data layout of the wavefront Path integrator.
//
// HBM layout (all 16-byte records so every lane moves one dwordx4 and a wave one
// contiguous KiB; "SoA of float4"):
//
//   path state, ping-pong (compacted every bounce by `shade`):
//     rayO[i] = (o.x, o.y, o.z, bits(flags))        flags: bounces | specular<<8
//     rayD[i] = (d.x, d.y, d.z, bits(sample_id))    sample_id = index into sample_buf
//     thru[i] = (beta.r, beta.g, beta.b, bits(sampler dimension))
//     rngs[i] = (state.lo, state.hi, inc.lo, inc.hi)          PCG32 stream of the pixel
//     (camera bounce of a Path render traced by the packet kernel: rayO and thru are NOT stored — one origin for all,
//      throughput one; see YK_CTRL_CAM_O)
//   per bounce, not carried:
//     hit[i]                       source triangle or -1             (trace -> shade)
//     pend[i] = (rgb, kind << 29 | sample slot in the batch)  emission / background term   (shade -> accumulate)
//     shO/shD/shC[i*n_lights+l]    NEE shadow ray + its contribution (shade -> shadow -> accumulate)
//     vis[i*VS+l]                  0 none, 1 pending/visible, 2 occluded; VS = YK_VIS_STRIDE(n_lights): 4 bytes per path for up to 4 lights
//     shq[k]                       compacted list of pending shadow slots
//   per chunk:
//     sample_buf[sample_id] = (L.r, L.g, L.b, -)    radiance of one camera sample (Path: first written by k_accumulate of the camera bounce)
//     pixel_xy[pixel]       = x | y<<16             pixel of each chunk-local pixel index
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yuki_hip.h"
#include "yk_bsdf.h"
#include "yk_rng.h"

namespace yk {

// BVH interior node, 64 B: both children's AABBs (exact f32 copies of the
// reference's 32-byte nodes' bounds, bvh.rs:536-556) + child references.
//   q0 = (c0.min.xyz, c0.max.x) q1 = (c0.max.yz, c1.min.xy) q2 = (c1.min.z, c1.max.xyz)
//   q3 = (ref0, ref1 | split_axis << 28, 0, 0); ref: bit31 set -> leaf, low 28 bits = first
//        primitive, else index of an interior node (bit 30: in DevScene::top_nodes)
struct DevNode {
    float4 q0, q1, q2;
    uint4 q3;
};
static_assert(sizeof(DevNode) == 64, "DevNode must be 64 bytes");

// stride of a path's verdict bytes in vis[]: up to four lights share one aligned word (one store in k_shade, one load in k_accumulate)
#define YK_VIS_STRIDE(nl) ((nl) <= 4u ? 4u : (nl))
#define YK_LEAF_BIT 0x80000000u
#define YK_REF_NONE 0xffffffffu
#define YK_AXIS_SHIFT 28
#define YK_AXIS_MASK (3u << YK_AXIS_SHIFT)
#define YK_REF_INDEX_MAX ((1u << YK_AXIS_SHIFT) - 1u)
// ref bit 30: index into DevScene::top_nodes (the first levels of the tree, which the
// traversal kernels keep in LDS) instead of DevScene::nodes
#define YK_TOP_BIT 0x40000000u
#define YK_TOP_MAX 1023

// 128-byte (one cache line) 4-wide node: a reference interior node P collapsed with its two
// children A = P+1 and B = second child.  Slots 0,1 = A's children (or A itself + NONE when
// A is a leaf), slots 2,3 = B's.  Box i = (q[..]) packed like DevNode, twice.  The three
// split axes give the reference's visiting order for a ray: B's group first when the
// direction is negative along P's axis, and within a group the second child first when it is
// negative along that child's axis (bvh.rs:184-194 applied at both levels).
//   q0..q2 = boxes of slots 0,1   q3..q5 = boxes of slots 2,3
//   q6 = refs (leaf bit | first primitive, index of a DevNode4, or YK_REF_NONE)
//   q7 = (axis_P | axis_A << 2 | axis_B << 4, 0, 0, 0)
struct DevNode4 {
    float4 q0, q1, q2, q3, q4, q5;
    uint4 q6, q7;
};
static_assert(sizeof(DevNode4) == 128, "DevNode4 must be 128 bytes");

// Light record (lights/*.rs).  n of a rectangular light is constant
// (sample_to_world * Normal(0,-1,0), rectangular_light.rs:48) and precomputed.
struct DevLight {
    uint32_t kind;
    float p[3];
    float i[3];
    float cos_total_width, cos_falloff_start;
    float w2l[16];  // spot: world_to_light.m
    float s2w[16];  // rect: sample_to_world.m
    float n[3];     // rect: transformed normal
    float area;
};

// Sphere record (shapes/sphere.rs:14-35): object_to_world (matrix + inverse);
// world_to_object is the same pair swapped.
struct DevSphere {
    float o2w[16];
    float w2o[16];
    float radius;
    int material;
    unsigned swaps_handedness;
    unsigned pad;
};

// primitive flag bits in tris[3p+2].w; bits 4-6: the primitive's BSDF kind (MK_*), which the render-loop traversal
// kernels copy into the hit word so that k_shade can sort a block's paths by kind without a dependent fetch
#define YK_PRIM_LAST 1u
#define YK_PRIM_SPHERE 2u
#define YK_PRIM_KIND_SHIFT 4
// hit[i] of the render loop: -1 = miss, else leaf-order slot (< 2^28, YK_REF_INDEX_MAX) | BSDF kind << 28
#define YK_HIT_KIND_SHIFT 28
#define YK_HIT_PRIM_MASK 0x0fffffffu
#define YK_HIT_WORD(prim, pflags) ((int)((prim) | ((((pflags) >> YK_PRIM_KIND_SHIFT) & 7u) << YK_HIT_KIND_SHIFT)))

// mesh flag bits
#define YK_MESH_NORMALS 1u
#define YK_MESH_UVS 2u
#define YK_MESH_SWAPS 4u

struct DevScene {
    const DevNode* nodes;
    const DevNode4* nodes4;  // 4-wide collapse of the same tree (null: use `nodes`), root = entry 0
    const DevNode* top_nodes;  // breadth-first copy of the first n_top interior nodes; child refs carry YK_TOP_BIT inside the set
    uint32_t n_top;
    const DevNode* top_nodes_any;  // the same for the any-hit kernel, which has room for more (4-byte stack entries): its own set, refs consistent within it
    uint32_t n_top_any;
    const float4* tris;  // 3 per primitive in leaf order: (p0, bits(area_light)) (p1, bits(source shape)) (p2, bits(YK_PRIM_*))
    const uint4* prim_shade;   // per primitive in leaf order: (i0, i1, i2, material << 6 | material kind (MK_*) << 3 | YK_MESH_* flags)
    // per primitive in leaf order, 4 x float4: (n0, uv0.x) (n1, uv0.y) (n2, uv1.x) (uv1.y, uv2.x, uv2.y, 0) — the vertex normals and uvs
    // k_shade needs, addressed by the hit alone; null when no mesh has normals or uvs
    const float4* prim_attr;
    const DevSphere* spheres;  // source shape s >= n_triangles is spheres[s - n_triangles]
    uint32_t n_triangles;
    uint32_t root_ref;
    float root_bmin[3], root_bmax[3];
    // shading data, indexed by source triangle
    const uint32_t* indices;
    const float* points;
    const float* normals;
    const float* uvs;
    const uint32_t* tri_mesh;
    const int32_t* tri_material;
    const int32_t* tri_area_light;
    const uint32_t* mesh_flags;
    const Material* materials;
    const DevLight* lights;
    uint32_t n_lights;
    float background[3];
    // image textures: texels of all textures back to back (rgb, 0), per-texture (first texel, width, height, 0)
    const float4* texels;
    const uint4* tex_info;
};

struct DevCamera {
    float c2w[16];  // camera_to_world.m
    float r2c[16];  // raster_to_camera.m
};

struct PathBuffers {
    float4* rayO;
    float4* rayD;
    float4* thru;
    uint4* rngs;
};

// control block, zeroed ONCE per batch (no per-bounce resets: every bounce has its own words):
// [3] error flags; bounce b owns the 8 words at YK_CTRL_BOUNCE(b): +0 paths entering the bounce
// (written by raygen / by shade of bounce b-1), +1 / +2 shadow queue lengths (area / delta
// lights), +3..5 work-queue heads of its closest-hit and any-hit launches.  The per-stage entry
// points (yk_trace_closest / yk_trace_any) use word 0 as the ray count and YK_CTRL_HEADS as head.
#define YK_CTRL_WORDS 1024
#define YK_CTRL_ERR 3
#define YK_CTRL_CANCELLED 2  // in the context's error block (yk_internal.h): CancelRef::dev
// A float4 BEHIND the zeroed words (its own cache line: the words above are hammered by queue atomics): the camera's ray
// origin, written by raygen.  The camera bounce of a Path render whose rays go to the wave-packet kernel is "lean": every
// ray starts there with throughput one, so raygen stores neither rayO nor thru and the packet kernel, k_shade and
// k_accumulate of that bounce take the constants instead of reading 48 bytes per path.
#define YK_CTRL_CAM_O (YK_CTRL_WORDS + 32)
#define YK_CTRL_ALLOC_WORDS (YK_CTRL_WORDS + 64)
#define YK_CTRL_HEADS 8
#define YK_CTRL_STRIDE 8
#define YK_CTRL_BOUNCE(b) (8 + YK_CTRL_STRIDE * (b))
#define YK_CTRL_SHQ 1   // shadow rays towards area lights (scattered directions)
#define YK_CTRL_SHQ2 2  // shadow rays towards point / spot / distant lights (one target: coherent)
#define YK_CTRL_HEAD 3  // + {0: closest, 1: any, 2: any (delta queue)}
#define YK_CTRL_MAX_DEPTH ((YK_CTRL_WORDS - 8) / YK_CTRL_STRIDE - 1)

// Interruption (integrators/mod.rs:153: the reference polls its predicate once per pixel sample).  Two words: `host` lives in
// pinned host memory and is written by the host thread that polls the predicate; `dev` is a word in device memory that EVERY
// queue-driven kernel reads when it starts — every block of it: k_shade's and k_accumulate's blocks live for a window or two of a
// long launch — and then treats its queue as empty.  Who carries the
// news from `host` to `dev`: (1) ONE wave of every persistent traversal launch — the first wave of block 0 — reads `host`
// whenever it claims work (a PCIe round trip: every wave doing so halved the traversal rate), raises `dev` and poisons the
// launch's queue head so that no wave claims again; (2) the host, with a 4-byte copy on a stream of its own.  Both words only ever
// go from 0 to 1 during a submission, so a later kernel never sees less than an earlier one did: nothing reads what an
// interrupted kernel left unwritten.  Null pointers: no interruption (per-stage entry points).
struct CancelRef {
    const unsigned* host;
    unsigned* dev;
};
#define YK_HEAD_POISON 0xC0000000u  // above any queue length (< 2^31); 2^30 below the wrap, a wave claims at most once more

// A kernel's look at `dev` when it starts: an ordinary (cached, scalar) load like the one of its queue length — whatever was
// written before the launch began is visible to it.  (An agent-scope atomic load here, one per block of k_shade's 65,536, cost
// 4 ms of 30 per frame: it goes to memory and the block's first barrier waits for it.)  Blocks that start later during a long
// grid-stride launch usually see a word raised meanwhile as well; nothing depends on that.
__device__ __forceinline__ bool cancel_raised(const CancelRef& c) { return c.dev && *c.dev != 0u; }
// the relay wave's look at the host's word (one lane): raises `dev`, poisons `head`
__device__ __forceinline__ bool cancel_relay(const CancelRef& c, unsigned* head) {
    if (!c.host || __hip_atomic_load(c.host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) return false;
    atomicOr(c.dev, 1u);
    atomicMax(head, YK_HEAD_POISON);
    return true;
}

struct RenderParams {
    SamplerCfg sampler;
    uint32_t max_depth;
    uint32_t has_clamp;
    float clamp;
    uint32_t integrator;  // yk_integrator_kind
    // Samples rendered per entry of the call's pixel table: sample id = entry * spe + k.  Plain
    // film: spe = spp and sample index = k.  With a sample-index table (accumulating film,
    // yk_li) the sample index is table[entry] + k; spe = the number of passes rendered at once.
    uint32_t spe;
    uint32_t sample_base;  // sample index of an entry when there is no table: 0 for the plain film, FilmTile.sample shared by all tiles otherwise
    CancelRef cancel;
};

// sample id -> (entry of the pixel table, sample within the entry): id = entry * spe + k.  `spe` is uniform and nearly always a
// power of two (8x8, 16x16 strata; 1 for yk_li): a shift and a mask then, the division otherwise.
__device__ __forceinline__ void split_sample_id(uint32_t sid, uint32_t spe, uint32_t& entry, uint32_t& k) {
    if ((spe & (spe - 1u)) == 0u) {
        entry = sid >> (31u - (uint32_t)__clz((int)spe));
        k = sid & (spe - 1u);
    } else {
        entry = sid / spe;
        k = sid % spe;
    }
}

}  // namespace yk
