//! Raw bindings for `include/yuki_hip.h` (ABI version 1).  Field order and types follow the
//! header line by line; see the header for the reference file:line each item replaces.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const YK_ABI_VERSION: u32 = 1;

pub type yk_status = c_int;
pub const YK_OK: yk_status = 0;
pub const YK_ERR_INVALID_ARGUMENT: yk_status = 1;
pub const YK_ERR_NO_DEVICE: yk_status = 2;
pub const YK_ERR_DEVICE: yk_status = 3;
pub const YK_ERR_OUT_OF_MEMORY: yk_status = 4;
pub const YK_ERR_UNSUPPORTED: yk_status = 5;
pub const YK_ERR_BVH_BUILD: yk_status = 6;
pub const YK_ERR_CANCELLED: yk_status = 7;
pub const YK_ERR_STACK_OVERFLOW: yk_status = 8;

pub const YK_MAT_MATTE: u32 = 0;
pub const YK_MAT_GLASS: u32 = 1;
pub const YK_MAT_METAL: u32 = 2;
pub const YK_MAT_GLOSSY: u32 = 3;
pub const YK_MAT_FLAG_REMAP: u32 = 1;
pub const YK_MAT_FLAG_TEXTURED_A: u32 = 2;
pub const YK_LIGHT_POINT: u32 = 0;
pub const YK_LIGHT_SPOT: u32 = 1;
pub const YK_LIGHT_DISTANT: u32 = 2;
pub const YK_LIGHT_RECT: u32 = 3;
pub const YK_SPLIT_SAH: u32 = 0;
pub const YK_SPLIT_MIDDLE: u32 = 1;
pub const YK_SPLIT_EQUAL_COUNTS: u32 = 2;
pub const YK_SAMPLER_UNIFORM: u32 = 0;
pub const YK_SAMPLER_STRATIFIED: u32 = 1;
pub const YK_INTEGRATOR_WHITTED: u32 = 0;
pub const YK_INTEGRATOR_PATH: u32 = 1;
pub const YK_INTEGRATOR_BVH_INTERSECTIONS: u32 = 2;
pub const YK_INTEGRATOR_GEOMETRY_NORMALS: u32 = 3;
pub const YK_INTEGRATOR_SHADING_NORMALS: u32 = 4;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct yk_mesh_desc {
    pub has_normals: u8,
    pub has_uvs: u8,
    pub swaps_handedness: u8,
    pub pad: u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct yk_sphere_desc {
    pub object_to_world: [f32; 16],
    pub world_to_object: [f32; 16],
    pub radius: f32,
    pub material: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct yk_material_desc {
    pub kind: u32,
    pub a: [f32; 3],
    pub b: [f32; 3],
    pub c: f32,
    pub flags: u32,
    pub a_texture: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct yk_texture_desc {
    pub width: u32,
    pub height: u32,
    pub rgb: *const f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct yk_light_desc {
    pub kind: u32,
    pub p: [f32; 3],
    pub i: [f32; 3],
    pub cos_total_width: f32,
    pub cos_falloff_start: f32,
    pub world_to_light: [f32; 16],
    pub sample_to_world: [f32; 16],
    pub sample_to_world_inv: [f32; 16],
    pub area: f32,
}

#[repr(C)]
pub struct yk_scene_desc {
    pub n_vertices: u32,
    pub points: *const f32,
    pub normals: *const f32,
    pub uvs: *const f32,
    pub n_triangles: u32,
    pub indices: *const u32,
    pub tri_mesh: *const u32,
    pub tri_material: *const i32,
    pub tri_area_light: *const i32,
    pub n_meshes: u32,
    pub meshes: *const yk_mesh_desc,
    pub n_spheres: u32,
    pub spheres: *const yk_sphere_desc,
    pub n_materials: u32,
    pub materials: *const yk_material_desc,
    pub n_lights: u32,
    pub lights: *const yk_light_desc,
    pub background: [f32; 3],
    pub split_method: u32,
    pub max_shapes_in_node: u32,
    pub shape_order: *const u32,
    pub n_textures: u32,
    pub textures: *const yk_texture_desc,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct yk_camera {
    pub camera_to_world: [f32; 16],
    pub camera_to_world_inv: [f32; 16],
    pub raster_to_camera: [f32; 16],
    pub raster_to_camera_inv: [f32; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct yk_camera_params {
    pub position: [f32; 3],
    pub target: [f32; 3],
    pub up: [f32; 3],
    pub fov_axis: u32,
    pub fov_degrees: f32,
    pub res_x: u16,
    pub res_y: u16,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct yk_sampler_desc {
    pub kind: u32,
    pub nx: u32,
    pub ny: u32,
    pub jitter: u32,
    pub seed: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct yk_integrator_desc {
    pub kind: u32,
    pub max_depth: u32,
    pub has_clamp: u32,
    pub indirect_clamp: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct yk_tile {
    pub x0: u16,
    pub y0: u16,
    pub x1: u16,
    pub y1: u16,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct yk_bvh_node {
    pub bmin: [f32; 3],
    pub bmax: [f32; 3],
    pub a: u32,
    pub count: u16,
    pub axis: u8,
    pub is_leaf: u8,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct yk_scene_info {
    pub n_nodes: u64,
    pub n_interior: u64,
    pub n_shapes: u64,
    pub bounds_min: [f32; 3],
    pub bounds_max: [f32; 3],
    pub build_seconds: f64,
    pub upload_seconds: f64,
    pub device_bytes: u64,
    pub max_leaf_shapes: u32,
    pub tree_depth: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct yk_combiner_info {
    pub submissions: u64,
    pub tiles: u64,
    pub requeued: u64,
    pub largest_submission: u32,
    pub lanes: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct yk_render_stats {
    pub rays: u64,
    pub shadow_rays: u64,
    pub samples: u64,
    pub seconds_total: f64,
    pub seconds_trace: f64,
    pub seconds_shadow: f64,
    pub seconds_shade: f64,
    pub trace_launches: u32,
    pub batches: u32,
    pub shadow_launches: u32,
    pub reserved: u32,
}

pub enum yk_context {}
pub enum yk_scene {}
pub enum yk_loaded_scene {}
pub enum yk_tile_list {}
pub const YK_MULTI_SHARED_DEVICES: u32 = 1;
pub const YK_MULTI_PEER_COPY: u32 = 2;
pub enum yk_multi {}
pub enum yk_multi_scene {}
pub enum yk_multi_film {}
pub enum yk_dist {}
pub enum yk_combiner {}
pub type yk_cancel_fn = Option<unsafe extern "C" fn(user: *mut c_void) -> c_int>;

extern "C" {
    pub fn yk_abi_version() -> u32;
    pub fn yk_status_string(s: yk_status) -> *const c_char;
    pub fn yk_context_create(device: c_int, out: *mut *mut yk_context) -> yk_status;
    pub fn yk_context_destroy(ctx: *mut yk_context);
    pub fn yk_last_error(ctx: *const yk_context, buf: *mut c_char, cap: usize) -> yk_status;
    pub fn yk_context_stream(ctx: *const yk_context) -> *mut c_void;
    pub fn yk_context_set_option(ctx: *mut yk_context, key: *const c_char, value: i64) -> yk_status;
    pub fn yk_camera_init(params: *const yk_camera_params, out: *mut yk_camera) -> yk_status;
    pub fn yk_film_tiles(res_x: u16, res_y: u16, tile_dim: u16, out: *mut yk_tile, cap: usize) -> usize;
    pub fn yk_make_rect_light(light_to_world: *const f32, light_to_world_inv: *const f32, radiance: *const f32, size: *const f32, out: *mut yk_light_desc) -> yk_status;
    pub fn yk_make_spot_light(light_to_world: *const f32, light_to_world_inv: *const f32, intensity: *const f32, total_width_degrees: f32, falloff_start_degrees: f32, out: *mut yk_light_desc) -> yk_status;
    pub fn yk_make_point_light(light_to_world: *const f32, intensity: *const f32, out: *mut yk_light_desc) -> yk_status;
    pub fn yk_film_update_tiles(tiles: *const yk_tile, n_tiles: usize, tile_rgb: *const f32, res_x: u16, res_y: u16, film_rgb: *mut f32) -> yk_status;
    pub fn yk_scene_create(ctx: *mut yk_context, desc: *const yk_scene_desc, out: *mut *mut yk_scene) -> yk_status;
    pub fn yk_scene_destroy(scene: *mut yk_scene);
    pub fn yk_scene_get_info(scene: *const yk_scene, out: *mut yk_scene_info) -> yk_status;
    pub fn yk_scene_export_bvh(scene: *const yk_scene, nodes: *mut yk_bvh_node, shape_order: *mut u32) -> yk_status;
    pub fn yk_render_tiles(ctx: *mut yk_context, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, tiles: *const yk_tile, n_tiles: usize, out_rgb: *mut f32, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_render_tiles_device(ctx: *mut yk_context, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, tiles: *const yk_tile, n_tiles: usize, d_out_rgb: *mut c_void, stream: *mut c_void, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_render_tiles_accumulating(ctx: *mut yk_context, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, tiles: *const yk_tile, tile_samples: *const u16, n_tiles: usize, out_rgb: *mut f32, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_render_tiles_accumulating_device(ctx: *mut yk_context, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, tiles: *const yk_tile, tile_samples: *const u16, n_tiles: usize, d_out_rgb: *mut c_void, stream: *mut c_void, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_film_accumulate_tiles(tiles: *const yk_tile, n_tiles: usize, tile_rgb: *const f32, res_x: u16, res_y: u16, film_rgb: *mut f32, tile_sample_counts: *mut u32) -> yk_status;
    pub fn yk_film_accumulate_tiles_device(ctx: *mut yk_context, tiles: *const yk_tile, n_tiles: usize, d_tile_rgb: *const c_void, res_x: u16, res_y: u16, d_film_rgb: *mut c_void, stream: *mut c_void) -> yk_status;
    pub fn yk_tile_list_create(ctx: *mut yk_context, tiles: *const yk_tile, tile_samples: *const u16, n_tiles: usize, out: *mut *mut yk_tile_list) -> yk_status;
    pub fn yk_tile_list_destroy(list: *mut yk_tile_list);
    pub fn yk_render_tile_list_device(ctx: *mut yk_context, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, list: *const yk_tile_list, d_out_rgb: *mut c_void, stream: *mut c_void, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_film_update_tile_list_device(ctx: *mut yk_context, list: *const yk_tile_list, d_tile_rgb: *const c_void, res_x: u16, res_y: u16, d_film_rgb: *mut c_void, stream: *mut c_void, accumulate: c_int) -> yk_status;
    pub fn yk_render_tiles_accumulating_passes(ctx: *mut yk_context, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, tiles: *const yk_tile, tile_samples: *const u16, n_tiles: usize, n_passes: u32, out_rgb: *mut f32, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_render_tile_list_samples_device(ctx: *mut yk_context, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, list: *const yk_tile_list, first_sample: u32, n_passes: u32, d_out_rgb: *mut c_void, stream: *mut c_void, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_context_interrupt(ctx: *mut yk_context) -> yk_status;
    pub fn yk_combiner_create(contexts: *const *mut yk_context, n_contexts: u32, max_tiles: u32, linger_us: u32, out: *mut *mut yk_combiner) -> yk_status;
    pub fn yk_combiner_destroy(combiner: *mut yk_combiner);
    pub fn yk_combiner_render_tile(combiner: *mut yk_combiner, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, tile: *const yk_tile, accumulating_sample: i32, tile_pixels: *mut f32, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_combiner_get_info(combiner: *const yk_combiner, out: *mut yk_combiner_info) -> yk_status;
    pub fn yk_combiner_last_error(combiner: *const yk_combiner, buf: *mut c_char, cap: usize) -> yk_status;
    pub fn yk_render_tile_list_passes_device(ctx: *mut yk_context, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, list: *const yk_tile_list, n_passes: u32, d_out_rgb: *mut c_void, stream: *mut c_void, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_film_accumulate_tile_list_passes_device(ctx: *mut yk_context, list: *const yk_tile_list, d_passes_rgb: *const c_void, n_passes: u32, res_x: u16, res_y: u16, d_film_rgb: *mut c_void, stream: *mut c_void) -> yk_status;
    pub fn yk_write_exr(path: *const c_char, width: u32, height: u32, rgb: *const f32) -> yk_status;
    pub fn yk_write_pfm(path: *const c_char, width: u32, height: u32, rgb: *const f32) -> yk_status;
    pub fn yk_render_tile(ctx: *mut yk_context, scene: *const yk_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, tile: *const yk_tile, tile_pixels: *mut f32, out_rays: *mut u64) -> yk_status;
    pub fn yk_film_update_tiles_device(ctx: *mut yk_context, tiles: *const yk_tile, n_tiles: usize, d_tile_rgb: *const c_void, res_x: u16, res_y: u16, d_film_rgb: *mut c_void, stream: *mut c_void) -> yk_status;
    pub fn yk_li(ctx: *mut yk_context, scene: *const yk_scene, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, n: usize, ray_o: *const f32, ray_d: *const f32, pixel_xy: *const u16, sample_index: *const u32, dimension: u32, out_li: *mut f32, out_ray_counts: *mut u32) -> yk_status;
    pub fn yk_image_texture_load(path: *const c_char, out: *mut yk_texture_desc) -> yk_status;
    pub fn yk_image_texture_free(tex: *mut yk_texture_desc);
    pub fn yk_load_ply(path: *const c_char, split_method: u32, max_shapes_in_node: u32, out: *mut *mut yk_loaded_scene) -> yk_status;
    pub fn yk_load_pbrt(path: *const c_char, split_method: u32, max_shapes_in_node: u32, out: *mut *mut yk_loaded_scene) -> yk_status;
    pub fn yk_loaded_scene_get(loaded: *const yk_loaded_scene, desc: *mut yk_scene_desc, camera: *mut yk_camera_params, tile_dim: *mut u16) -> yk_status;
    pub fn yk_loaded_scene_destroy(loaded: *mut yk_loaded_scene);
    pub fn yk_loader_last_error() -> *const c_char;
    // several GPUs of one process (RenderManager's role for GPU workers) and one process per GPU
    pub fn yk_multi_create(devices: *const c_int, n_devices: u32, out: *mut *mut yk_multi) -> yk_status;
    pub fn yk_multi_create_ex(devices: *const c_int, n_devices: u32, flags: u32, out: *mut *mut yk_multi) -> yk_status;
    pub fn yk_multi_deal(res_x: u16, res_y: u16, tile_dim: u16, n_ranks: u32, rank: u32, out: *mut yk_tile, cap: usize, out_pixels: *mut u64) -> usize;
    pub fn yk_multi_destroy(m: *mut yk_multi);
    pub fn yk_multi_device_count(m: *const yk_multi) -> u32;
    pub fn yk_multi_context(m: *mut yk_multi, rank: u32) -> *mut yk_context;
    pub fn yk_multi_set_option(m: *mut yk_multi, key: *const c_char, value: i64) -> yk_status;
    pub fn yk_multi_last_error(m: *const yk_multi, buf: *mut c_char, cap: usize) -> yk_status;
    pub fn yk_multi_scene_create(m: *mut yk_multi, desc: *const yk_scene_desc, out: *mut *mut yk_multi_scene) -> yk_status;
    pub fn yk_multi_scene_destroy(scene: *mut yk_multi_scene);
    pub fn yk_multi_scene_get_info(scene: *const yk_multi_scene, out: *mut yk_scene_info) -> yk_status;
    pub fn yk_multi_film_create(m: *mut yk_multi, res_x: u16, res_y: u16, tile_dim: u16, out: *mut *mut yk_multi_film) -> yk_status;
    pub fn yk_multi_film_destroy(film: *mut yk_multi_film);
    pub fn yk_multi_film_device_ptr(film: *const yk_multi_film) -> *mut c_void;
    pub fn yk_multi_render_film(m: *mut yk_multi, scene: *const yk_multi_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, film: *mut yk_multi_film, film_rgb: *mut f32, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_multi_accumulate_film(m: *mut yk_multi, scene: *const yk_multi_scene, camera: *const yk_camera, sampler: *const yk_sampler_desc, integrator: *const yk_integrator_desc, film: *mut yk_multi_film, first_sample: u32, n_passes: u32, film_rgb: *mut f32, stats: *mut yk_render_stats, cancel: yk_cancel_fn, user: *mut c_void) -> yk_status;
    pub fn yk_multi_film_clear(m: *mut yk_multi, film: *mut yk_multi_film) -> yk_status;
    pub fn yk_multi_interrupt(m: *mut yk_multi) -> yk_status;
    pub fn yk_multi_sync(m: *mut yk_multi) -> yk_status;
    pub fn yk_dist_unique_id(id: *mut u8) -> yk_status;
    pub fn yk_dist_create(ctx: *mut yk_context, id: *const u8, rank: u32, world: u32, out: *mut *mut yk_dist) -> yk_status;
    pub fn yk_dist_destroy(dist: *mut yk_dist);
    pub fn yk_dist_gather(dist: *mut yk_dist, d_send: *const c_void, d_recv: *mut c_void, count: usize, stream: *mut c_void) -> yk_status;
}

/// `yk_last_error` as a `String` (empty when none).
pub fn combiner_last_error(combiner: *const yk_combiner) -> String {
    let mut buf = vec![0u8; 512];
    unsafe { yk_combiner_last_error(combiner, buf.as_mut_ptr() as *mut c_char, buf.len()) };
    let n = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
    String::from_utf8_lossy(&buf[..n]).into_owned()
}

pub fn last_error(ctx: *const yk_context) -> String {
    let mut buf = vec![0u8; 512];
    unsafe { yk_last_error(ctx, buf.as_mut_ptr() as *mut c_char, buf.len()) };
    let n = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
    String::from_utf8_lossy(&buf[..n]).into_owned()
}
