//! `yuki/src/integrators/hip_path.rs` — the Path integrator on an MI355X through
//! libyuki_hip.so.  SOURCE ONLY (never compiled: no Rust toolchain in the build image).
//!
//! Upstream wiring:
//!   integrators/mod.rs:33-40   add `HipPath(path::Params)` and `HipWhitted(whitted::Params)` to `IntegratorType` (+ Display/EnumString)
//!   integrators/mod.rs:43-52   `IntegratorType::HipPath(p) => Box::new(HipPath::new(p, hip_device()))`,
//!                              `IntegratorType::HipWhitted(p) => Box::new(HipPath::whitted(p, hip_device()))`
//!   renderer/render_manager.rs build the `HipDevice` when a scene is loaded (see gpu_worker.rs)
#![cfg(feature = "hip")]

use super::{path, whitted, Integrator, RadianceResult};
use crate::{
    camera::Camera,
    describe::{LightDesc, MaterialDesc, SamplerDesc, ShapeDesc, TextureDesc},
    film::FilmTile,
    lights::AreaLight,
    materials::Material,
    math::{Ray, Spectrum, Transform},
    sampling::Sampler,
    scene::{Scene, SplitMethod},
    shapes::Mesh,
};
use allocators::ScopedScratch;
use std::{collections::HashMap, os::raw::c_void, ptr, sync::Arc};
use yuki_hip_sys as sys;

fn flat(m: &crate::math::Matrix4x4<f32>) -> [f32; 16] {
    let mut o = [0.0; 16];
    for r in 0..4 {
        for c in 0..4 {
            o[4 * r + c] = m.row(r)[c];
        }
    }
    o
}

fn key<T: ?Sized>(a: &Arc<T>) -> usize {
    Arc::as_ptr(a) as *const () as usize
}

/// One HIP context + the uploaded scene, shared by every `HipPath` instance — plus a second context and the
/// combiner that merges the render workers' per-tile calls (two submissions in flight: yk_combiner, "lanes").
pub struct HipDevice {
    pub ctx: *mut sys::yk_context,
    pub scene: *mut sys::yk_scene,
    lane2: *mut sys::yk_context,
    pub combiner: *mut sys::yk_combiner,
}
// yk_render_tiles* serialise per context inside the library; the combiner is built for concurrent callers
unsafe impl Send for HipDevice {}
unsafe impl Sync for HipDevice {}

impl Drop for HipDevice {
    fn drop(&mut self) {
        unsafe {
            sys::yk_combiner_destroy(self.combiner);
            sys::yk_scene_destroy(self.scene);
            sys::yk_context_destroy(self.lane2);
            sys::yk_context_destroy(self.ctx);
        }
    }
}

impl HipDevice {
    /// One device: flatten the scene, create a context there, upload.
    pub fn new(scene: &Scene, device: i32) -> Result<Self, String> {
        flatten_scene(scene, |desc| unsafe {
            let mut ctx = ptr::null_mut();
            let st = sys::yk_context_create(device, &mut ctx);
            if st != sys::YK_OK {
                return Err(format!("yk_context_create: status {st}"));
            }
            let mut scn = ptr::null_mut();
            let st = sys::yk_scene_create(ctx, desc, &mut scn);
            if st != sys::YK_OK {
                let why = sys::last_error(ctx);
                sys::yk_context_destroy(ctx);
                return Err(format!("yk_scene_create: {why}"));
            }
            // the workers' lanes: this context and one more on the same device (a scene serves every context of its device)
            let mut lane2 = ptr::null_mut();
            let mut combiner = ptr::null_mut();
            let mut st = sys::yk_context_create(device, &mut lane2);
            if st == sys::YK_OK {
                let lanes = [ctx, lane2];
                st = sys::yk_combiner_create(lanes.as_ptr(), 2, 0, 100, &mut combiner);
            }
            if st != sys::YK_OK {
                sys::yk_scene_destroy(scn);
                if !lane2.is_null() {
                    sys::yk_context_destroy(lane2);
                }
                sys::yk_context_destroy(ctx);
                return Err(format!("yk_combiner_create: status {st}"));
            }
            Ok(Self { ctx, scene: scn, lane2, combiner })
        })
    }
}

/// Flattens `Scene` (scene/mod.rs:41-49) into a `yk_scene_desc` that lives for the duration of `f`
/// (the description borrows the vectors built here; the library copies what it keeps).
/// `scene.shapes` is already in BVH leaf order (bvh.rs:96 returns the reordered vec), and
/// the library rebuilds the same hierarchy from `shape_order` = identity over that order,
/// so the device BVH visits primitives exactly like `scene.bvh`.
pub(crate) fn flatten_scene<R>(scene: &Scene, f: impl FnOnce(&sys::yk_scene_desc) -> R) -> R {
    {
        let mut mesh_ids: HashMap<usize, u32> = HashMap::new();
        let mut mat_ids: HashMap<usize, i32> = HashMap::new();
        let mut light_ids: HashMap<usize, i32> = HashMap::new();
        let (mut points, mut normals, mut uvs) = (Vec::<f32>::new(), Vec::<f32>::new(), Vec::<f32>::new());
        let mut mesh_base = Vec::<u32>::new();
        let mut meshes = Vec::<sys::yk_mesh_desc>::new();
        let (mut any_normals, mut any_uvs) = (false, false);
        let mut register_mesh = |m: &Arc<Mesh>| -> u32 {
            *mesh_ids.entry(key(m)).or_insert_with(|| {
                mesh_base.push((points.len() / 3) as u32);
                for p in &m.points {
                    points.extend_from_slice(&[p.x, p.y, p.z]);
                }
                for i in 0..m.points.len() {
                    match m.normals.get(i) {
                        Some(n) => normals.extend_from_slice(&[n.x, n.y, n.z]),
                        None => normals.extend_from_slice(&[0.0; 3]),
                    }
                    match m.uvs.get(i) {
                        Some(t) => uvs.extend_from_slice(&[t.x, t.y]),
                        None => uvs.extend_from_slice(&[0.0; 2]),
                    }
                }
                any_normals |= !m.normals.is_empty();
                any_uvs |= !m.uvs.is_empty();
                meshes.push(sys::yk_mesh_desc {
                    has_normals: !m.normals.is_empty() as u8,
                    has_uvs: !m.uvs.is_empty() as u8,
                    swaps_handedness: m.transform_swaps_handedness as u8,
                    pad: 0,
                });
                (meshes.len() - 1) as u32
            })
        };

        // lights first: area lights are looked up by the triangles that carry them
        let mut lights = Vec::<sys::yk_light_desc>::new();
        for l in &scene.lights {
            let mut d: sys::yk_light_desc = unsafe { std::mem::zeroed() };
            match l.describe() {
                LightDesc::Point { p, i } => {
                    d.kind = sys::YK_LIGHT_POINT;
                    d.p = [p.x, p.y, p.z];
                    d.i = [i.r, i.g, i.b];
                }
                LightDesc::Spot { world_to_light, p, i, cos_total_width, cos_falloff_start } => {
                    d.kind = sys::YK_LIGHT_SPOT;
                    d.p = [p.x, p.y, p.z];
                    d.i = [i.r, i.g, i.b];
                    d.cos_total_width = cos_total_width;
                    d.cos_falloff_start = cos_falloff_start;
                    d.world_to_light = flat(world_to_light.m());
                }
                LightDesc::Distant { w, radiance } => {
                    d.kind = sys::YK_LIGHT_DISTANT;
                    d.p = [w.x, w.y, w.z];
                    d.i = [radiance.r, radiance.g, radiance.b];
                }
                LightDesc::Rectangular { sample_to_world, l, area } => {
                    d.kind = sys::YK_LIGHT_RECT;
                    d.i = [l.r, l.g, l.b];
                    d.sample_to_world = flat(sample_to_world.m());
                    d.sample_to_world_inv = flat(sample_to_world.m_inv());
                    d.area = area;
                }
            }
            // `Arc<dyn Light>` and the `Arc<dyn AreaLight>` held by triangles are the same allocation
            light_ids.insert(key(l), lights.len() as i32);
            lights.push(d);
        }

        let mut textures = Vec::<sys::yk_texture_desc>::new();
        let mut texture_store = Vec::<Vec<f32>>::new();
        let mut materials = Vec::<sys::yk_material_desc>::new();
        let mut register_material = |m: &Arc<dyn Material>| -> Result<i32, String> {
            if let Some(&id) = mat_ids.get(&key(m)) {
                return Ok(id);
            }
            fn constant<T: Copy>(t: &TextureDesc<T>, what: &str) -> Result<T, String> {
                match t {
                    TextureDesc::Constant(v) => Ok(*v),
                    TextureDesc::Image { .. } => Err(format!("HIP path: image texture on {what} is not supported")),
                }
            }
            let mut d = sys::yk_material_desc::default();
            match m.describe() {
                MaterialDesc::Matte { kd, sigma } => {
                    d.kind = sys::YK_MAT_MATTE;
                    d.c = constant(&sigma, "matte sigma")?;
                    match kd {
                        TextureDesc::Constant(v) => d.a = [v.r, v.g, v.b],
                        TextureDesc::Image { data, width, height } => {
                            let mut rgb = Vec::with_capacity(3 * data.len());
                            for s in data {
                                rgb.extend_from_slice(&[s.r, s.g, s.b]);
                            }
                            texture_store.push(rgb);
                            textures.push(sys::yk_texture_desc { width: width as u32, height: height as u32, rgb: ptr::null() });
                            d.flags |= sys::YK_MAT_FLAG_TEXTURED_A;
                            d.a_texture = (textures.len() - 1) as u32;
                        }
                    }
                }
                MaterialDesc::Glass { r, t, eta } => {
                    let (r, t) = (constant(&r, "glass R")?, constant(&t, "glass T")?);
                    d.kind = sys::YK_MAT_GLASS;
                    d.a = [r.r, r.g, r.b];
                    d.b = [t.r, t.g, t.b];
                    d.c = eta;
                }
                MaterialDesc::Metal { eta, k, roughness, remap_roughness } => {
                    let (eta, k) = (constant(&eta, "metal eta")?, constant(&k, "metal k")?);
                    d.kind = sys::YK_MAT_METAL;
                    d.a = [eta.r, eta.g, eta.b];
                    d.b = [k.r, k.g, k.b];
                    d.c = constant(&roughness, "metal roughness")?;
                    d.flags = remap_roughness as u32;
                }
                MaterialDesc::Glossy { rs, roughness, remap_roughness } => {
                    let rs = constant(&rs, "glossy Rs")?;
                    d.kind = sys::YK_MAT_GLOSSY;
                    d.a = [rs.r, rs.g, rs.b];
                    d.c = constant(&roughness, "glossy roughness")?;
                    d.flags = remap_roughness as u32;
                }
            }
            materials.push(d);
            mat_ids.insert(key(m), (materials.len() - 1) as i32);
            Ok((materials.len() - 1) as i32)
        };

        let (mut indices, mut tri_mesh, mut tri_material, mut tri_area_light) = (Vec::<u32>::new(), Vec::<u32>::new(), Vec::<i32>::new(), Vec::<i32>::new());
        let mut spheres = Vec::<sys::yk_sphere_desc>::new();
        let mut order = Vec::<(bool, u32)>::new(); // (is_sphere, id) in scene.shapes order
        for s in scene.shapes.iter() {
            match s.describe() {
                ShapeDesc::Triangle { mesh, vertices, material, area_light } => {
                    let mid = register_mesh(mesh);
                    let base = mesh_base[mid as usize];
                    order.push((false, tri_mesh.len() as u32));
                    indices.extend(vertices.iter().map(|&v| base + v as u32));
                    tri_mesh.push(mid);
                    tri_material.push(register_material(material)?);
                    tri_area_light.push(area_light.map_or(-1, |a| *light_ids.get(&key(a)).unwrap_or(&-1)));
                }
                ShapeDesc::Sphere { object_to_world, world_to_object, radius, material } => {
                    order.push((true, spheres.len() as u32));
                    spheres.push(sys::yk_sphere_desc {
                        object_to_world: flat(object_to_world.m()),
                        world_to_object: flat(world_to_object.m()),
                        radius,
                        material: register_material(material)?,
                    });
                }
            }
        }
        let nt = tri_mesh.len() as u32;
        let shape_order: Vec<u32> = order.iter().map(|&(sph, id)| if sph { nt + id } else { id }).collect();
        for (t, store) in textures.iter_mut().zip(texture_store.iter()) {
            t.rgb = store.as_ptr();
        }

        let desc = sys::yk_scene_desc {
            n_vertices: (points.len() / 3) as u32,
            points: points.as_ptr(),
            normals: if any_normals { normals.as_ptr() } else { ptr::null() },
            uvs: if any_uvs { uvs.as_ptr() } else { ptr::null() },
            n_triangles: nt,
            indices: indices.as_ptr(),
            tri_mesh: tri_mesh.as_ptr(),
            tri_material: tri_material.as_ptr(),
            tri_area_light: tri_area_light.as_ptr(),
            n_meshes: meshes.len() as u32,
            meshes: meshes.as_ptr(),
            n_spheres: spheres.len() as u32,
            spheres: spheres.as_ptr(),
            n_materials: materials.len() as u32,
            materials: materials.as_ptr(),
            n_lights: lights.len() as u32,
            lights: lights.as_ptr(),
            background: [scene.background.r, scene.background.g, scene.background.b],
            split_method: match scene.load_settings.split_method {
                SplitMethod::SurfaceAreaHeuristic => sys::YK_SPLIT_SAH,
                SplitMethod::Middle => sys::YK_SPLIT_MIDDLE,
                SplitMethod::EqualCounts => sys::YK_SPLIT_EQUAL_COUNTS,
            },
            max_shapes_in_node: scene.load_settings.max_shapes_in_node as u32,
            shape_order: shape_order.as_ptr(),
            n_textures: textures.len() as u32,
            textures: textures.as_ptr(),
        };
        f(&desc)
    }
}

pub(crate) fn camera_desc(camera: &Camera) -> sys::yk_camera {
    let (c2w, r2c): (&Transform<f32>, &Transform<f32>) = camera.transforms();
    sys::yk_camera {
        camera_to_world: flat(c2w.m()),
        camera_to_world_inv: flat(c2w.m_inv()),
        raster_to_camera: flat(r2c.m()),
        raster_to_camera_inv: flat(r2c.m_inv()),
    }
}

pub(crate) fn sampler_desc(sampler: &dyn Sampler) -> sys::yk_sampler_desc {
    match sampler.describe() {
        SamplerDesc::Uniform { pixel_samples, rng_seed } => sys::yk_sampler_desc { kind: sys::YK_SAMPLER_UNIFORM, nx: pixel_samples, ny: 1, jitter: 1, seed: rng_seed },
        SamplerDesc::Stratified { pixel_samples, jitter_samples, rng_seed } => {
            sys::yk_sampler_desc { kind: sys::YK_SAMPLER_STRATIFIED, nx: pixel_samples.0 as u32, ny: pixel_samples.1 as u32, jitter: jitter_samples as u32, seed: rng_seed }
        }
    }
}

pub(crate) fn integrator_desc(p: &path::Params) -> sys::yk_integrator_desc {
    sys::yk_integrator_desc { kind: sys::YK_INTEGRATOR_PATH, max_depth: p.max_depth, has_clamp: p.indirect_clamp.is_some() as u32, indirect_clamp: p.indirect_clamp.unwrap_or(0.0) }
}

/// whitted.rs:17-25 (the device keeps at most 16 suspended calls: max_depth <= 16)
pub(crate) fn whitted_desc(p: &whitted::Params) -> sys::yk_integrator_desc {
    sys::yk_integrator_desc { kind: sys::YK_INTEGRATOR_WHITTED, max_depth: p.max_depth, has_clamp: 0, indirect_clamp: 0.0 }
}

/// Either device integrator: what differs is the `yk_integrator_desc` handed to the library.
pub struct HipPath {
    desc: sys::yk_integrator_desc,
    gpu: Arc<HipDevice>,
}

impl HipPath {
    pub fn new(params: path::Params, gpu: Arc<HipDevice>) -> Self {
        Self { desc: integrator_desc(&params), gpu }
    }
    pub fn whitted(params: whitted::Params, gpu: Arc<HipDevice>) -> Self {
        Self { desc: whitted_desc(&params), gpu }
    }
}

impl Integrator for HipPath {
    fn li(&self, _s: &ScopedScratch, _ray: Ray<f32>, _scene: &Scene, _depth: u32, _sampler: &mut Box<dyn Sampler>) -> RadianceResult {
        unimplemented!("per-ray li goes through yk_li; the UI's debug ray keeps using the CPU Path")
    }

    /// One tile through the device, the way the unchanged render manager calls it: from `num_cpus - 1` worker threads at once.
    /// The calls that are waiting at the same time share a submission (yk_combiner): 0.24 ms per tile from 15 workers on the
    /// cfg3 scene against 0.67 ms with a context per worker and 2.0 ms from one thread.  The whole queue in one
    /// call (gpu_worker.rs) is still 10x faster: 256 pixels per caller is little.
    fn render(
        &self,
        _scratch: &ScopedScratch,
        _scene: &Scene,
        camera: &Camera,
        sampler: &Arc<dyn Sampler>,
        accumulating: bool,
        tile: &mut FilmTile,
        tile_pixels: &mut [Spectrum<f32>],
        early_termination_predicate: &mut dyn FnMut() -> bool,
    ) -> usize {
        assert!(tile_pixels.len() >= tile.bb.area() as usize); // integrators/mod.rs:131
        if early_termination_predicate() {
            return 0;
        }
        let (cam, smp, integ) = (camera_desc(camera), sampler_desc(sampler.as_ref()), self.desc);
        // the predicate the worker polls per sample (integrators/mod.rs:153): the combiner polls it from THIS thread about every
        // 100 us while the tile waits or renders (it consumes a channel message, render_worker.rs:240-249: it must fire here)
        unsafe extern "C" fn poll(user: *mut c_void) -> std::os::raw::c_int {
            let pred = &mut *(user as *mut &mut dyn FnMut() -> bool);
            pred() as std::os::raw::c_int
        }
        let mut pred_ref: &mut dyn FnMut() -> bool = early_termination_predicate;
        let user = &mut pred_ref as *mut &mut dyn FnMut() -> bool as *mut c_void;
        let t = sys::yk_tile { x0: tile.bb.p_min.x, y0: tile.bb.p_min.y, x1: tile.bb.p_max.x, y1: tile.bb.p_max.y };
        let mut stats = sys::yk_render_stats::default();
        // Spectrum<f32> is three packed f32 (math/spectrum.rs:45-55)
        let out = tile_pixels.as_mut_ptr() as *mut f32;
        // accumulating: FilmTile.sample is the sample's global index (types_fit, integrators/mod.rs:140-141); otherwise -1 = all samples
        let sample: i32 = if accumulating { tile.sample as i32 } else { -1 };
        let st = unsafe { sys::yk_combiner_render_tile(self.gpu.combiner, self.gpu.scene, &cam, &smp, &integ, &t, sample, out, &mut stats, Some(poll), user) };
        if st == sys::YK_ERR_CANCELLED {
            return 0; // tile contents undefined, discarded by the caller like render_worker.rs:252-255
        }
        assert_eq!(st, sys::YK_OK, "HIP render failed: {}", sys::combiner_last_error(self.gpu.combiner));
        stats.rays as usize // the submission's rays shared out by tile area: exact in sum, which is what render_manager.rs:277-281 does with it
    }
}
