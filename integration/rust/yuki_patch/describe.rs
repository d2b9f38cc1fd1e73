//! UPSTREAM ADDITIONS (no behaviour change): read-only descriptions of the trait objects a
//! `Scene` holds, so that an external integrator can flatten it.  All fields of `Triangle`,
//! `Sphere`, the materials, the lights and the textures are private in the reference
//! (shapes/triangle.rs:17-22, shapes/sphere.rs:15-21, materials/*.rs, lights/*.rs,
//! textures/*.rs), and the traits have no downcast hook.
//!
//! Add to each trait one defaulted method and implement it in the named files.

use crate::{
    lights::AreaLight,
    materials::Material,
    math::{Point3, Spectrum, Transform, Vec3},
    shapes::Mesh,
};
use std::sync::Arc;

/// textures/mod.rs — `trait Texture<T>`: `fn describe(&self) -> TextureDesc<T>;`
pub enum TextureDesc<'a, T> {
    /// textures/constant.rs: `TextureDesc::Constant(self.value)`
    Constant(T),
    /// textures/image_texture.rs: `TextureDesc::Image { data: &self.data, width: self.width, height: self.height }`
    Image { data: &'a [T], width: usize, height: usize },
}

/// materials/mod.rs — `trait Material`: `fn describe(&self) -> MaterialDesc;`
pub enum MaterialDesc<'a> {
    /// materials/matte.rs: kd, sigma
    Matte { kd: TextureDesc<'a, Spectrum<f32>>, sigma: TextureDesc<'a, f32> },
    /// materials/glass.rs: r, t, eta
    Glass { r: TextureDesc<'a, Spectrum<f32>>, t: TextureDesc<'a, Spectrum<f32>>, eta: f32 },
    /// materials/metal.rs: eta, k, roughness, remap_roughness
    Metal { eta: TextureDesc<'a, Spectrum<f32>>, k: TextureDesc<'a, Spectrum<f32>>, roughness: TextureDesc<'a, f32>, remap_roughness: bool },
    /// materials/glossy.rs: rs, roughness, remap_roughness
    Glossy { rs: TextureDesc<'a, Spectrum<f32>>, roughness: TextureDesc<'a, f32>, remap_roughness: bool },
}

/// shapes/mod.rs — `trait Shape`: `fn describe(&self) -> ShapeDesc;`
pub enum ShapeDesc<'a> {
    /// shapes/triangle.rs
    Triangle { mesh: &'a Arc<Mesh>, vertices: [usize; 3], material: &'a Arc<dyn Material>, area_light: Option<&'a Arc<dyn AreaLight>> },
    /// shapes/sphere.rs
    Sphere { object_to_world: &'a Transform<f32>, world_to_object: &'a Transform<f32>, radius: f32, material: &'a Arc<dyn Material> },
}

/// lights/mod.rs — `trait Light`: `fn describe(&self) -> LightDesc;`
pub enum LightDesc<'a> {
    /// lights/point_light.rs
    Point { p: Point3<f32>, i: Spectrum<f32> },
    /// lights/spot_light.rs
    Spot { world_to_light: &'a Transform<f32>, p: Point3<f32>, i: Spectrum<f32>, cos_total_width: f32, cos_falloff_start: f32 },
    /// lights/distant_light.rs
    Distant { w: Vec3<f32>, radiance: Spectrum<f32> },
    /// lights/rectangular_light.rs (also the `AreaLight` its triangles point at)
    Rectangular { sample_to_world: &'a Transform<f32>, l: Spectrum<f32>, area: f32 },
}

// camera.rs — `impl Camera`:
//     pub fn transforms(&self) -> (&Transform<f32>, &Transform<f32>) { (&self.camera_to_world, &self.raster_to_camera) }
//
// sampling/mod.rs — `trait Sampler`:
//     fn describe(&self) -> SamplerDesc;
// with
pub enum SamplerDesc {
    /// sampling/uniform.rs: pixel_samples, rng_seed
    Uniform { pixel_samples: u32, rng_seed: u64 },
    /// sampling/stratified.rs: pixel_samples, jitter_samples, rng_seed
    Stratified { pixel_samples: (u16, u16), jitter_samples: bool, rng_seed: u64 },
}
