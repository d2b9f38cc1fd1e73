//! `yuki/src/renderer/gpu_multi.rs` — all GPUs of the node behind one worker.  SOURCE ONLY.
//!
//! `RenderManager` owns every worker of the process (render_manager.rs:78-97) and deals the
//! film's tiles to them (:125-143; "interleave tiles" is its TODO at :206-210); finished tiles
//! go back through `Film::update_tile` (film.rs:210-282).  With `yk_multi` the workers are the
//! node's GPUs and the dealing, the exchange and the write-back happen inside the library:
//! one context + host thread per device, the BVH built once and copied to each, spiral tile i
//! on device i mod G, slabs to device 0 over RCCL (ncclSend / ncclRecv on the render streams),
//! `Film::update_tile` there, one read-back of the finished film.
#![cfg(feature = "hip")]

use super::render_worker::{Message, Payload, WorkerInfo};
use crate::{
    film::FilmSettings,
    integrators::{
        hip_path::{camera_desc, flatten_scene, integrator_desc, sampler_desc},
        IntegratorType,
    },
    math::Spectrum,
    scene::Scene,
};
use std::{
    os::raw::{c_int, c_void},
    ptr,
    sync::mpsc::{Receiver, Sender},
    time::Instant,
};
use yuki_hip_sys as sys;

/// Every GPU the process was given, the scene on each, and the film's tile lists.
pub struct HipNode {
    multi: *mut sys::yk_multi,
    scene: *mut sys::yk_multi_scene,
    film: *mut sys::yk_multi_film,
    res: (u16, u16),
}
unsafe impl Send for HipNode {}

impl Drop for HipNode {
    fn drop(&mut self) {
        unsafe {
            sys::yk_multi_film_destroy(self.film);
            sys::yk_multi_scene_destroy(self.scene);
            sys::yk_multi_destroy(self.multi);
        }
    }
}

fn why(m: *const sys::yk_multi) -> String {
    let mut buf = [0i8; 512];
    unsafe {
        sys::yk_multi_last_error(m, buf.as_mut_ptr(), buf.len());
        std::ffi::CStr::from_ptr(buf.as_ptr()).to_string_lossy().into_owned()
    }
}

impl HipNode {
    /// `devices[0]` assembles the film.  Called when a scene is loaded / the film is resized.
    pub fn new(scene: &Scene, devices: &[i32], film: &FilmSettings) -> Result<Self, String> {
        unsafe {
            let mut multi = ptr::null_mut();
            let st = sys::yk_multi_create(devices.as_ptr(), devices.len() as u32, &mut multi);
            if st != sys::YK_OK {
                return Err(format!("yk_multi_create: status {st}"));
            }
            let mut scn = ptr::null_mut();
            let st = flatten_scene(scene, |desc| sys::yk_multi_scene_create(multi, desc, &mut scn));
            if st != sys::YK_OK {
                let e = why(multi);
                sys::yk_multi_destroy(multi);
                return Err(format!("yk_multi_scene_create: {e}"));
            }
            let mut flm = ptr::null_mut();
            let st = sys::yk_multi_film_create(multi, film.res.x, film.res.y, film.tile_dim, &mut flm);
            if st != sys::YK_OK {
                let e = why(multi);
                sys::yk_multi_scene_destroy(scn);
                sys::yk_multi_destroy(multi);
                return Err(format!("yk_multi_film_create: {e}"));
            }
            Ok(Self { multi, scene: scn, film: flm, res: (film.res.x, film.res.y) })
        }
    }
}

/// The library calls the predicate from ONE thread at a time and never again after it returned
/// non-zero (include/yuki_hip.h, yk_multi_render_film: the devices' host threads share a latch) —
/// what `Receiver::try_recv` (!Sync, consuming: render_worker.rs:240-249) needs — and the first
/// non-zero answer stops every device, so an interrupt does not cost the other GPUs' shares.
struct CancelCtx<'a> {
    from_parent: &'a Receiver<Option<Payload>>,
    interrupted_by: Option<Option<Payload>>,
}
unsafe extern "C" fn cancel_trampoline(user: *mut c_void) -> c_int {
    let c = &mut *(user as *mut CancelCtx);
    if let Ok(msg) = c.from_parent.try_recv() {
        c.interrupted_by = Some(msg);
        1
    } else {
        0
    }
}

/// One payload = one frame: every tile of the film on its device, then the film itself.
/// (The per-tile queue of render_manager.rs:125-143 is not consulted: the library's deal is the
/// same interleave over the same spiral, film.rs:333-376.)  `payload.accumulate` (the interactive
/// mode: render_manager.rs:125-143 re-queues every tile once per sample index and
/// Film::update_tile adds, film.rs:260-272): all `spp` passes in submissions of PASSES passes,
/// the film cleared first, the host's copy refreshed after each submission like the reference's
/// per-pass tile updates.  The predicate is polled about every 100 us while the GPUs work and a
/// non-zero answer returns within a few milliseconds (yk_cancel_fn).
pub fn render_payload(node: &HipNode, info: WorkerInfo, payload: &Payload, from_parent: &Receiver<Option<Payload>>, to_parent: &Sender<Message>) -> Option<Option<Payload>> {
    let params = match &payload.integrator_type {
        IntegratorType::HipPath(p) => p.clone(),
        _ => unreachable!("the GPU worker only runs HipPath"),
    };
    payload.tiles.lock().unwrap().clear(); // the whole film is rendered below
    let (cam, smp, integ) = (camera_desc(&payload.camera), sampler_desc(payload.sampler.as_ref()), integrator_desc(&params));
    let mut pixels = vec![Spectrum::<f32>::zeros(); node.res.0 as usize * node.res.1 as usize];
    let mut stats = sys::yk_render_stats::default();
    let mut cancel = CancelCtx { from_parent, interrupted_by: None };
    let start = Instant::now();
    let user = &mut cancel as *mut CancelCtx as *mut c_void;
    let mut rays = 0usize;
    if payload.accumulate {
        const PASSES: u32 = 4; // passes per submission: 1.5x the rays per second of pass-by-pass
        let spp = payload.sampler.samples_per_pixel() as u32;
        assert_eq!(unsafe { sys::yk_multi_film_clear(node.multi, node.film) }, sys::YK_OK);
        let mut first = 0u32;
        while first < spp {
            let n = PASSES.min(spp - first);
            let st = unsafe {
                sys::yk_multi_accumulate_film(node.multi, node.scene, &cam, &smp, &integ, node.film, first, n, pixels.as_mut_ptr() as *mut f32, &mut stats, Some(cancel_trampoline), user)
            };
            if st == sys::YK_ERR_CANCELLED {
                return cancel.interrupted_by;
            }
            assert_eq!(st, sys::YK_OK, "multi-GPU render failed: {}", why(node.multi));
            first += n;
            rays += stats.rays as usize;
            // the running sum over `first` samples: what the film holds in accumulate mode (film.rs:260-272)
            payload.film.lock().unwrap().set_accumulated_pixels(&pixels, first as u16);
        }
    } else {
        let st = unsafe {
            sys::yk_multi_render_film(node.multi, node.scene, &cam, &smp, &integ, node.film, pixels.as_mut_ptr() as *mut f32, &mut stats, Some(cancel_trampoline), user)
        };
        if st == sys::YK_ERR_CANCELLED {
            return cancel.interrupted_by;
        }
        assert_eq!(st, sys::YK_OK, "multi-GPU render failed: {}", why(node.multi));
        rays = stats.rays as usize;
        // row-major RGB, row 0 = top: the layout of Film::pixels (film.rs:67-113)
        payload.film.lock().unwrap().set_pixels(&pixels);
    }
    let _ = to_parent.send(Message::TileDone { info, ray_count: rays, elapsed_s: start.elapsed().as_secs_f32() });
    let _ = to_parent.send(Message::Finished(info));
    None
}
