//! `yuki/src/renderer/gpu_worker.rs` — replaces the `num_cpus - 1` tile workers
//! (render_manager.rs:78-97) with ONE worker that drains the whole tile queue into a single
//! `yk_render_tiles[_accumulating]` call.  A 16x16 tile is 256 pixels; the device wants
//! 10^7-10^8 camera samples in flight, so per-tile calls waste it.  SOURCE ONLY.
#![cfg(feature = "hip")]

use super::render_worker::{Message, Payload, WorkerInfo};
use crate::{
    film::FilmTile,
    integrators::{
        hip_path::{camera_desc, integrator_desc, sampler_desc, HipDevice},
        IntegratorType,
    },
    math::Spectrum,
};
use std::{
    os::raw::{c_int, c_void},
    sync::{
        mpsc::{Receiver, Sender},
        Arc,
    },
    time::Instant,
};
use yuki_hip_sys as sys;

struct CancelCtx<'a> {
    from_parent: &'a Receiver<Option<Payload>>,
    interrupted_by: Option<Option<Payload>>,
}

/// Forwards the manager's "new payload / kill" message as the early-termination predicate
/// (render_worker.rs:240-249).  Polled by the library between batches.
unsafe extern "C" fn cancel_trampoline(user: *mut c_void) -> c_int {
    let c = &mut *(user as *mut CancelCtx);
    if let Ok(msg) = c.from_parent.try_recv() {
        c.interrupted_by = Some(msg);
        1
    } else {
        0
    }
}

/// Passes of the accumulating film rendered per submission (8: 2.6x the rays per second of one).
const PASSES: usize = 8;

/// Body of the worker loop for one payload (the surrounding recv/kill handling is the
/// reference's `launch`, render_worker.rs:62-137, unchanged).
pub fn render_payload(gpu: &Arc<HipDevice>, info: WorkerInfo, payload: &Payload, from_parent: &Receiver<Option<Payload>>, to_parent: &Sender<Message>) -> Option<Option<Payload>> {
    let params = match &payload.integrator_type {
        IntegratorType::HipPath(p) => p.clone(),
        _ => unreachable!("the GPU worker only runs HipPath"),
    };
    // take every queued tile (render_worker.rs:172-180 pops one at a time)
    let tiles: Vec<FilmTile> = payload.tiles.lock().unwrap().drain(..).collect();
    if tiles.is_empty() {
        let _ = to_parent.send(Message::Finished(info));
        return None;
    }
    let yk_tiles: Vec<sys::yk_tile> = tiles.iter().map(|t| sys::yk_tile { x0: t.bb.p_min.x, y0: t.bb.p_min.y, x1: t.bb.p_max.x, y1: t.bb.p_max.y }).collect();
    let samples: Vec<u16> = tiles.iter().map(|t| t.sample as u16).collect();
    let n_px: usize = tiles.iter().map(|t| t.bb.area() as usize).sum();
    // Accumulating (interactive) mode: one pass is one sample per pixel, too little work for a
    // submission, so PASSES passes (samples t.sample .. t.sample + PASSES - 1) are rendered at
    // once, pass-major in `out`; render_manager.rs:135-143 then re-queues with sample + PASSES.
    // never beyond the sampler's samples per pixel: the library refuses sample + passes > spp (render_manager.rs:135-143
    // queues samples 0 .. spp-1 only; past that the stratified permutation is undefined)
    let left = payload.sampler.samples_per_pixel() as usize - tiles.iter().map(|t| t.sample as usize).max().unwrap_or(0);
    let passes: usize = if payload.accumulate { PASSES.min(left.max(1)) } else { 1 };
    let mut out = vec![Spectrum::<f32>::zeros(); n_px * passes];
    let (cam, smp, integ) = (camera_desc(&payload.camera), sampler_desc(payload.sampler.as_ref()), integrator_desc(&params));
    let mut stats = sys::yk_render_stats::default();
    let mut cancel = CancelCtx { from_parent, interrupted_by: None };
    let start = Instant::now();
    let st = unsafe {
        let user = &mut cancel as *mut CancelCtx as *mut c_void;
        if payload.accumulate {
            sys::yk_render_tiles_accumulating_passes(gpu.ctx, gpu.scene, &cam, &smp, &integ, yk_tiles.as_ptr(), samples.as_ptr(), yk_tiles.len(), passes as u32, out.as_mut_ptr() as *mut f32, &mut stats, Some(cancel_trampoline), user)
        } else {
            sys::yk_render_tiles(gpu.ctx, gpu.scene, &cam, &smp, &integ, yk_tiles.as_ptr(), yk_tiles.len(), out.as_mut_ptr() as *mut f32, &mut stats, Some(cancel_trampoline), user)
        }
    };
    if st == sys::YK_ERR_CANCELLED {
        return cancel.interrupted_by; // tile contents undefined: dropped like render_worker.rs:252-255
    }
    assert_eq!(st, sys::YK_OK, "HIP render failed: {}", sys::last_error(gpu.ctx));
    // `out` is tile-major, each tile row-major: exactly the `tile_pixels` layout of Film::update_tile
    let elapsed_s = start.elapsed().as_secs_f32();
    let mut film = payload.film.lock().unwrap();
    for pass in 0..passes {
        let mut off = pass * n_px;
        for t in &tiles {
            let n = t.bb.area() as usize;
            if film.matches(t) {
                film.update_tile(t, &out[off..off + n]); // copy, or accumulate + samples[tile] += 1: film.rs:210-282
            }
            off += n;
        }
    }
    drop(film);
    // one progress message for the whole batch; ray_count is the reference's (path.rs:87)
    let _ = to_parent.send(Message::TileDone { info, ray_count: stats.rays as usize, elapsed_s });
    let _ = to_parent.send(Message::Finished(info));
    None
}
