#!/usr/bin/env python3
"""Headline benchmark: Mray/s of the Path integrator at 1080p (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over the workload: every pixel x sample
of the 1920x1080 film through raygen -> {trace, shade, shadow, accumulate} x 8 ->
resolve, scene and camera already resident in HBM.  Workload (SURVEY.md §8(d),
BASELINE.json configs[2], the configuration the >=1 Gray/s target is quoted on):
~1 M-triangle synthetic scene, SAH BVH, Path 8 bounces, Stratified 8x8 = 64 spp.

N > 1 (launched by torch.distributed.run, one rank per GPU): the film's spiral
tile list (film.rs:333-376) is dealt round-robin to the ranks, the scene is
replicated, each rank renders its tiles into HBM and one RCCL gather moves the
per-tile radiance to rank 0, which scatters it into the film (Film::update_tile).
Total work is fixed -> "scaling": "strong".  Steps are enqueued without host
synchronisation, alternately on two contexts (frames in flight = 2), so the latency
tail of one step overlaps the bulk of the next; N = 1 times synchronous steps (live
per-kernel HIP-event timings); `--two-in-flight` adds the two-in-flight rate of the same
steps as `extra.two_in_flight`.

Metric = the reference's own: closest-hit rays / second (path.rs:87,
app/window.rs:911-916); shadow rays are traced but not counted.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto 4 hardware queues by default and streams that share a queue serialise;
# two frames in flight use four busy streams (two contexts x {main, side}) beside torch's and
# RCCL's.  Must be set before the HIP runtime initialises (measured: 148.1 -> 141.7 ms per frame).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
GATHER_CEILING_GBPS = 14600.0
HBM_COPY_CEILING_GBPS = 6290.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def workload(name):
    from yuki_amd import core as yk
    from yuki_amd import scenes

    if name == "cfg3":
        return dict(scene="cfg3", res=(1920, 1080), sampler=yk.SamplerType.Stratified((8, 8), True), depth=8,
                    desc="cfg3: city 40x20 displaced icospheres, 1,024,012 triangles, SAH BVH (max_shapes_in_node 1), Path 8 bounces, Stratified 8x8, 1920x1080")
    if name == "cfg2":
        return dict(scene="cfg2", res=(1920, 1080), sampler=yk.SamplerType.Uniform(16), depth=8,
                    desc="cfg2: bunny-class 69,312-triangle mesh, SAH BVH, Path 8 bounces, Uniform 16 spp, 1920x1080")
    if name == "smoke":
        return dict(scene="city-small", res=(320, 180), sampler=yk.SamplerType.Stratified((2, 2), True), depth=8,
                    desc="smoke: city-small 7,692 triangles, Path 8, Stratified 2x2, 320x180")
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(sd, cam, sampler, integ, tiles, n_sample_tiles):
    """The oracle (CPU restatement of the reference, kind 'port') timed on this
    box's host cores on a bounded sample of the same workload."""
    from oracle import binding as oracle

    t0 = time.time()
    osc = oracle.OracleScene(sd)
    build_s = time.time() - t0
    sample = tiles[:n_sample_tiles]
    # render_manager.rs:78: num_cpus - 1 workers; a one-GPU box gives us a 16-core share
    share = int(os.environ.get("YK_CPU_SHARE", min(16, os.cpu_count() or 2)))
    cores = max(1, share - 1)
    t0 = time.time()
    _, rays, stats = osc.render_tiles(cam.matrices, sampler, integ, sample, n_threads=cores, want_stats=True)
    dt = time.time() - t0
    osc.close()
    counters = dict(
        node_tests_per_ray=stats.closest_node_tests / max(1, stats.closest_rays),
        shape_tests_per_ray=stats.closest_shape_tests / max(1, stats.closest_rays),
        shadow_rays_per_ray=stats.shadow_rays / max(1, stats.closest_rays),
        shadow_node_tests_per_ray=stats.shadow_node_tests / max(1, stats.closest_rays),
        shadow_shape_tests_per_ray=stats.shadow_shape_tests / max(1, stats.closest_rays),
    )
    return dict(value=rays / dt * 1e-6, unit="Mray/s", cores=cores, kind="port",
                sample=f"first {len(sample)} spiral-order 16x16 tiles of the same frame at full spp ({rays} rays in {dt:.1f} s; oracle BVH build {build_s:.1f} s excluded)"), counters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--async-steps", action="store_true", help="N=1: enqueue the timed steps without host synchronisation, as N>1 always does")
    ap.add_argument("--cpu-sample-tiles", type=int, default=384)
    ap.add_argument("--rccl-single", action="store_true",
                    help="N=1 through the N>1 code path: a one-rank RCCL process group, asynchronous slots, gather, scatter "
                         "(checks the collective's ordering on the context streams on a single GPU)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 rehearsal on a single GPU: every rank uses cuda:0 and the gather goes through gloo and host memory "
                         "(RCCL refuses two ranks on one device); exercises the N>1 control flow, its number means nothing")
    ap.add_argument("--two-in-flight", action="store_true",
                    help="N=1: after the timed region, also time the same K steps asynchronously on two contexts (extra.two_in_flight); "
                         "off by default so that a rocprofv3 summary of the default command holds undisturbed launches only")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="asynchronous steps alternate between this many contexts/streams (default: 2 for N>1, 1 for N=1); "
                         "the latency tail of step k then overlaps the bulk of step k+1")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    use_dist = world > 1 or args.rccl_single  # the multi-rank control flow (also with one rank, for rehearsal)
    if use_dist:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_on_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from yuki_amd import scenes
    from yuki_amd import core as yk
    from yuki_amd import dist as ydist

    wl = workload(args.workload)
    t0 = time.time()
    sd = scenes.by_name(wl["scene"])
    gen_s = time.time() - t0
    opts = {}
    if args.batch_paths:
        opts["batch_paths"] = args.batch_paths
    ctx = yk.Context(local_rank, **opts)
    scene = yk.Scene(ctx, sd)  # one device copy, rendered by every context of this rank
    info = scene.info()
    fs = yk.FilmSettings(res=wl["res"], tile_dim=16)
    cam = yk.Camera(sd.camera, fs)
    sampler = wl["sampler"]
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=wl["depth"]))
    it = yk.IntegratorType.instantiate(ctx, integ)
    tiles = yk.film_tiles(fs)
    spp = yk.samples_per_pixel(sampler)
    if rank == 0:
        log(f"[bench] scene {sd.name}: gen {gen_s:.2f}s, BVH build {info.build_seconds:.2f}s ({info.n_nodes} nodes, depth {info.tree_depth}), "
            f"upload {info.upload_seconds:.2f}s, {info.device_bytes / 1e6:.0f} MB in HBM; {len(tiles)} tiles, {spp} spp")

    # tile i -> rank i mod G (interleaved deal of the spiral order)
    my_tiles = ydist.shard_tiles(tiles, rank, world)
    slab_px = ydist.slab_pixels(tiles, world)
    slab = torch.zeros(slab_px * 3, dtype=torch.float32, device=dev)
    film = torch.zeros(wl["res"][1] * wl["res"][0] * 3, dtype=torch.float32, device=dev) if rank == 0 else None
    gathered = [torch.zeros_like(slab) for _ in range(world)] if (rank == 0 and use_dist) else None

    # Prepared tile lists: the pixel tables live on the device, so a step needs no upload.
    my_list = yk.TileList(ctx, my_tiles)
    rank_lists = [yk.TileList(ctx, ydist.shard_tiles(tiles, r, world)) for r in range(world)] if (rank == 0 and use_dist) else None
    # N > 1 (or --async-steps): every launch of a step — render, RCCL gather, film scatter — is
    # enqueued on one stream and nothing waits on the host inside the timed region;
    # ray counts are taken from one synchronous step beforehand (every step renders the same frame).
    async_steps = use_dist or args.async_steps
    # Asynchronous steps alternate between `in_flight` slots — a context (work buffers, HIP
    # streams), a torch stream, a slab and gather buffers each — so that the latency tail of step k
    # (late bounces: few rays, every launch as long as its longest ray) runs beside the bulk of
    # step k+1.  Every slot renders the same scene copy and tile list; steps stay ordered per slot.
    in_flight = max(1, args.frames_in_flight or (2 if use_dist else 1)) if async_steps else 1
    slots = [dict(ctx=ctx, it=it, slab=slab, gathered=gathered, film=film)]
    for _ in range(1, in_flight):
        c2 = yk.Context(local_rank, **opts)
        slots.append(dict(ctx=c2, it=yk.IntegratorType.instantiate(c2, integ), slab=torch.zeros_like(slab),
                          gathered=[torch.zeros_like(slab) for _ in range(world)] if gathered is not None else None,
                          film=torch.zeros_like(film) if film is not None else None))
    # A slot's work — render, RCCL gather, film scatter — is ordered on its context's own stream
    # (torch sees it as an ExternalStream): no further stream takes part, so the main / side stream
    # pairs of the slots are the only busy streams (HIP shares hardware queues between streams).
    for sl in slots:
        sl["stream"] = torch.cuda.ExternalStream(sl["ctx"].stream_handle, device=dev) if async_steps else None
    step_no = [0]

    def step(want_stats=True):
        sl = slots[step_no[0] % in_flight]
        step_no[0] += 1
        if sl["stream"] is not None:
            with torch.cuda.stream(sl["stream"]):
                return step_on(sl, want_stats)
        return step_on(sl, want_stats)

    def step_on(sl, want_stats):
        st = sl["it"].render_tile_list_device(scene, cam, sampler, my_list, sl["slab"].data_ptr(), stream=None, want_stats=want_stats)
        if use_dist:
            if args.rehearse_on_one_gpu:  # gloo gathers host tensors
                host = sl["slab"].cpu()
                parts = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0:
                    for r in range(world):
                        sl["gathered"][r].copy_(parts[r])
            else:
                dist.gather(sl["slab"], sl["gathered"], dst=0)  # RCCL, ordered after the render through torch's current stream = the slot's
            if rank == 0:
                for r in range(world):
                    rank_lists[r].update_film_device(sl["gathered"][r].data_ptr(), wl["res"], sl["film"].data_ptr(), ctx=sl["ctx"])
        else:
            my_list.update_film_device(sl["slab"].data_ptr(), wl["res"], sl["film"].data_ptr(), ctx=sl["ctx"])
        return st

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if in_flight > 1:  # setup: every slot allocates its work buffers on its first step
        sync()
        for _ in range(in_flight):
            step()
            sync()
    for _ in range(args.warmup):
        step()
    sync()
    probe = step() if async_steps else None  # untimed, alone on the GPU: per-step ray counts and kernel timings for the asynchronous mode
    sync()
    t0 = time.perf_counter()
    rays = shadow = 0
    t_trace = t_shadow = t_shade = t_dev = 0.0
    launches = shadow_launches = 0
    for _ in range(args.steps):
        st = step(want_stats=not async_steps) or probe
        rays += st.rays
        shadow += st.shadow_rays
        t_trace += st.seconds_trace
        t_shadow += st.seconds_shadow
        t_shade += st.seconds_shade
        t_dev += st.seconds_total
        launches += st.trace_launches
        shadow_launches += st.shadow_launches
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([rays, shadow], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rays_all, shadow_all = int(c[0].item()), int(c[1].item())
    else:
        rays_all, shadow_all = rays, shadow

    # N = 1, synchronous default: also time the same K steps enqueued asynchronously on two
    # contexts — what N > 1 does by default — so that the scaling figures have a like-for-like base.
    two_in_flight = None
    if world == 1 and in_flight == 1 and args.two_in_flight:
        c2 = yk.Context(local_rank, **opts)
        pair = [(ctx, it, slab, film), (c2, yk.IntegratorType.instantiate(c2, integ), torch.zeros_like(slab), torch.zeros_like(film))]

        def enqueue(i):  # everything of a step on its context's own stream
            c_, it_, slab_, film_ = pair[i % 2]
            it_.render_tile_list_device(scene, cam, sampler, my_list, slab_.data_ptr(), stream=None, want_stats=False)
            my_list.update_film_device(slab_.data_ptr(), wl["res"], film_.data_ptr(), ctx=c_)

        torch.cuda.synchronize()
        for i in range(2):
            enqueue(i)
            torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(args.steps):
            enqueue(i)
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp
        if not torch.equal(pair[1][3], film):
            raise SystemExit("[bench] films of the two contexts differ")
        two_in_flight = {"value": rays_all / tp * 1e-6, "ms_per_step": tp / args.steps * 1e3,
                         "note": "same K steps, enqueued without host sync alternately on two contexts/streams (the N>1 default)"}
        c2.close()

    if rank == 0:
        film_mean = film.view(-1, 3).mean(dim=0).tolist()
        for sl in slots[1:]:
            if not torch.equal(sl["film"], film):
                raise SystemExit("[bench] films of the slots in flight differ")
        value = rays_all / elapsed * 1e-6
        cpu = None
        counters = None
        stored = os.path.join(ROOT, "profiles", f"oracle_counters_{args.workload}.json")
        if world == 1 and not args.no_cpu_baseline:
            cpu, counters = cpu_baseline(sd, cam, sampler, integ, tiles, args.cpu_sample_tiles)
            log(f"[bench] cpu_baseline {cpu['value']:.3f} Mray/s on {cpu['cores']} threads; oracle counters {counters}")
        elif os.path.exists(stored):
            counters = json.load(open(stored))["counters"]
        roofline = None
        if counters and t_trace > 0:
            # dominant kernel = k_trace_closest.  Algorithmic bytes per ray it traces:
            #   32 B per BVH node test + 36 B per leaf triangle test (the oracle's counters
            #   on this scene/seed) + 32 B ray read + 16 B hit record   
            b_closest = 32.0 * counters["node_tests_per_ray"] + 36.0 * counters["shape_tests_per_ray"] + 48.0
            achieved = b_closest * rays / t_trace / 1e9
            # whole-path figure of SURVEY §8(d): B_ray = 32*N_node + 36*N_tri + 304 with shadow tests attributed
            b_ray = 32.0 * (counters["node_tests_per_ray"] + counters["shadow_node_tests_per_ray"]) + 36.0 * (counters["shape_tests_per_ray"] + counters["shadow_shape_tests_per_ray"]) + 304.0
            pmc = os.path.join(ROOT, "profiles", f"pmc_{args.workload}.json")
            pmc_d = json.load(open(pmc)) if os.path.exists(pmc) else {}

            def family(name, kernels, bytes_total, seconds, n_launch, units, traffic):
                ach = bytes_total / max(seconds, 1e-12) / 1e9
                return dict(bound="hbm", kernel=name, kernels=kernels, achieved=ach, peak=HBM_PEAK_GBPS, unit="GB/s", frac=ach / HBM_PEAK_GBPS, traffic=traffic,
                            bytes_per_ray=bytes_total / max(1, units), avg_launch_ms=seconds / max(1, n_launch) * 1e3, launches=n_launch,
                            rays_per_launch=units / max(1, n_launch), seconds_per_step=seconds / args.steps,
                            frac_of_copy_ceiling=ach / HBM_COPY_CEILING_GBPS,
                            # measured ceiling of per-lane 64-byte gathers from an L1/L2-resident table (tools/micro/gather_bench.hip,
                            # DESIGN.md §4): the unit the traversal kernels are actually bound by
                            frac_of_gather_ceiling=ach / GATHER_CEILING_GBPS)

            # any-hit family: 32 B per node test + 36 B per triangle test of the shadow rays (the oracle's counters are per
            # counted ray, so x rays) + 32 B ray read + 4 B slot per shadow ray
            bytes_any = (32.0 * counters["shadow_node_tests_per_ray"] + 36.0 * counters["shadow_shape_tests_per_ray"]) * rays + 36.0 * shadow
            fam_closest = family("k_trace_closest", ["k_trace_closest_pt", "k_trace_closest_packet"], b_closest * rays, t_trace, launches, rays,
                                 pmc_d.get("closest", {}).get("hbm_bytes_per_launch", pmc_d.get("hbm_bytes_per_launch")))
            fam_any = family("k_trace_any", ["k_trace_any_pt", "k_trace_any_packet"], bytes_any, t_shadow, shadow_launches, shadow,
                             pmc_d.get("any", {}).get("hbm_bytes_per_launch"))
            # the dominant kernel family is the one with the larger summed launch duration (HIP events on the launch streams;
            # any-hit launches share the GPU with the next bounce's closest-hit launch, so their durations include that)
            roofline, other = (fam_any, fam_closest) if t_shadow > t_trace else (fam_closest, fam_any)
            roofline["other"] = other
            roofline.update(path_bytes_per_ray=b_ray, path_achieved=b_ray * (rays / max(t_dev, 1e-9)) / 1e9,
                            path_frac=b_ray * (rays / max(t_dev, 1e-9)) / 1e9 / HBM_PEAK_GBPS)
        out = {
            "metric": "Mray/s (primary+secondary), Path integrator @1080p",
            "value": value,
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["desc"], "triangles": sd.n_triangles, "spp": spp, "max_depth": wl["depth"], "tiles": int(len(tiles)),
                       "partition": f"spiral tiles dealt round-robin to {world} rank(s), RCCL gather to rank 0" if use_dist else "single GPU",
                       "frames_in_flight": in_flight},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "extra": {"rays_per_step": rays_all // args.steps, "shadow_rays_per_step": shadow_all // args.steps,
                      "shadow_Mray_per_s": shadow_all / elapsed * 1e-6, "rank0_device_s_per_step": t_dev / args.steps,
                      "rank0_trace_s": t_trace / args.steps, "rank0_shadow_s": t_shadow / args.steps, "rank0_shade_s": t_shade / args.steps,
                      "two_in_flight": two_in_flight, "film_mean_rgb": film_mean, "bvh_build_s": info.build_seconds, "scene_upload_s": info.upload_seconds,
                      "steps_mode": "asynchronous (no host sync inside the timed region; per-kernel times and ray counts from one untimed probe step)" if async_steps
                      else "synchronous (per-kernel HIP-event times read after every step of the timed region)"},
        }
        print(json.dumps(out), flush=True)
    scene.close()
    for sl in slots:
        sl["ctx"].close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
