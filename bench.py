#!/usr/bin/env python3
"""Headline benchmark: Mray/s of the Path integrator at 1080p (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over the workload: every pixel x sample
of the 1920x1080 film through raygen -> {trace, shade, shadow, accumulate} x 8 ->
resolve, scene and camera already resident in HBM.  Workload (SURVEY.md §8(d),
BASELINE.json configs[2], the configuration the >=1 Gray/s target is quoted on):
~1 M-triangle synthetic scene, SAH BVH, Path 8 bounces, Stratified 8x8 = 64 spp.

N > 1 (one rank per GPU; the driver launches the ranks with torch.distributed.run — when
WORLD_SIZE is not set, `bench.py --gpus N` starts them itself, as child processes, before this
process has loaded torch or touched HIP, and relays rank 0's line): the film's spiral
tile list (film.rs:333-376) is dealt round-robin to the ranks, the scene is
replicated, each rank renders its tiles into HBM and one RCCL gather moves the
per-tile radiance to rank 0, which scatters it into the film (Film::update_tile).
Total work is fixed -> "scaling": "strong".  Steps are enqueued without host
synchronisation, alternately on two contexts (frames in flight = 2), so the latency
tail of one step overlaps the bulk of the next — at N = 1 too when the frame is a single
batch (`--sync-steps` restores one host synchronisation per step).  Per-kernel launch times
for the roofline come from an untimed probe step in which every launch runs alone.

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
HBM_COPY_CEILING_GBPS = 6290.0
L2_PEAK_GBPS = 34500.0  # MI355X_MICROARCH.md §L2: ~34.5 TB/s aggregate


def self_launch(args, stdout_fd):
    """`--gpus N` (N > 1) without a launcher: start the N ranks as children of this process — which has
    not imported torch and never touches HIP — relay their output (rank 0 prints the JSON line) and
    exit with their status.  Never falls through to a one-GPU run."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"[bench] WORLD_SIZE unset: launching {args.gpus} ranks: {' '.join(cmd)}")
    rc = subprocess.call(cmd, env=env, stdout=stdout_fd)  # the ranks write the JSON record to this process's real stdout
    raise SystemExit(rc)


def gather_ceiling(table_bytes):
    """Measured ceiling of per-lane 64-byte gathers (tools/micro/gather_bench.hip, raw output and parsed
    table tracked in profiles/): the entry whose table is the smallest one not smaller than `table_bytes`."""
    best = None
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if f.endswith("_gather_bench.json"):
            best = f  # latest round
    if best is None:
        return None
    d = json.load(open(os.path.join(ROOT, "profiles", best)))
    rows = sorted(d["tables"], key=lambda r: r["table_bytes"])
    row = next((r for r in rows if r["table_bytes"] >= table_bytes), rows[-1])
    return dict(GBps=row["GBps"], table_bytes=row["table_bytes"], tcp_accesses_per_s=row.get("tcp_accesses_per_s"), source=f"profiles/{best}")


def pmc_traffic(workload):
    """HBM-side bytes per launch of the two traversal kernel families from the latest tracked PMC passes."""
    best = None
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if f.endswith(f"_pmc_{workload}.json") or f == f"pmc_{workload}.json":
            best = f if best is None or f.startswith("r") else best
    if best is None:
        return {}, None
    return json.load(open(os.path.join(ROOT, "profiles", best))), f"profiles/{best}"


def git_blob_hash(path):
    """`git hash-object` of a tracked file, computed here (the GPU box has no .git): a line that quotes a stored PMC set names
    the exact content it read, so a stale set is visible next to the commit the bench ran at."""
    import hashlib

    data = open(os.path.join(ROOT, path), "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


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
    if name == "cfg5":
        return dict(scene="cfg5", res=(3840, 2160), sampler=yk.SamplerType.Stratified((16, 16), True), depth=16,
                    desc="cfg5: city 100x80 displaced icospheres, 10,240,012 triangles (GGX metal / perfect glass mix), SAH BVH, Path 16 bounces, Stratified 16x16, 3840x2160")
    if name == "cfg1":
        return dict(scene="cornell", res=(512, 512), sampler=yk.SamplerType.Uniform(1), depth=3, whitted=True,
                    desc="cfg1: built-in Cornell box, Whitted depth 3, Uniform 1 spp, 512x512")
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
    # Exactly ONE line on stdout, the JSON record: libraries (RCCL's version banner, gloo) print to fd 1 behind Python's
    # back, so fd 1 points at stderr until the record is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scene-file", default=None,
                    help="render a scene FILE instead of the generated one: .ply (Scene::ply, scene/mod.rs:99-152) or .pbrt (scene::pbrt::load, "
                         "scene/pbrt/mod.rs:94-857) through yk_load_ply / yk_load_pbrt; camera from the file, film resolution from a .pbrt's Film "
                         "directive (a .ply keeps the workload's), sampler / depth / integrator from --workload.  "
                         "tools/write_scene_files.py writes BASELINE's configs as such files")
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--async-steps", action="store_true", help="(default since round 2 for frames of one batch) enqueue the timed steps without host synchronisation")
    ap.add_argument("--solo-steps", action="store_true",
                    help="N=1 diagnostic: every step like the roofline's probe step - side stream off, one work set, one host synchronisation per step - "
                         "so that a rocprofv3 --kernel-trace --stats summary of this command holds each kernel's own duration (profiles/*_solo_kernel_stats.csv)")
    ap.add_argument("--sync-steps", action="store_true", help="N=1: one host synchronisation per step inside the timed region (the round-1 mode)")
    ap.add_argument("--cpu-sample-tiles", type=int, default=384)
    ap.add_argument("--rccl-single", action="store_true",
                    help="N=1 through the N>1 code path: a one-rank RCCL process group, asynchronous slots, gather, scatter "
                         "(checks the collective's ordering on the context streams on a single GPU)")
    ap.add_argument("--gather", choices=["abi", "torch"], default="abi",
                    help="N>1: who moves the slabs to rank 0 - 'abi': the library's own RCCL exchange (yk_dist_gather: ncclSend/ncclRecv on the "
                         "rendering context's stream, communicator joined through the C ABI); 'torch': torch.distributed.gather on the nccl backend. "
                         "'abi' falls back to 'torch' when the communicator cannot be created (reported in config.gather)")
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

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, real_stdout)  # does not return; nothing GPU-related has been imported yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")

    import torch

    if world > 1 and not args.rehearse_on_one_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) visible")
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
        # one node: the out-of-band sockets of gloo and of RCCL's bootstrap can use the loopback interface, which exists
        # and resolves everywhere (the container's hostname may not); GPU-to-GPU data does not travel over it
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if args.rehearse_on_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            # control plane (barriers, the max over ranks) on gloo / CPU tensors; the nccl (= RCCL) backend serves
            # `--gather torch` only and creates its communicator lazily, at the first collective on a device tensor
            torch.cuda.set_device(local_rank)
            dist.init_process_group("cpu:gloo,cuda:nccl")
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from yuki_amd import scenes
    from yuki_amd import core as yk
    from yuki_amd import dist as ydist

    wl = workload(args.workload)
    t0 = time.time()
    if args.scene_file:
        from yuki_amd import loaders

        is_pbrt = args.scene_file.lower().endswith(".pbrt")
        sd, _cam_params, film_settings = (loaders.load_pbrt if is_pbrt else loaders.load_ply)(args.scene_file)
        if is_pbrt:
            wl["res"] = tuple(film_settings.res)
        wl["desc"] = f"scene file {os.path.basename(args.scene_file)} ({sd.n_triangles} triangles, loaded by yk_load_{'pbrt' if is_pbrt else 'ply'}); sampler / depth of {args.workload}: " + wl["desc"]
    else:
        sd = scenes.by_name(wl["scene"])
    gen_s = time.time() - t0
    opts = {}
    if args.batch_paths:
        opts["batch_paths"] = args.batch_paths
    if args.solo_steps:
        opts.update(overlap_shadow=0, streams=1)
    ctx = yk.Context(local_rank, **opts)
    scene = yk.Scene(ctx, sd)  # one device copy, rendered by every context of this rank
    info = scene.info()
    fs = yk.FilmSettings(res=wl["res"], tile_dim=16)
    cam = yk.Camera(sd.camera, fs)
    sampler = wl["sampler"]
    integ = yk.IntegratorType.Whitted(wl["depth"]) if wl.get("whitted") else yk.IntegratorType.Path(yk.PathParams(max_depth=wl["depth"]))
    it = yk.IntegratorType.instantiate(ctx, integ)
    tiles = yk.film_tiles(fs)
    spp = yk.samples_per_pixel(sampler)
    if rank == 0:
        log(f"[bench] scene {sd.name}: {'load' if args.scene_file else 'gen'} {gen_s:.2f}s, BVH build {info.build_seconds:.2f}s ({info.n_nodes} nodes, depth {info.tree_depth}), "
            f"upload {info.upload_seconds:.2f}s, {info.device_bytes / 1e6:.0f} MB in HBM; {len(tiles)} tiles, {spp} spp")

    # tile i -> rank i mod G (interleaved deal of the spiral order)
    my_tiles = ydist.shard_tiles(tiles, rank, world)
    slab_px = ydist.slab_pixels(tiles, world)
    slab = torch.zeros(slab_px * 3, dtype=torch.float32, device=dev)
    film = torch.zeros(wl["res"][1] * wl["res"][0] * 3, dtype=torch.float32, device=dev) if rank == 0 else None

    def gather_buffers():  # rank 0: one allocation, world slabs back to back (what yk_dist_gather fills), and its per-rank views
        if not (rank == 0 and use_dist):
            return None, None
        whole = torch.zeros(world * slab.numel(), dtype=torch.float32, device=dev)
        return whole, [whole[r * slab.numel():(r + 1) * slab.numel()] for r in range(world)]

    gathered_all, gathered = gather_buffers()

    # Prepared tile lists: the pixel tables live on the device, so a step needs no upload.
    my_list = yk.TileList(ctx, my_tiles)
    rank_lists = [yk.TileList(ctx, ydist.shard_tiles(tiles, r, world)) for r in range(world)] if (rank == 0 and use_dist) else None
    # N > 1 (or --async-steps): every launch of a step — render, RCCL gather, film scatter — is
    # enqueued on one stream and nothing waits on the host inside the timed region;
    # ray counts are taken from one synchronous step beforehand (every step renders the same frame).
    # N = 1 runs its timed steps the way N > 1 does — enqueued without host synchronisation, alternating between two contexts —
    # when the frame is a single batch (the like-for-like base of the scaling figures; the roofline's launch times come from the
    # solo probe step, not from the timed region).  A frame of several batches already keeps two work sets busy and needs the
    # HBM for them: it stays on one context with synchronous steps.
    single_batch = (wl["res"][0] * wl["res"][1] * spp) <= (args.batch_paths or (128 << 20))
    if args.solo_steps:
        args.sync_steps = True
    async_steps = use_dist or args.async_steps or (single_batch and not args.sync_steps)
    # Asynchronous steps alternate between `in_flight` slots — a context (work buffers, HIP
    # streams), a torch stream, a slab and gather buffers each — so that the latency tail of step k
    # (late bounces: few rays, every launch as long as its longest ray) runs beside the bulk of
    # step k+1.  Every slot renders the same scene copy and tile list; steps stay ordered per slot.
    # Two slots; three once a rank's share is a quarter of the frame or less (1/4 share 35.97 -> 35.30 ms, 1/8 share 19.31 -> 18.85 ms;
    # 1/2 share 67.4 -> 67.9, whole frame 132.3 -> 133.7: tools/overlap_frames.py).
    share_single_batch = (wl["res"][0] * wl["res"][1] * spp) // max(1, world) <= (args.batch_paths or (128 << 20))
    default_slots = 3 if (world >= 4 and share_single_batch) else 2  # a share of several batches keeps two work sets of 128 M paths per slot: HBM
    in_flight = max(1, args.frames_in_flight or (default_slots if (use_dist or single_batch) else 1)) if async_steps else 1
    slots = [dict(ctx=ctx, it=it, slab=slab, gathered=gathered, gathered_all=gathered_all, film=film)]
    for _ in range(1, in_flight):
        c2 = yk.Context(local_rank, **opts)
        ga, gl = gather_buffers()
        slots.append(dict(ctx=c2, it=yk.IntegratorType.instantiate(c2, integ), slab=torch.zeros_like(slab), gathered=gl, gathered_all=ga,
                          film=torch.zeros_like(film) if film is not None else None))

    def cpu_all_reduce(values, op):  # control-plane reduction on the gloo backend
        t = torch.tensor(values, dtype=torch.float64)
        dist.all_reduce(t, op=op)
        return t.tolist()

    # Who carries the slabs: the library's own RCCL exchange (one communicator per slot, joined through the C ABI with an
    # id that rank 0 publishes in the rendezvous store), torch.distributed's nccl backend, or - rehearsal - gloo.
    gather_mode = None
    if use_dist:
        gather_mode = "gloo-host" if args.rehearse_on_one_gpu else args.gather
        if gather_mode == "abi":
            ok = 1.0
            try:
                from torch.distributed.distributed_c10d import _get_default_store

                store = _get_default_store()
                for k, sl in enumerate(slots):
                    key = f"yk_dist_id_{k}"
                    if rank == 0:
                        try:
                            uid = yk.Dist.unique_id()
                        except Exception as e:  # noqa: BLE001 - publish the failure: the other ranks must not wait for an id
                            log(f"[bench] rank 0: no RCCL id ({e})")
                            uid = b""
                        store.set(key, uid)
                    uid = bytes(store.get(key))
                    if len(uid) != yk.Dist.ID_BYTES:
                        raise RuntimeError("rank 0 could not create an RCCL unique id")
                    sl["dist"] = yk.Dist(sl["ctx"], uid, rank, world)
            except Exception as e:  # noqa: BLE001 - any failure means: use the other carrier, on every rank
                log(f"[bench] rank {rank}: yk_dist unavailable ({e}); falling back to torch.distributed.gather")
                ok = 0.0
            if cpu_all_reduce([ok], dist.ReduceOp.MIN)[0] < 1.0:
                for sl in slots:
                    if sl.get("dist"):
                        sl["dist"].close()
                        sl["dist"] = None
                gather_mode = "torch (fallback from abi)"
    # A slot's work — render, RCCL gather, film scatter — is ordered on its context's own stream
    # (torch sees it as an ExternalStream): no further stream takes part, so the main / side stream
    # pairs of the slots are the only busy streams (HIP shares hardware queues between streams).
    for sl in slots:
        sl["stream"] = torch.cuda.ExternalStream(sl["ctx"].stream_handle, device=dev) if async_steps else None
    step_no = [0]

    def step(want_stats=True, slot=None):
        if slot is None:
            slot = step_no[0] % in_flight
            step_no[0] += 1
        sl = slots[slot]
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
            elif gather_mode == "abi":  # ncclSend / ncclRecv on the slot's context stream, behind the render
                sl["dist"].gather(sl["slab"].data_ptr(), sl["gathered_all"].data_ptr() if rank == 0 else 0, sl["slab"].numel())
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
            cpu_all_reduce([0.0], dist.ReduceOp.SUM)  # barrier
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
    # Roofline probe (rank 0, N = 1): one untimed step with the side stream off, so that every traversal launch runs
    # alone on the GPU and its HIP-event duration is the kernel's own (in the default mode {any-hit, accumulate}(b)
    # share the GPU with closest-hit(b+1) and their events include the time they wait for each other).
    solo = None
    if world == 1 and not use_dist:
        ctx.set_option("overlap_shadow", 0)
        ctx.set_option("streams", 1)  # a frame of several batches: no second work set beside the first either
        solo = step(slot=0)  # `ctx` is slot 0's context
        sync()
        if not args.solo_steps:
            ctx.set_option("overlap_shadow", 1)
            ctx.set_option("streams", 2)
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
        elapsed = cpu_all_reduce([elapsed], dist.ReduceOp.MAX)[0]
        rays_all, shadow_all = (int(v) for v in cpu_all_reduce([rays, shadow], dist.ReduceOp.SUM))
    else:
        rays_all, shadow_all = rays, shadow

    # N = 1, synchronous default: also time the same K steps enqueued asynchronously on two
    # contexts — what N > 1 does by default — so that the scaling figures have a like-for-like base.
    two_in_flight = None
    if world == 1 and in_flight == 1 and args.two_in_flight and not async_steps:
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
        if solo is not None and solo.seconds_trace > 0 and solo.trace_launches:
            # Dominant kernel family = the traversal family with the larger summed duration in the solo probe step.
            # frac = HBM-side bytes per launch (PMC: FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes of this
            # command, tracked under profiles/) / the family's average launch duration measured live here (HIP events on
            # the launch stream, side stream off) / 8 TB/s.  FETCH_SIZE counts what leaves L2, Infinity-Cache hits
            # included, so this is an upper bound on DRAM traffic.  The ALGORITHMIC bytes of SURVEY §8(d) — 32 B per node
            # test, 36 B per triangle test (oracle counters) — are mostly L1 / LDS / L2 hits, not HBM bytes: they are
            # reported as `algorithmic`, never as an HBM fraction.
            pmc_d, pmc_src = pmc_traffic(args.workload)
            if args.batch_paths or args.scene_file:  # the tracked per-launch counters were collected with the default batch size and the generated scene
                log("[bench] --batch-paths / --scene-file given: the PMC-based roofline entries (traffic, TA busy, L1 / L2 rates) are left out")
                pmc_d, pmc_src = {}, None
            pmc_blob = git_blob_hash(pmc_src) if pmc_src else None
            table_bytes = int(info.n_interior) * 64 + int(info.n_shapes) * 48  # what the lanes gather from: 64-B nodes, 48-B triangles
            ceil_l1 = gather_ceiling(0)
            ceil_tab = gather_ceiling(table_bytes)

            def family(name, key, kernels, seconds, n_launch, units, alg_bytes, stream_bytes_per_unit):
                avg_s = seconds / max(1, n_launch)
                fam = pmc_d.get(key, {})
                traffic = None
                if fam.get("fetch_size_bytes_per_launch") is not None:
                    # FETCH_SIZE counts per-lane gathers (nodes, triangles) at full size and coalesced 16-B-per-lane streams (the ray
                    # records) at half (calibration: profiles/*_gather_bench.json): add the missing half of the streamed bytes
                    traffic = fam["fetch_size_bytes_per_launch"] + 0.5 * stream_bytes_per_unit * units / max(1, n_launch) + fam.get("write_size_bytes_per_launch", 0.0)
                d = dict(bound="hbm", what="L2-miss (fabric-side) traffic incl. Infinity-Cache hits: PMC FETCH_SIZE (gathers at full size, + the uncounted half of the streamed ray records) + WRITE_SIZE per launch / live launch duration",
                         kernel=name, kernels=kernels, achieved=None, peak=HBM_PEAK_GBPS, unit="GB/s", frac=None, traffic=traffic, traffic_source=pmc_src,
                         traffic_source_git_blob=pmc_blob,
                         avg_launch_ms=avg_s * 1e3, launches=n_launch, rays_per_launch=units / max(1, n_launch),
                         timing="one untimed probe step with overlap_shadow=0: every launch alone on the GPU, HIP events on its stream")
                if traffic:
                    d["achieved"] = traffic / avg_s / 1e9
                    d["frac"] = d["achieved"] / HBM_PEAK_GBPS
                    d["frac_of_copy_ceiling"] = d["achieved"] / HBM_COPY_CEILING_GBPS
                if alg_bytes:
                    d["algorithmic"] = dict(bytes_per_launch=alg_bytes / max(1, n_launch), bytes_per_ray=alg_bytes / max(1, units), GBps=alg_bytes / max(seconds, 1e-12) / 1e9,
                                            note="32 B x node tests + 36 B x triangle tests (oracle counters on the CPU sample) + ray/result records; served mostly from L1/LDS/L2 - not an HBM figure")
                # the unit that does bind: the CU's vector-memory path.  L1 (TCP) tag lookups per second of the
                # family against the rate the gather micro-benchmark reaches with an L1-resident table (same counter, same pass).
                acc = fam.get("tcp_accesses_per_launch")
                if acc and ceil_l1 and ceil_l1.get("tcp_accesses_per_s"):
                    # TCP_TOTAL_ACCESSES counts 64 per vector-memory wave-instruction whatever the EXEC mask, while the gather loop's cost
                    # follows the ACTIVE lanes (masked modes 3-6 of the micro-benchmark): scale by the kernels' VALU lane utilisation
                    util = fam.get("valu_lane_utilisation", 1.0)
                    rate = acc * util / avg_s
                    d["vector_memory"] = dict(achieved=rate * 1e-9, peak=ceil_l1["tcp_accesses_per_s"] * 1e-9, unit="G active-lane L1 accesses/s", frac=rate / ceil_l1["tcp_accesses_per_s"],
                                              lane_utilisation=util, source=ceil_l1["source"],
                                              note="estimate: TCP_TOTAL_ACCESSES x VALU lane utilisation (a proxy for the EXEC mask of the loads) against the rate of the gather micro-benchmark (table <= L2)")
                # TA (the CU's vector-memory address unit) busy cycles, summed over the 256 CUs, against the launch's shader-clock
                # cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs): how much of the time the vector-memory path is occupied
                if fam.get("ta_busy_cycles_per_launch") and fam.get("grbm_gui_active_per_launch"):
                    d["ta_busy"] = dict(achieved=fam["ta_busy_cycles_per_launch"] / 256.0, peak=fam["grbm_gui_active_per_launch"] / 8.0, unit="cycles per CU per launch",
                                        frac=(fam["ta_busy_cycles_per_launch"] / 256.0) / (fam["grbm_gui_active_per_launch"] / 8.0),
                                        addr_stalled_by_tc_frac=(fam.get("ta_addr_stalled_by_tc_per_launch", 0.0) / 256.0) / (fam["grbm_gui_active_per_launch"] / 8.0))
                if fam.get("l2_read_bytes_per_launch"):
                    d["l2"] = dict(achieved=fam["l2_read_bytes_per_launch"] / avg_s / 1e9, peak=L2_PEAK_GBPS, unit="GB/s", frac=fam["l2_read_bytes_per_launch"] / avg_s / 1e9 / L2_PEAK_GBPS)
                # `bound` names the roofline `frac` is taken against (HBM); `binding` names the unit the kernel is actually
                # short of — the CU's vector-memory path — with its two measures side by side
                d["binding"] = dict(unit="vector-memory (TA -> TCP -> L2 per-lane gathers)",
                                    ta_busy_frac=d["ta_busy"]["frac"] if "ta_busy" in d else None,
                                    active_lane_gather_frac=d["vector_memory"]["frac"] if "vector_memory" in d else None,
                                    valu_lane_utilisation=fam.get("valu_lane_utilisation"),
                                    note="ta_busy_frac: TA busy cycles per CU / launch cycles; active_lane_gather_frac: active-lane L1 accesses/s against the gather micro-benchmark's L1-resident rate")
                return d

            alg_closest = alg_any = None
            if counters:
                per = solo.rays
                alg_closest = (32.0 * counters["node_tests_per_ray"] + 36.0 * counters["shape_tests_per_ray"] + 48.0) * per
                alg_any = (32.0 * counters["shadow_node_tests_per_ray"] + 36.0 * counters["shadow_shape_tests_per_ray"]) * per + 36.0 * solo.shadow_rays
            fam_closest = family("k_trace_closest", "closest", ["k_trace_closest_pt", "k_trace_closest_packet"], solo.seconds_trace, solo.trace_launches, solo.rays, alg_closest, 32.0)
            fam_any = family("k_trace_any", "any", ["k_trace_any_pt", "k_trace_any_packet"], solo.seconds_shadow, solo.shadow_launches, solo.shadow_rays, alg_any, 36.0)
            roofline, other = (fam_any, fam_closest) if solo.seconds_shadow > solo.seconds_trace else (fam_closest, fam_any)
            roofline["other"] = other
            roofline["gather_ceiling"] = dict(l1_resident=ceil_l1, at_table_size=ceil_tab, table_bytes=table_bytes,
                                              note="per-lane 64-byte record gathers, tools/micro/gather_bench.hip")
            roofline["solo_step"] = dict(ms=solo.seconds_total * 1e3, closest_ms=solo.seconds_trace * 1e3, any_ms=solo.seconds_shadow * 1e3, shade_ms=solo.seconds_shade * 1e3)
            for f in (roofline, other):  # a fraction above 1 means the model is wrong: refuse to print it
                for k in ("frac",):
                    if f.get(k) is not None and f[k] > 1.0:
                        raise SystemExit(f"[bench] roofline {f['kernel']}.{k} = {f[k]:.3f} > 1")
                for sub in ("vector_memory", "l2", "ta_busy"):
                    if sub in f and f[sub]["frac"] > 1.0:
                        raise SystemExit(f"[bench] roofline {f['kernel']}.{sub}.frac = {f[sub]['frac']:.3f} > 1")
        out = {
            "metric": "Mray/s (primary+secondary), Path integrator @1080p" if wl["res"] == (1920, 1080) else f"Mray/s (primary+secondary), {'Whitted' if wl.get('whitted') else 'Path'} integrator @{wl['res'][0]}x{wl['res'][1]}",
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
                       "gather": {"abi": "yk_dist_gather (C ABI: ncclSend/ncclRecv on the context stream)", None: None}.get(gather_mode, gather_mode),
                       "frames_in_flight": in_flight},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "extra": {"rays_per_step": rays_all // args.steps, "shadow_rays_per_step": shadow_all // args.steps,
                      "shadow_Mray_per_s": shadow_all / elapsed * 1e-6, "rank0_device_s_per_step": t_dev / args.steps,
                      "rank0_trace_s": t_trace / args.steps, "rank0_shadow_s": t_shadow / args.steps, "rank0_shade_s": t_shade / args.steps,
                      "two_in_flight": two_in_flight, "film_mean_rgb": film_mean, "bvh_build_s": info.build_seconds, "scene_upload_s": info.upload_seconds,
                      "scene_load_s": gen_s if args.scene_file else None,
                      "steps_mode": "asynchronous (no host sync inside the timed region; per-kernel times and ray counts from one untimed probe step)" if async_steps
                      else "synchronous (per-kernel HIP-event times read after every step of the timed region)"},
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    for sl in slots:  # communicators first: they hold their context
        if sl.get("dist"):
            sl["dist"].close()
    scene.close()
    for sl in slots:
        sl["ctx"].close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
