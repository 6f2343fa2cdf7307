#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame for the 1080p octree mesh scene (Crown + Vase + Crystal3, hard shadows,
reflections with maxDepth 2, reflective floor) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step is one frame: every rank renders the 32x32 tiles it owns (tile i -> rank i % N) of the same frame from its own
resident copy of the scene, the tile buffers are gathered to rank 0 over RCCL, and rank 0 reassembles the frame.
The scene, octree and output buffers are resident in HBM before the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
FP32_PEAK_TFLOPS = 157.3   # MI355X FP32 vector peak


WORKLOADS = {
    1: "config1: reference scene 1 (three spheres, box, floor) with the X preset, {w}x{h}",
    2: "config2: cube + Chalice procedural meshes, {w}x{h}, primary + hard-shadow rays, mesh detail {detail}",
    3: "config3: Crown + Vase + Crystal3 procedural meshes, {w}x{h}, hard shadows, reflections maxDepth {depth} with reflective floor, octree depth 8",
    4: "config4: nine-model room (Table, BookCaseLargeBooks, Chest_gold, Pizza, Pug, Chalice, Vase, Crown, Book2_Open), {w}x{h}, "
       "hard shadows, reflections maxDepth {depth} with reflective floor, mesh detail {detail}",
    5: "config5: field of 144 baked Chicken / WoodLog copies, {w}x{h}, hard shadows, reflections maxDepth {depth} with reflective "
       "floor, mesh detail {detail}",
}


def load_package():
    name = "cuda_ray_tracer_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "cuda-ray-tracer_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


INFINITY_CACHE_BYTES = 256e6   # a scene that fits is served by the L2s and the Infinity Cache: HBM is then not the roof


def algorithmic_bytes_per_ray(st, pixels):
    """128 B per octree node visited (one record = one cache line holding the eight quantised child boxes, the child
    reference and the leaf references), 36 B per triangle tested (three positions), 108 B per closest-hit attribute fetch
    (3 vertices x normal, Ka, Kd), 3 B of frame per pixel."""
    rays = max(int(st.rays), 1)
    return (128.0 * st.node_visits + 36.0 * st.tri_tests + 108.0 * st.attr_fetches + 3.0 * pixels) / rays


def algorithmic_flops_per_ray(st, info, pixels):
    """26 flop per child box and per model gate evaluated, 51 per triangle test, 30/14/26 per sphere/plane/box test made on the
    linear tables (counted by the instrumented kernel: tests whose result the reference discards are not made and not
    counted), ~120 per shaded hit (primary + reflection levels ~ rays/2)."""
    rays = max(int(st.rays), 1)
    return (26.0 * st.node_tests + 51.0 * st.tri_tests + 26.0 * st.gate_tests + float(st.table_flops)) / rays + 60.0


def cpu_baseline(scenes, built, cfg, seconds_target=15.0):
    """Times the CPU oracle (the reference's algorithm: box gate + every triangle) on a bounded sample: the same scene
    and camera at a reduced resolution chosen so the run takes roughly `seconds_target` seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    O = oracle_lib.Oracle()
    threads = min(16, os.cpu_count() or 1)
    # probe at a tiny size to pick the sample size
    t0 = time.perf_counter()
    _, _, _, rays, _ = O.render(built.desc, built.camera, cfg, 48, 27, threads=threads, want_ids=False)
    probe = max(time.perf_counter() - t0, 1e-3)
    scale = max(1.0, min(40.0, (seconds_target / probe) ** 0.5))
    w, h = int(48 * scale), int(27 * scale)
    t0 = time.perf_counter()
    sample_rgb, _, _, rays, _ = O.render(built.desc, built.camera, cfg, w, h, threads=threads, want_ids=False)
    dt = time.perf_counter() - t0
    if dt < 8.0:      # the probe under-estimated: one more pass at a size that lands near 12 s
        grow = min(4.0, (12.0 / max(dt, 1e-3)) ** 0.5)
        w, h = int(w * grow), int(h * grow)
        t0 = time.perf_counter()
        sample_rgb, _, _, rays, _ = O.render(built.desc, built.camera, cfg, w, h, threads=threads, want_ids=False)
        dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"same scene and camera rendered at {w}x{h} ({rays} rays, {dt:.1f} s) by the CPU oracle, "
                      f"box gate + brute-force triangle loop as the reference does, OpenMP over scanlines"}, (w, h, sample_rgb, rays)


def _pose(scenes, spec, k, frames):
    pos, yaw, pitch = scenes.camera_path_pose(spec, k, frames)
    return pos, spec.fov, yaw, pitch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, help="benchmark configuration (3 = headline 1080p mesh scene)")
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--leaf", type=int, default=0, help="octree leaf size override (0 = library default)")
    ap.add_argument("--depth", type=int, default=0, help="octree depth cap override (0 = library default)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="consecutive frames alternate over this many HIP streams (one renderer context each), so the start of a "
                         "frame overlaps the slow tail of the frame before it; 0 = 1 on one GPU (the roofline line is defined on "
                         "launches that do not overlap), 2 on several GPUs (a rank's share of a frame is as long as its heaviest "
                         "8x8 block and leaves most of the GPU idle)")
    ap.add_argument("--tile-slots", type=int, default=3, help="several GPUs: tile-buffer slots per rank (gathers that can be in flight or being filled)")
    ap.add_argument("--frames-per-gather", type=int, default=0, help="several GPUs: consecutive frames whose tiles one gather moves (0 = 4 on two GPUs, 8 on four, 16 from eight on: a launch costs at least the run time of its heaviest block, however little it renders)")
    ap.add_argument("--tile-deal", type=str, default="auto",
                    help="several GPUs: how tiles are dealt to the ranks, 'k0,k1' = rank 0 gets k0 and every other rank k1 of each period of consecutive tiles "
                         "(1,1 = plain interleaving); auto = rank 0's share reduced by what gathering and reassembling a frame costs it (--reassembly-share)")
    ap.add_argument("--reassembly-share", type=float, default=0.027,
                    help="time rank 0 spends on the gather and the reassembly of a frame, as a fraction of the time one GPU needs to render the frame "
                         "(measured on one GPU for the 1080p benchmark frame: 6 of 224 us)")
    ap.add_argument("--device-build", type=int, default=0, help="1: build the octrees on the GPU (0.2 s for the 10 M triangles of config 5, frames 5 %% slower), 0: on the host (1.8 s)")
    ap.add_argument("--path-frames", type=int, default=120,
                    help="also render this many frames of the benchmark's moving-camera path (scenes.camera_path_pose): once frame by "
                         "frame (frame state + one launch per frame) and once through the camera-path entry point (one launch per chunk); "
                         "0 = skip")
    ap.add_argument("--latency-steps", type=int, default=-1,
                    help="several GPUs: frames of the latency leg - one frame per gather, one frame in flight, the host waits for every "
                         "frame as the reference's loop does (launch, synchronise, present); -1 = as many as --steps (at most 100), 0 = skip")
    ap.add_argument("--path-chunk", type=int, default=0, help="frames per camera-path launch (0 = 30 on one GPU, the gather batch on several)")
    ap.add_argument("--launch-per-gather", type=int, default=-1,
                    help="several GPUs: 1 = the frames of one gather batch are rendered by ONE launch (the camera-path entry point, one "
                         "camera per frame) instead of one launch per frame; -1 = on when frames are gathered in batches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-leg", action="store_true",
                    help="skip the short extra leg that repeats the measurement with the other number of frames in flight (1 <-> 2); "
                         "profiling runs use this so that every launch of the run is of one kind")
    ap.add_argument("--out", type=str, default="", help="write the last frame as PNG (rank 0)")
    args = ap.parse_args()
    # Rank 0 prints exactly one line on stdout.  Libraries underneath write there too (RCCL prints a version banner at its
    # first collective), so file descriptor 1 points at stderr for the rest of the run and the result goes to the saved one.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    # A rank that is still running after this many seconds dumps its Python stacks and exits, so that a stalled exchange ends
    # the job instead of holding the node (several ranks: 600 s unless RT_BENCH_WATCHDOG says otherwise; 0 = never).
    watchdog = float(os.environ.get("RT_BENCH_WATCHDOG", "600" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "0"))
    if watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the render path has no CPU fallback)")
    # RT_BENCH_BACKEND=gloo rehearses the N-rank path on fewer GPUs than ranks (tiles are staged through host memory for
    # the gather); the measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    # RT_BENCH_TILED=1 runs the N-rank code path (tile buffer, gather, reassembly) with a single rank: measures what the
    # exchange adds on the host and the GPU when only one GPU is at hand.
    tiled = world > 1 or os.environ.get("RT_BENCH_TILED", "0") == "1"
    if tiled:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    load_package()
    from cuda_ray_tracer_amd import binding, scenes

    spec, cfg, w, h = scenes.config_scene(args.config, detail=args.detail)
    if args.width and args.height:
        w, h = args.width, args.height
    device_build = args.device_build == 1
    built = scenes.build_scene(spec, octree_max_depth=args.depth, octree_leaf_triangles=args.leaf, build_on_device=device_build)
    r = binding.Renderer(device_index)
    r.upload(built.desc)
    r.set_state(built.camera, cfg)
    info = r.scene_info()
    pixels = w * h

    # deterministic per-frame counters from the instrumented kernel (outside the timed region)
    st = r.count_rays(w, h) if rank == 0 else None

    stream = torch.cuda.current_stream().cuda_stream
    frame = torch.empty(pixels * 3, dtype=torch.uint8, device=dev)
    # RT_BENCH_EMULATE_WORLD=n with one tiled rank: that rank renders, gathers and reassembles the share it would have among
    # n ranks (the other shares stay black) - what one rank of an n-GPU run costs on the host and on the GPU.
    share = world
    if tiled and world == 1 and int(os.environ.get("RT_BENCH_EMULATE_WORLD", "0")) > 1:
        share = int(os.environ["RT_BENCH_EMULATE_WORLD"])
    from cuda_ray_tracer_amd import sharding
    deal = sharding.balanced_deal(share, args.reassembly_share) if args.tile_deal == "auto" else tuple(int(v) for v in args.tile_deal.split(","))
    if share == 1:
        deal = (1, 1)
    r.set_tile_deal(*deal)
    in_flight = args.frames_in_flight if args.frames_in_flight > 0 else (2 if share > 1 else 1)
    # The headline leg runs with `in_flight` frames in flight; a second, shorter leg repeats the measurement with the other of
    # 1 and 2, so that results at every GPU count can be compared like for like (ms_per_step_by_frames_in_flight).
    other_flight = 2 if in_flight == 1 else 1
    second_leg = not args.no_second_leg and share == world
    flight = [in_flight]
    lane_count = max(in_flight, other_flight) if second_leg else in_flight
    if os.environ.get("RT_BENCH_OWN_STREAM", "1" if lane_count > 1 else "0") == "1":
        # Several frames in flight: every lane on a stream of its own.  Work on the default stream does not overlap with work
        # on the other streams (measured: a rank's share of 1/8 of the frame, two lanes: 0.184 ms per frame with lane 0 on the
        # default stream, 0.100 ms on its own stream).
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
        stream = torch.cuda.current_stream().cuda_stream
    lanes = [(r, torch.cuda.current_stream(), frame)]
    last_frame = [frame]       # where the newest complete frame is
    for _ in range(lane_count - 1):
        extra = binding.Renderer(device_index)
        extra.upload(built.desc)
        extra.set_state(built.camera, cfg)
        extra.set_tile_deal(*deal)
        lanes.append((extra, torch.cuda.Stream(device=dev), None if tiled else torch.empty(pixels * 3, dtype=torch.uint8, device=dev)))
    if tiled:
        # A slot holds the tile buffers of `batch` consecutive frames, which one gather moves (the host cost of a collective
        # call is about that of rendering a frame's share).  Several slots: the gather of a slot (RCCL's own stream) overlaps
        # the render of the frames after it; a slot is rendered into again only when its gather has finished, so their number
        # bounds how many frames the pipeline holds.
        batch = max(1, args.frames_per_gather if args.frames_per_gather > 0 else ((16 if share >= 8 else 8 if share >= 4 else 4) if share > 1 else 1))
        slots = max(2, args.tile_slots)
        tile_bytes = binding.tile_buffer_bytes(w, h, share, deal)
        slot_bytes = tile_bytes * batch
        tiles = [torch.empty(slot_bytes, dtype=torch.uint8, device=dev) for _ in range(slots)]
        gathered = [torch.zeros(slot_bytes * share, dtype=torch.uint8, device=dev) for _ in range(slots)] if rank == 0 else [None] * slots
        gather_list = [list(g.chunk(share))[:world] for g in gathered] if rank == 0 else [None] * slots
        if backend != "nccl":
            host_tiles = torch.empty(slot_bytes, dtype=torch.uint8)
            host_gathered = torch.empty(slot_bytes * world, dtype=torch.uint8) if rank == 0 else None
            host_list = list(host_gathered.chunk(world)) if rank == 0 else None

    kernel_events = []
    host_parts = [0.0, 0.0, 0.0]    # host seconds spent issuing: render, completion of the previous gather, gather
    pending = []     # (slot, frames in it, async gather handle) of the gathers still in flight
    if tiled:
        # The exchange (gather, reassembly on rank 0) runs on a stream of its own, so that the render streams go from
        # frame to frame without waiting: they only wait before overwriting a slot whose gather has not finished.
        xs = torch.cuda.Stream(device=dev) if os.environ.get("RT_BENCH_XSTREAM", "1") == "1" else torch.cuda.current_stream()
        batch_frames = torch.empty(batch * pixels * 3, dtype=torch.uint8, device=dev) if rank == 0 else None     # the reassembled frames of one batch
        rendered_ev = [[torch.cuda.Event() for _ in range(batch)] for _ in range(slots)]
        free_ev = [None] * slots

    def finish_pending():
        while pending:
            slot, frames, work = pending.pop(0)
            with torch.cuda.stream(xs):
                if work is not None:
                    work.wait()                  # the exchange stream waits for the gather; the host does not
                if free_ev[slot] is None:
                    free_ev[slot] = torch.cuda.Event()
                free_ev[slot].record(xs)
                if rank == 0:
                    # rank q's buffer of the slot's j-th frame starts at q * slot_bytes + j * tile_bytes; the frames of the batch
                    # are reassembled by one launch (beside render launches that fill the GPU, a reassembly kernel mostly waits
                    # for registers to come free - once per launch)
                    r.untile_batch_device(w, h, share, gathered[slot].data_ptr(), slot_bytes, tile_bytes, frames, batch_frames.data_ptr(), xs.cuda_stream)
                    last_frame[0] = batch_frames.narrow(0, (frames - 1) * pixels * 3, pixels * 3)

    # Several GPUs: a rank's share of a frame is short, and issuing it costs the host as much as rendering it costs the GPU.
    # With frames gathered in batches anyway, the whole batch goes out as one launch (rt_render_path_tiles_device).
    batch_launch = [tiled and batch > 1 and args.launch_per_gather != 0]
    step_index = [0]
    event_period = int(os.environ.get("RT_BENCH_EVENT_PERIOD", "4"))     # every n-th launch is timed with events (0 = none)

    def sample_events(k):
        return event_period > 0 and k % event_period == 0


    def step(record):
        if not tiled:
            ctx, s, buf = lanes[step_index[0] % flight[0]]
            record = record and sample_events(step_index[0])
            step_index[0] += 1
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
            ctx.render_device(w, h, buf.data_ptr(), s.cuda_stream)
            if record:
                e1.record(s)
                kernel_events.append((e0, e1))
            return
        slot, j = (step_index[0] // batch) % slots, step_index[0] % batch
        if batch_launch[0]:
            step_index[0] += 1
            if j == batch - 1:
                issue_batch(slot, batch, record)
            return
        ctx, s, _ = lanes[step_index[0] % flight[0]]
        step_index[0] += 1
        h0 = time.perf_counter()
        if free_ev[slot] is not None:
            s.wait_event(free_ev[slot])
        record = record and sample_events(step_index[0])
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
        ctx.render_tiles_device(w, h, rank, share, tiles[slot].data_ptr() + j * tile_bytes, tile_bytes, s.cuda_stream)
        if record:
            e1.record(s)
            kernel_events.append((e0, e1))
        rendered_ev[slot][j].record(s)
        host_parts[0] += time.perf_counter() - h0
        if j == batch - 1:
            exchange(slot, batch)

    def issue_batch(slot, frames, record):
        """One launch for `frames` consecutive frames of this rank's tiles into `slot`, then the slot's gather."""
        batch_no = (step_index[0] - 1) // batch
        ctx, s, _ = lanes[batch_no % flight[0]]
        h0 = time.perf_counter()
        if free_ev[slot] is not None:
            s.wait_event(free_ev[slot])
        record = record and sample_events(batch_no)
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
        ctx.render_path_tiles_device(w, h, rank, share, [built.camera] * frames, tiles[slot].data_ptr(), tile_bytes, s.cuda_stream)
        if record:
            e1.record(s)
            kernel_events.append((e0, e1, frames))
        for j in range(frames):
            rendered_ev[slot][j].record(s)
        host_parts[0] += time.perf_counter() - h0
        exchange(slot, frames)

    def exchange(slot, frames):
        h1 = time.perf_counter()
        finish_pending()                         # the gather before this one ran while this slot was being rendered
        h2 = time.perf_counter()
        with torch.cuda.stream(xs):
            for j in range(frames):
                xs.wait_event(rendered_ev[slot][j])
            if backend == "nccl":
                work = dist.gather(tiles[slot], gather_list[slot], dst=0, async_op=True)
            else:
                host_tiles.copy_(tiles[slot])
                dist.gather(host_tiles, host_list, dst=0)
                if rank == 0:
                    gathered[slot].copy_(host_gathered)
                work = None
        pending.append((slot, frames, work))
        host_parts[1] += h2 - h1; host_parts[2] += time.perf_counter() - h2

    def fence():
        if tiled:
            if step_index[0] % batch != 0:       # a slot that is only partly filled: gather it as it is
                if batch_launch[0]:
                    issue_batch((step_index[0] // batch) % slots, step_index[0] % batch, False)
                else:
                    exchange((step_index[0] // batch) % slots, step_index[0] % batch)
                step_index[0] += batch - step_index[0] % batch
            finish_pending()
            if os.environ.get("RT_BENCH_DEBUG_HANG"):
                t_end = time.perf_counter() + 5.0
                while time.perf_counter() < t_end and not all(l[1].query() for l in lanes) or not xs.query():
                    if time.perf_counter() >= t_end:
                        break
                    time.sleep(0.01)
                print("lanes done:", [l[1].query() for l in lanes], "xs done:", xs.query(), "rendered:", [[e.query() for e in evs] for evs in rendered_ev],
                      "free:", [e.query() if e is not None else None for e in free_ev], "steps issued:", step_index[0], file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    # A context orders its tiles (and chooses the blocks to split) from the costs its previous frames recorded, ranked after
    # every eighth frame: forty frames per context (8 ms of config 3) bring that feedback to its steady state before the warm-up
    # steps the caller asked for (with eight, twenty timed steps after five warm-up steps took 0.218 instead of 0.209 ms each).
    # The choice of blocks to split is looked at every 32nd frame while nothing is split: a frame that profits from splitting
    # (config 2) is in its steady state after three such looks.
    for ctx, s, buf in lanes:
        for _ in range(104):
            if tiled:
                ctx.render_tiles_device(w, h, rank, share, tiles[0].data_ptr(), tile_bytes, s.cuda_stream)
            else:
                ctx.render_device(w, h, buf.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    if tiled and backend == "nccl":
        # RCCL sets up its communicator and peer connections at the first collective of a kind: do that before any step
        dist.gather(tiles[0], gather_list[0], dst=0)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    enqueue = time.perf_counter() - t0       # host time to issue the steps (the GPU may still be busy)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # per frame: a launch that renders several frames counts with that many
    kernel_ms = (sum(ev[0].elapsed_time(ev[1]) for ev in kernel_events) / sum(ev[2] if len(ev) > 2 else 1 for ev in kernel_events)
                 if kernel_events else elapsed * 1e3 / args.steps)
    by_flight = {str(in_flight): elapsed * 1e3 / args.steps}
    # What the timed frames look like: the last one, as rank 0 holds it (copied now, before any other leg renders again).
    timed_frame = last_frame[0].cpu().numpy().reshape(h, w, 3).copy() if rank == 0 else None
    if second_leg:
        flight[0] = other_flight
        steps2 = max(1, min(args.steps, 200))
        for _ in range(max(args.warmup, 2 * other_flight)):
            step(False)
        fence()
        t1 = time.perf_counter()
        for _ in range(steps2):
            step(False)
        fence()
        elapsed2 = time.perf_counter() - t1
        if world > 1:
            tmax = torch.tensor([elapsed2], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed2 = float(tmax.item())
        by_flight[str(other_flight)] = elapsed2 * 1e3 / steps2
        flight[0] = in_flight

    # Latency mode on several GPUs: the reference keeps ONE frame in flight (RayTracer.cpp:405-417: launch, synchronise, then the
    # host reads the frame).  One frame per gather, no batching, the host waits for the reassembled frame before it issues the next.
    latency_ms = None
    latency_steps = (min(args.steps, 100) if args.latency_steps < 0 else args.latency_steps) if tiled else 0
    if latency_steps > 0:
        fence()
        ctx0, s0, _ = lanes[0]
        lat_tiles = torch.empty(tile_bytes, dtype=torch.uint8, device=dev)
        lat_gathered = torch.zeros(tile_bytes * share, dtype=torch.uint8, device=dev) if rank == 0 else None
        lat_list = list(lat_gathered.chunk(share))[:world] if rank == 0 else None
        lat_frame = torch.empty(pixels * 3, dtype=torch.uint8, device=dev) if rank == 0 else None
        if backend != "nccl":
            lat_host = torch.empty(tile_bytes, dtype=torch.uint8)
            lat_host_all = torch.empty(tile_bytes * world, dtype=torch.uint8) if rank == 0 else None
            lat_host_list = list(lat_host_all.chunk(world)) if rank == 0 else None

        def latency_frame():
            ctx0.render_tiles_device(w, h, rank, share, lat_tiles.data_ptr(), tile_bytes, s0.cuda_stream)
            with torch.cuda.stream(s0):
                if backend == "nccl":
                    dist.gather(lat_tiles, lat_list, dst=0)
                else:
                    s0.synchronize()
                    lat_host.copy_(lat_tiles)
                    dist.gather(lat_host, lat_host_list, dst=0)
                    if rank == 0:
                        lat_gathered.narrow(0, 0, tile_bytes * world).copy_(lat_host_all)
                if rank == 0:
                    r.untile_device(w, h, share, lat_gathered.data_ptr(), lat_frame.data_ptr(), s0.cuda_stream)
            torch.cuda.synchronize()

        for _ in range(max(3, args.warmup)):
            latency_frame()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(latency_steps):
            latency_frame()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        latency_ms = dt * 1e3 / latency_steps
        if rank == 0 and share == world:
            latency_frame_equal = bool(torch.equal(lat_frame, last_frame[0].reshape(-1).to(lat_frame.device)))
        else:
            latency_frame_equal = None

    # The moving-camera frame loop of the reference (Window.cpp:339-421), on the benchmark's fixed camera path.
    path_result = None
    if args.path_frames > 0:
        n_path = args.path_frames
        cams = [binding.camera_look(*_pose(scenes, spec, k, n_path)) for k in range(n_path)]
        ctx0, s0, _ = lanes[0]
        flight[0] = 1
        if not tiled:
            chunk = max(1, min(args.path_chunk if args.path_chunk > 0 else 30, n_path))
            frames_buf = torch.empty(chunk * pixels * 3, dtype=torch.uint8, device=dev)
            def loop_pass():
                for k in range(n_path):
                    ctx0.set_state(cams[k], cfg)
                    ctx0.render_device(w, h, frames_buf.data_ptr() + (k % chunk) * pixels * 3, s0.cuda_stream)
            def path_pass():
                for k0 in range(0, n_path, chunk):
                    ctx0.render_path_device(w, h, cams[k0:k0 + chunk], frames_buf.data_ptr(), s0.cuda_stream)
        else:
            chunk = batch
            def loop_pass():
                keep = batch_launch[0]
                batch_launch[0] = False              # one frame state and one launch per frame
                for k in range(n_path):
                    ctx0.set_state(cams[k], cfg)
                    step(False)
                fence()
                batch_launch[0] = keep
            def path_pass():
                for k0 in range(0, n_path, chunk):
                    part = cams[k0:k0 + chunk]
                    slot = (step_index[0] // batch) % slots
                    if free_ev[slot] is not None:
                        s0.wait_event(free_ev[slot])
                    ctx0.render_path_tiles_device(w, h, rank, share, part, tiles[slot].data_ptr(), tile_bytes, s0.cuda_stream)
                    for j in range(len(part)):
                        rendered_ev[slot][j].record(s0)
                    step_index[0] += batch
                    exchange(slot, len(part))
        timings = {}
        for name, fn in (("frame_by_frame", loop_pass), ("one_launch_per_chunk", path_pass)):
            fn(); fence()                       # warm-up pass: tile order under motion, allocations
            t1 = time.perf_counter()
            fn(); fence()
            dt = time.perf_counter() - t1
            if world > 1:
                tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            timings[name] = dt * 1e3 / n_path
        ctx0.set_state(built.camera, cfg)
        flight[0] = in_flight
        path_result = {"frames": n_path, "frames_per_launch": chunk, "ms_per_frame": timings,
                       "note": "camera moves every frame (yaw sweep + dolly); tile order of a frame comes from the costs of the frames before it"}

    if rank == 0:
        import zlib
        import numpy as np
        ms_per_step = elapsed * 1e3 / args.steps
        rays = int(st.rays)
        # one rank rehearsing its share of an n-rank run traces 1/share of the frame's rays per step
        mrays = (rays / (share / world)) / (ms_per_step * 1e-3) / 1e6
        b_ray = algorithmic_bytes_per_ray(st, pixels)
        f_ray = algorithmic_flops_per_ray(st, info, pixels)
        # the render kernel of this rank traces ~1/share of the frame's rays per launch
        launch_rays = rays / share
        achieved_gbs = b_ray * launch_rays / (kernel_ms * 1e-3) / 1e9
        achieved_tflops = f_ray * launch_rays / (kernel_ms * 1e-3) / 1e12
        # Which roof: a scene that fits the 256 MB Infinity Cache is read from HBM about once per frame whatever the rays do,
        # and the kernel is bound by vector-instruction issue (FP32 VALU peak); a larger scene is priced against HBM.
        # A larger scene is priced against HBM - unless its algorithmic bytes per second exceed what HBM can deliver, which
        # proves that the rays of a frame are served from the caches (they touch the visible part of the scene again and again).
        resident = int(info.deviceBytes) <= INFINITY_CACHE_BYTES or achieved_gbs > HBM_PEAK_GBS
        # The timed frames are checked, not only timed: the last one must equal, byte for byte, a frame of the same scene and
        # camera rendered outside the timed region through the blocking host entry point (one launch, one GPU).
        check = {"frame_crc32": zlib.crc32(timed_frame.tobytes()) & 0xFFFFFFFF, "frame_nonzero_fraction": float((timed_frame != 0).mean())}
        if share == world:
            untimed, _ = r.render(w, h)
            check["equals_untimed_render"] = bool(np.array_equal(untimed, timed_frame))
        traffic = valu_insts = lane_util = gpu_cycles = None
        # HBM bytes per launch from the PMC counters of an earlier profiling run of this workload (tools/profile_all.sh)
        tpath = os.path.join(ROOT, "profiles", f"traffic_config{args.config}.json")
        if not os.path.exists(tpath):
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("workload") == f"config{args.config}" and world == 1 and (w, h) == tuple(tj.get("size", ())) and (args.config != 5 or abs(args.detail - 7.0) < 1e-9):
                    traffic = tj.get("hbm_bytes_per_launch")
                    valu_insts = tj.get("valu_instructions_per_launch")
                    lane_util = tj.get("valu_lane_utilisation")
                    gpu_cycles = tj.get("gpu_cycles_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s + ms/frame, 1080p octree mesh scene, 1/2/4/8 MI355X",
            "value": mrays, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.config].format(w=w, h=h, depth=cfg.maxDepth, detail=args.detail),
                       "triangles": int(info.triangleCount), "octree_nodes": int(info.nodeCount), "rays_per_frame": rays,
                       "rays_per_pixel": rays / pixels, "tile": binding.RT_TILE, "octrees_built_on": "gpu" if device_build else "host",
                       "frames_in_flight": in_flight, "frames_per_gather": batch if tiled else None, "tile_deal": list(deal) if tiled else None,
                       "frames_per_launch": (batch if batch_launch[0] else 1) if tiled else 1,
                       "launches_overlap": in_flight > 1,
                       "parallelism": ("1 GPU" if not tiled else f"1 GPU through the tile / gather / reassembly path (rehearsal, share of {share} ranks)") if world == 1 else f"image tiles over {world} GPUs, scene replicated, RCCL gather to rank 0"},
            "kernel_ms": kernel_ms, "host_issue_ms_per_step": enqueue * 1e3 / args.steps,
            "host_issue_parts_ms": {k: v * 1e3 / (args.steps + args.warmup) for k, v in zip(("render", "complete_previous", "gather"), host_parts)} if tiled else None,
            "ms_per_step_by_frames_in_flight": by_flight,
            "roofline": ({"bound": "valu", "achieved": achieved_tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": achieved_tflops / FP32_PEAK_TFLOPS} if resident else
                         {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved_gbs / HBM_PEAK_GBS}),
            "check": check,
        }
        if path_result is not None:
            out["camera_path"] = path_result
        if latency_ms is not None:
            # beside the throughput figure above (frames gathered in batches, two batches in flight): what one frame takes from
            # launch to reassembled frame with nothing else in flight, as the reference's own loop runs
            out["ms_per_frame_latency"] = latency_ms
            out["latency_mode"] = {"frames": latency_steps, "frames_per_gather": 1, "frames_in_flight": 1, "host_waits_every_frame": True,
                                   "mrays_per_s": (rays / (share / world)) / (latency_ms * 1e-3) / 1e6,
                                   "frame_equals_throughput_mode": latency_frame_equal}
        out["roofline"].update({
            "traffic": traffic, "scene_bytes": int(info.deviceBytes),
            "why_this_bound": (("scene resident in the 256 MB Infinity Cache: HBM is read about once per frame, the kernel is bound by FP32 "
                                "vector issue" if int(info.deviceBytes) <= INFINITY_CACHE_BYTES else
                                "scene larger than the Infinity Cache, but the algorithmic bytes per second exceed the HBM peak: the rays are "
                                "served from the caches, the kernel is bound by FP32 vector issue") if resident
                               else "scene larger than the 256 MB Infinity Cache: priced against HBM bandwidth"),
            "flops_per_ray": f_ray, "bytes_per_ray": b_ray, "algorithmic_gbs": achieved_gbs, "algorithmic_tflops": achieved_tflops,
            "algorithmic_gbs_over_hbm_peak": achieved_gbs / HBM_PEAK_GBS,
            "hbm_frac_measured": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "node_visits_per_ray": st.node_visits / rays, "node_tests_per_ray": st.node_tests / rays,
            "tri_tests_per_ray": st.tri_tests / rays, "attr_fetches_per_ray": st.attr_fetches / rays})
        # What is left, said in the kernel's own currency.  The arithmetic contract (no contraction, correctly rounded division
        # and square root) keeps the kernel off the packed-fma peak that `frac` is priced against; what it can be measured by is
        # the share of the SIMDs' issue slots its vector instructions fill: instructions per launch x 4 cycles per wave64
        # instruction / (1024 SIMDs x shader cycles of the launch), both from the PMC pass of this workload
        # (profiles/traffic_config*.json; the clock is the one the GPU really ran at: cycles / launch duration, about 2.5 GHz).
        # 1 - issue_frac is idle issue time; 1 - lane_utilisation is the share of lanes switched off in the instructions that
        # do issue.  useful_flops_per_ray leaves out the traversal's speculative visits (subtrees entered for entries of a ray
        # that were pending side by side and that the hit found meanwhile would have pruned): counted by a second pass of the
        # instrumented kernel that is bounded by each ray's final hit distance from the start.
        useful_f_ray = (26.0 * st.useful_node_tests + 51.0 * st.useful_tri_tests + 26.0 * st.gate_tests + float(st.table_flops)) / max(rays, 1) + 60.0
        out["roofline"].update({
            "useful_flops_per_ray": useful_f_ray,
            "useful_frac": useful_f_ray * launch_rays / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
            "speculative_node_tests": 1.0 - st.useful_node_tests / max(int(st.node_tests), 1),
            "speculative_tri_tests": 1.0 - st.useful_tri_tests / max(int(st.tri_tests), 1),
            "issue_frac": (valu_insts * 4.0 / (1024.0 * gpu_cycles)) if valu_insts and gpu_cycles else None,
            "shader_clock_ghz": (gpu_cycles / (kernel_ms * 1e-3) / 1e9) if gpu_cycles else None,
            "lane_utilisation": lane_util,
            "valu_instructions_per_launch": valu_insts})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], (sw, sh, sample_rgb, sample_rays) = cpu_baseline(scenes, built, cfg)
            # the CPU leg's frame doubles as a check of the GPU path on this very scene and camera
            g_rgb, g_st = r.render(sw, sh)
            diff = np.abs(g_rgb.astype(np.int16) - sample_rgb.astype(np.int16)).max(axis=-1)
            check["vs_cpu_sample"] = {"size": [sw, sh], "pixels_beyond_1_lsb": int((diff > 1).sum()), "pixels_differing": int((diff > 0).sum()),
                                      "ray_count_equal": int(g_st.rays) == int(sample_rays)}
        if args.out:
            from PIL import Image
            Image.fromarray(timed_frame).save(args.out)
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    for ctx, _, _ in lanes:
        ctx.close()
    if tiled:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
