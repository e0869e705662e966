#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X sphere-tracing render path.

    python bench.py --gpus N --steps K --warmup W
(N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`,
one rank per GPU, RCCL through torch.distributed.)

A step = one frame of BASELINE.json's metric workload (C3: Dense Sphere Grid, 125 spheres,
3840x2160, sphere tracing + BVH, iteration-heatmap shader): render + fused shade into
{depth, normal, sdfEval, iters, RGBA} resident in HBM, then the diagnostics reduction of
main.ts:528-548 on the device.  At N > 1 the frame's rows are sharded over the ranks
(interleaved 16-row stripes); every rank reduces the counters of its own rows, and one gather
brings its RGBA rows and its 32-byte partial diagnostics to rank 0, which reassembles the frame
and combines the partial sums -- total work fixed, so scaling is "strong".

Frames are independent, so `--frames-in-flight S` (default 12) enqueues consecutive frames on S HIP streams with
S buffer sets: the tail of a frame's persistent kernel -- its slowest rays, ~0.3 ms during which most CUs idle --
overlaps the following frames.  Every frame is still rendered, shaded and reduced in full; S = 1 is strictly serial.
With frames in flight a launch uses one persistent workgroup per CU instead of four (the other frames' workgroups
fill the CU) and the process asks the HIP runtime for sixteen hardware queues (GPU_MAX_HW_QUEUES, default four) so
that the streams do not share queues.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- HBM: 12 B/pixel of mandatory output / render-kernel time (HIP events)
  cpu_baseline -- the oracle (C restatement, kind "port") on the host cores, rank 0, N = 1 (the only leg that loads oracle/)
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime starts: a hardware queue per stream in flight (default 4)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_PIXEL = 12  # depth 1 + normal 3 + sdfEval 2 + iters 2 + RGBA 4, ~0 read

WORKLOADS = {
    "C2": dict(name="C2: Grid of Spheres (9), 1920x1080, sphere-tracing + BVH, Phong shader",
               preset=2, accel="BVH", width=1920, height=1080, shader="phong"),
    "C3": dict(name="C3: Dense Sphere Grid (125 spheres), 3840x2160, sphere-tracing + BVH, iteration-heatmap shader",
               preset=3, accel="BVH", width=3840, height=2160, shader="iteration-heatmap"),
    "C5": dict(name="C5: synthetic 10000 random spheres (splitmix64 0x5EED5EED), 3840x2160, sphere-tracing + Octree, "
                    "iteration-heatmap shader",
               synthetic=10000, accel="Octree", width=3840, height=2160, shader="iteration-heatmap"),
    # boxes / tori / rotated transforms (not BASELINE configs; same frame size as C3)
    "N3": dict(name="N3: Pyramid of Boxes (preset 9), 3840x2160, sphere-tracing + BVH, Phong shader",
               preset=9, accel="BVH", width=3840, height=2160, shader="phong"),
    "N3mixed": dict(name="N3: 40 synthetic mixed primitives (spheres, boxes, tori, rotated), 3840x2160, "
                         "sphere-tracing + BVH, Phong shader",
                    mixed=40, accel="BVH", width=3840, height=2160, shader="phong"),
    # SDF operators / Mandelbulb through the expression-program interpreter
    "N4chicken": dict(name="N4: Chicken (preset 17: nine nested smooth unions over ten boxes), 3840x2160, "
                           "sphere-tracing + BVH, Phong shader",
                      preset=17, accel="BVH", width=3840, height=2160, shader="phong"),
    "N4screw": dict(name="N4: Screw (preset 16: Round(Twist(Box)), Math.sin/cos per evaluation), 3840x2160, "
                         "sphere-tracing + BVH, Phong shader",
                    preset=16, accel="BVH", width=3840, height=2160, shader="phong"),
    "N4mandelbulb": dict(name="N4: Mandelbulb (preset 13: 80 escape iterations with atan2/asin/pow/sin/cos/log), "
                              "1920x1080, sphere-tracing + BVH, iteration-heatmap shader",
                         preset=13, accel="BVH", width=1920, height=1080, shader="iteration-heatmap"),
}


def cpu_baseline(wl, budget_s=20.0):
    """Oracle (C restatement of the reference's path) on this host's cores.  Bounded sample:
    every `stride`-th row of the same frame, rows spread over a thread pool (ctypes drops
    the GIL), scaled to frames/s; stride is chosen from a quick probe so the sample costs
    about `budget_s` of CPU work."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    W, H = wl["width"], wl["height"]
    spheres = O.synthetic_spheres(wl["synthetic"]) if "synthetic" in wl else None
    prims = O.synthetic_mixed_prims(wl["mixed"]) if "mixed" in wl else None
    sc = O.OracleScene(preset=wl.get("preset"), accel=wl["accel"], spheres=spheres, prims=prims)
    # 16 = the CPU share of a one-GPU box (the host itself reports every core of the node)
    cores = max(1, min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 16))
    t0 = time.time()
    probe_rows = list(range(H // 64, H, H // 8))[:8]
    for y in probe_rows:
        sc.render(W, H, y, y + 1)
    per_row = (time.time() - t0) / len(probe_rows)
    stride = max(1, int(per_row * H / budget_s + 0.999))
    rows = list(range(0, H, stride))

    def work(y):
        d, n, s, i = sc.render(W, H, y, y + 1)
        O.shade(wl["shader"], d, n, s, i, W, 1)
        return int(s.astype("int64").sum())

    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(work, rows))
    dt = time.time() - t0
    fps = 1.0 / (dt * H / len(rows))
    return {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "every %d-th row of the %dx%d frame (%d rows, %.1f s wall on %d threads), oracle/rm_oracle.c "
                      "render + shade, scaled by H/rows" % (stride, W, H, len(rows), dt, cores),
            "sample_avg_sdf_calls_per_pixel": total / (len(rows) * W)}


def js_engine_baseline(wl, rows=32):
    """The JS restatement (oracle/rm_oracle.js) on this host's node, one thread, on a band of `rows`
    rows in the middle of the frame: what the reference's own language runtime does per worker.
    Returns None when node is absent or the workload has no JS description (synthetic scenes)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("node") is None or "preset" not in wl:
        return None
    W, H = wl["width"], wl["height"]
    y0 = H // 2 - rows // 2
    cfg = dict(preset=wl["preset"], accel=wl["accel"], width=W, height=H, shader=wl["shader"], yStart=y0, yEnd=y0 + rows)
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "cfg.json"), "w") as f:
            json.dump(cfg, f)
        try:
            out = subprocess.check_output(["node", os.path.join(ROOT, "oracle", "rm_oracle.js"), "render",
                                           os.path.join(td, "cfg.json"), os.path.join(td, "out")], timeout=120)
            st = json.loads(out)
        except Exception:
            return None
    ms = st.get("render_ms")
    if not ms:
        return None
    return {"value": 1.0 / (ms * 1e-3 * H / rows), "unit": "frames/s", "threads": 1, "engine": st.get("engine"),
            "sample": "rows [%d,%d) of the %dx%d frame in %.0f ms, scaled by H/rows; the reference runs "
                      "min(4, cores-1) such workers (main.ts:318)" % (y0, y0 + rows, W, H, ms)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)   # eight rounds of the twelve frames in flight: ramp-up and drain are
    ap.add_argument("--warmup", type=int, default=16)  # ~3 % of the timed region (20 steps: 783 frames/s, 96 steps: 808)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--partition", default="interleaved", choices=["interleaved", "contiguous"])
    ap.add_argument("--stripe", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--analytics-sweep", action="store_true",
                    help="rotate the camera by 0.015 rad of yaw per frame like the reference's Analytics view "
                         "(main.ts:438-441) and report the per-frame metric series (main.ts:550-566); N = 1 only")
    ap.add_argument("--opt", action="append", default=[], help="kernel option key=value (rm_set_option)")
    ap.add_argument("--frames-in-flight", type=int, default=12,
                    help="frames enqueued concurrently, each on its own HIP stream with its own buffers: the tail of a "
                         "frame's persistent kernel (its slowest rays) overlaps the next frames; 1 = strictly serial")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON record: libraries that write to file descriptor 1 themselves (RCCL prints
    # a five-line version banner there when a process group starts) are sent to stderr until that line is printed
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import cpu_raymarcher_amd as R
    from cpu_raymarcher_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # Rehearsal mode (RM_BENCH_BACKEND=gloo): every rank uses cuda:0 and the gather is staged
    # through host memory, so the multi-rank code path can be run end to end on a one-GPU box.
    # The driver's runs use the default: one GPU per rank, RCCL ("nccl") over xGMI.
    backend = os.environ.get("RM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll = dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

            class _Done:
                def wait(self):
                    return True

            class _HostStagedGather:  # same call shape as torch.distributed.gather on CUDA tensors
                @staticmethod
                def gather(tensor, gather_list=None, dst=0, async_op=False):
                    torch.cuda.synchronize()
                    src = tensor.cpu()
                    out = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
                    dist.gather(src, out, dst=dst)
                    if rank == dst:
                        for g, o in zip(gather_list, out):
                            g.copy_(o)
                    return _Done()
            coll = _HostStagedGather

    wl = WORKLOADS[args.workload]
    W, H = wl["width"], wl["height"]
    ctx = R.Context(dev_index)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    # With frames in flight a launch runs best with few persistent workgroups (one per CU): the other frames' workgroups
    # fill the CUs, and each workgroup stages the scene tables once for four times as many tiles.  Measured on the final
    # kernel (scripts/bench_variants.sh, scripts/overlap_probe.py): whole frame 858 (one per CU), 841 (two), 825 (three)
    # frames/s; the 1/8-row shard of an 8-GPU job 0.22 (one) against 0.26 ms (two) per frame.  The static tile share
    # (option `static`) lost to this setting and is no longer used here.
    in_flight_bpc = None
    in_flight_opts = {}  # option -> (value with frames in flight, library default restored for the launch measured alone)
    if max(1, args.frames_in_flight) > 1:
        in_flight_opts["blocks_per_cu"] = (1, 4)  # final kernel: one per CU 858, two 841 frames/s; 1/8-row shard 0.22 against 0.26 ms
        if world == 1:  # whole frames: 256-pixel work items of 8 x 32 pixels (744 -> 763 frames/s; alone 2.00 -> 2.24 ms)
            in_flight_opts["item_px"] = (256, 128)
            in_flight_opts["tile_w"] = (8, 16)
        for k in list(in_flight_opts):
            if any(kv.startswith(k + "=") for kv in args.opt):
                del in_flight_opts[k]
        for k, (v, _) in in_flight_opts.items():
            ctx.set_option(k, v)
        in_flight_bpc = in_flight_opts.get("blocks_per_cu", (None, None))[0]
    scene = R.Scene(wl["accel"], ctx=ctx)
    if "synthetic" in wl:
        from cpu_raymarcher_amd.synthetic import synthetic_spheres  # SURVEY 8(d) C5 definition
        sp = synthetic_spheres(wl["synthetic"])
        scene.loadSpheres(sp[:, :3], sp[:, 3])
    elif "mixed" in wl:
        from cpu_raymarcher_amd.synthetic import mixed_prims_as_triples, synthetic_mixed_prims
        scene.loadPrims(mixed_prims_as_triples(synthetic_mixed_prims(wl["mixed"]), R.make_transform))
    else:
        scene.loadPreset(wl["preset"])
    tracer = R.SphereTracer()
    u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)  # noqa: E731
    acc = torch.zeros(4, dtype=torch.int64, device=dev)
    ev_pairs = []

    S = max(1, args.frames_in_flight)
    # (the analytics sweep keeps frames in flight too: every frame has its own camera, launch parameters and
    # accumulator, and the per-frame series is read in frame order after the timed region)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    if world == 1:
        sets = []
        for _ in range(S):
            sets.append(dict(depth=u8(W * H), normal=u8(3 * W * H), rgba=u8(4 * W * H),
                             sdf=torch.zeros(W * H, dtype=torch.int16, device=dev),
                             iters=torch.zeros(W * H, dtype=torch.int16, device=dev),
                             acc=torch.zeros(4, dtype=torch.int64, device=dev)))
        acc = sets[0]["acc"]
        frame_no = [0]

        series = []  # analytics sweep: one accumulator per frame, read after the timed region

        def step(timed):
            k = frame_no[0] % S
            frame_no[0] += 1
            b = sets[k]
            if args.analytics_sweep:
                scene.camera.rotateCamera(0, 0.015)  # main.ts:438-441
            with torch.cuda.stream(streams[k]):
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                tracer.runRaymarcher(scene, b["depth"], b["normal"], b["sdf"], b["iters"], W, H, 0.0,
                                     shadedBuffer=b["rgba"], shader=wl["shader"])
                if timed:
                    e1.record()
                    ev_pairs.append((e0, e1))
                if args.analytics_sweep and timed:
                    a = torch.zeros(4, dtype=torch.int64, device=dev)
                    ctx.reduce_counters_enqueue(b["sdf"], b["iters"], a)
                    series.append((scene.camera.yaw, a))
                ctx.reduce_counters_enqueue(b["sdf"], b["iters"], b["acc"])

        def finish():
            pass
    else:
        # Only RGBA travels (north_star: "RCCL gather ... of the per-tile RGBA buffers").  The per-pixel counters stay on
        # the rank that produced them: each rank reduces its own rows (main.ts:528-548 is a sum / max / min, so partial
        # results combine exactly) into a 32-byte accumulator in the tail of its packed buffer, which rides along in the
        # same gather; rank 0 combines the N partial accumulators per frame.
        gather_counters = args.partition != "interleaved"  # the per-range fallback path keeps the three-section gather
        sections = ("rgba", "sdf", "iters") if gather_counters else ("rgba",)
        layout = D.FrameLayout(W, H, world, sections, args.partition, args.stripe, tail=0 if gather_counters else 32)
        render_rows = D.gpu_render_rows(ctx, scene, W, H, wl["shader"], layout)
        timed_flag = [False]
        my_px = W * sum(b - a for a, b in layout.rows(rank))
        local_counters = {}  # packed buffer -> this rank's sdfEval / iters for that buffer set

        def extra(packed):
            key = packed.data_ptr()
            if key not in local_counters:
                local_counters[key] = {"sdf": torch.zeros(layout.cap * W, dtype=torch.int16, device=dev),
                                       "iters": torch.zeros(layout.cap * W, dtype=torch.int16, device=dev)}
            return local_counters[key]

        def timed_render_rows(a, b, local, packed):
            if timed_flag[0]:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                render_rows(a, b, local, packed)
                e1.record()
                ev_pairs.append((e0, e1))
            else:
                render_rows(a, b, local, packed)

        render_all = D.gpu_render_all(ctx, scene, W, H, wl["shader"], layout, rank, extra=None if gather_counters else extra)

        def timed_render_all(packed):
            if timed_flag[0]:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                render_all(packed)
                e1.record()
                ev_pairs.append((e0, e1))
            else:
                render_all(packed)
            if not gather_counters:  # this rank's partial diagnostics, into the tail that travels with the gather
                c = extra(packed)
                tail = packed[layout.tail_offset:layout.tail_offset + 32].view(torch.int64)
                ctx.reduce_counters_enqueue(c["sdf"][:my_px], c["iters"][:my_px], tail)

        shr = D.ShardedFrameRenderer(layout, rank, world, timed_render_rows, u8, coll,
                                     render_all=timed_render_all if render_all else None,
                                     frames_in_flight=max(2, S), streams=streams if S > 1 else None)
        # rank 0 reassembly: one indexed row-gather per section (D.GpuFrameAssembler)
        asm = None
        if rank == 0:
            asm = D.GpuFrameAssembler(layout, dev, shr.nbuf)
            shr.recv = asm.gather_lists()

        # one accumulator per buffer set: reductions of different frames run concurrently on different streams
        accs = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(shr.nbuf)]
        acc = accs[0]
        lo32 = torch.tensor(0xFFFFFFFF, dtype=torch.int64, device=dev)

        def assemble(slot):
            with shr.on_stream(slot):
                frame = asm.assemble(slot)
                if gather_counters:
                    ctx.reduce_counters_enqueue(frame["sdf"].view(torch.int16), frame["iters"].view(torch.int16), accs[slot])
                else:  # combine the ranks' partial accumulators: sums, max of the low word, min of the high word
                    part = asm.recv2d[slot][:, layout.tail_offset:layout.tail_offset + 32].view(torch.int64)
                    torch.sum(part[:, :2], dim=0, out=accs[slot][:2])
                    accs[slot][2] = torch.amax(part[:, 2] & lo32) | (torch.amin(part[:, 2] >> 32) << 32)

        pending = []

        def step(timed):
            timed_flag[0] = timed
            slot = shr.submit()
            pending.append(slot)
            if len(pending) >= shr.nbuf:  # the oldest frame in flight is assembled while the newer ones render / gather
                s0 = pending.pop(0)
                shr.finish(s0)
                if rank == 0:
                    assemble(s0)

        def finish():
            while pending:
                s0 = pending.pop(0)
                shr.finish(s0)
                if rank == 0:
                    assemble(s0)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, not a step: every stream's first launch creates its hardware queue and touches its buffer set (tens of
    # milliseconds each).  With fewer warm-up steps than streams that cost would land in the timed region, so each
    # stream is used once here; the W warm-up steps follow.
    if S > args.warmup:
        for _ in range(S):
            step(False)
        finish()
        sync()
    for _ in range(args.warmup):
        step(False)
    finish()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    finish()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: the render kernel.  Average launch duration from the HIP events recorded
    # on the launch stream; per launch this rank rendered rows_launched / launches pixels-rows.
    kern_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / max(1, len(ev_pairs))
    n_launches = len(ev_pairs)
    if world == 1:
        px_per_launch = W * H
    else:
        rows_mine = layout.rows(rank)
        launches = 1 if args.partition == "interleaved" else max(1, len(rows_mine))
        px_per_launch = W * sum(b - a for a, b in rows_mine) / launches
    # With several frames in flight the launches overlap: a launch's own duration (kern_ms) then spans the other
    # frames' work as well, and bytes / duration would under-count by the overlap factor.  The device-level figure
    # is bytes per launch x launches / wall time of the timed region; the launch duration of the kernel running
    # alone is measured right after the timed region (serial launches, HIP events on the launch stream).
    kern_serial_ms = kern_ms
    if S > 1 and world == 1:
        ser = []
        b = sets[0]
        for k, (_, dflt) in in_flight_opts.items():
            ctx.set_option(k, dflt)  # the launch running alone uses the library defaults
        with torch.cuda.stream(streams[0]):
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                tracer.runRaymarcher(scene, b["depth"], b["normal"], b["sdf"], b["iters"], W, H, 0.0,
                                     shadedBuffer=b["rgba"], shader=wl["shader"])
                e1.record()
                ser.append((e0, e1))
                torch.cuda.synchronize()
        kern_serial_ms = sum(a.elapsed_time(c) for a, c in ser) / len(ser)
        for k, (v, _) in in_flight_opts.items():
            ctx.set_option(k, v)
    bytes_per_launch = ALG_BYTES_PER_PIXEL * px_per_launch
    if S > 1:
        achieved = bytes_per_launch * n_launches / elapsed / 1e9 if elapsed > 0 else 0.0
    else:
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    achieved_serial = bytes_per_launch / (kern_serial_ms * 1e-3) / 1e9 if kern_serial_ms > 0 else 0.0

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so
    # the per-launch figure measured with rocprofv3 for this workload is taken from the committed
    # profile (profiles/r01/traffic.json, which names its source), at N = 1 only.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "traffic.json")) as f:
            tj = json.load(f)
        if world == 1 and args.workload in tj and not args.opt:
            traffic = tj[args.workload]["bytes"]
    except (OSError, ValueError, KeyError):
        traffic = None

    if rank == 0:
        d = ctx.decode_acc(acc)
        fps = args.steps / elapsed
        out = {
            "metric": "frames/sec, 3840x2160 Dense-Grid sphere-trace (+ avg SDF-calls/pixel)" if args.workload == "C3"
                      else "frames/sec, " + wl["name"],
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["name"], "width": W, "height": H, "acceleration_structure": wl["accel"],
                       "shader": wl["shader"], "camera": {"pitch": 0.0, "yaw": 0.0}, "frames_in_flight": S,
                       "parallelism": "1 GPU" if world == 1 else
                       "row-tile shard x%d (%s, stripe %d) + RCCL gather of RGBA and per-rank diagnostics sums to rank 0"
                       % (world, args.partition, args.stripe)},
            "avg_sdf_calls_per_pixel": d["total_sdf"] / (W * H), "avg_iterations_per_pixel": d["total_iters"] / (W * H),
            "max_sdf_calls": d["max_sdf"], "min_sdf_calls": d["min_sdf"],
            "sphere_evals_per_s": d["total_sdf"] * fps,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/r01/traffic.json"
                         if traffic else None,
                         "kernel": "render_kernel_v2<2, true, true, false>" if args.workload == "C3" else "render kernel",
                         "kernel_ms": kern_serial_ms, "kernel_ms_in_flight": kern_ms, "frames_in_flight": S,
                         "persistent_workgroups_per_cu": in_flight_bpc or 4,
                         "in_flight_options": {k: v for k, (v, _) in in_flight_opts.items()},
                         "achieved_one_launch_alone": achieved_serial,
                         "basis": ("device level: algorithmic bytes per launch x %d launches / wall time of the timed "
                                   "region (%d frames in flight overlap; kernel_ms is the launch running alone, "
                                   "kernel_ms_in_flight the mean overlapped launch)" % (n_launches, S)) if S > 1 else
                                  "algorithmic bytes per launch / mean launch duration (HIP events, timed region)",
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "FP64-VALU/divergence bound, not HBM bound: 12 B/pixel out, ~1e3 FP64 ops/pixel "
                                 "(DESIGN.md)"},
        }
        if world > 1:
            # outside the timed region: the gathered, reassembled frame must equal this frame
            # rendered whole on rank 0's GPU, byte for byte
            full = {"rgba": u8(4 * W * H), "sdf": u8(2 * W * H), "iters": u8(2 * W * H)}
            tracer.runRaymarcher(scene, None, None, full["sdf"].view(torch.int16), full["iters"].view(torch.int16),
                                 W, H, 0.0, shadedBuffer=full["rgba"], shader=wl["shader"])
            whole = torch.zeros(4, dtype=torch.int64, device=dev)
            ctx.reduce_counters_enqueue(full["sdf"].view(torch.int16), full["iters"].view(torch.int16), whole)
            torch.cuda.synchronize()
            out["gathered_frame_equals_single_gpu_frame"] = all(bool(torch.equal(asm.frame[s], full[s])) for s in asm.frame)
            out["combined_diagnostics_equal_single_gpu_diagnostics"] = ctx.decode_acc(whole) == d
        if world == 1 and args.analytics_sweep:
            out["config"]["camera"] = {"pitch": 0.0, "yaw": "+0.015 rad per frame (analytics sweep)"}
            out["analytics_series"] = [
                {"yaw": round(y, 6), "avg_sdf_calls": ctx.decode_acc(a)["total_sdf"] / (W * H),
                 "avg_iterations": ctx.decode_acc(a)["total_iters"] / (W * H), "max_sdf_calls": ctx.decode_acc(a)["max_sdf"],
                 ("frame_ms" if S == 1 else "frame_ms_overlapped"): e0.elapsed_time(e1)} for (y, a), (e0, e1) in zip(series, ev_pairs)]
        if world == 1 and not args.no_cpu_baseline and not args.analytics_sweep:
            out["cpu_baseline"] = cpu_baseline(wl)
            js = js_engine_baseline(wl)
            if js:
                out["cpu_baseline"]["js_engine_single_thread"] = js
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
