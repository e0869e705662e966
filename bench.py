#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X sphere-tracing render path.

    python bench.py --gpus N --steps K --warmup W
(N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`,
one rank per GPU, RCCL through torch.distributed.)

A step = one frame of BASELINE.json's metric workload (C3: Dense Sphere Grid, 125 spheres,
3840x2160, sphere tracing + BVH, iteration-heatmap shader): render + fused shade into
{depth, normal, sdfEval, iters, RGBA} resident in HBM, and the diagnostics of main.ts:528-548 (sum / max / min of the
counters), which the render kernel accumulates while it stores the counters (rm_render_attach_diagnostics; `--diagnostics
reduce` runs round 2's two reduction launches over the stored counters instead).  At N > 1 the frame's rows are sharded over the ranks
(interleaved 16-row stripes, dealt by a weighted round-robin: rank 0 also reassembles the frame, so it renders a
smaller share); every rank reduces the counters of its own rows, and one gather brings its RGBA rows and its
32-byte partial diagnostics to rank 0, which rebuilds the frame and combines the partial sums with ONE native
kernel -- total work fixed, so scaling is "strong".

Frames are independent, so `--frames-in-flight S` (default 12) enqueues consecutive frames on S HIP streams with
S buffer sets: the tail of a frame's persistent kernel -- its slowest rays, ~0.3 ms during which most CUs idle --
overlaps the following frames.  Every frame is still rendered, shaded and reduced in full; S = 1 is strictly serial.

Rank 0 prints ONE JSON line (contract in the task statement).  What each number is:
  value / ms_per_step   K frames with S frames in flight / wall time of the timed region (max over ranks)
  value_serial          the same frames strictly one after the other (S = 1), timed right after the timed region
  setup_frames          untimed frames run BEFORE the W warm-up frames (one per stream: a stream's first launch creates
                        its hardware queue); 0 when W >= S
  roofline              the render kernel ALONE: achieved = algorithmic bytes per launch / kernel_ms, where kernel_ms
                        is the mean duration of serial launches (HIP events on the launch stream); frac = achieved /
                        peak.  The overlapped figure (bytes x launches / wall time) is `achieved_in_flight`.
                        traffic = FETCH_SIZE + WRITE_SIZE per launch and valu = the instruction mix priced with the
                        measured issue costs, both from profiles/r03/pmc_<workload>.json -- used only if that file was
                        measured on the kernel sources now in the tree (source hash) with the options of this run.
  frames_verified       after the timed region every buffer set that was in flight (the last frame rendered into each of
                        the S sets, with the in-flight options and the tail-ramp values of the timed burst) is hashed
                        (SHA-256, all five buffers) against tests/golden/golden.json; the count of sets that match.  A
                        mismatch makes the run fail.
  host_enqueue_ms       host time this rank spent enqueueing one frame (render, gather, fan-in calls), mean over the
                        timed region; at N > 1 one value per rank
  cpu_baseline          the reference's policy on this box's host cores: oracle/rm_oracle.js (the JS restatement; the
                        reference itself cannot run here) on node worker_threads, N = max(1, min(4, cores - 1)) workers,
                        contiguous ceil(H/N)-row tiles, whole frames after a warm-up (main.ts:318,444-449), plus the C
                        port on all cores.  The only leg that loads oracle/.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

# Before anything initialises HIP / HSA (set after torch.cuda.set_device these have no effect):
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")           # a hardware queue per stream in flight (default 4)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this pool

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_PIXEL = 12  # depth 1 + normal 3 + sdfEval 2 + iters 2 + RGBA 4, ~0 read
PROFILE_DIR = os.path.join(ROOT, "profiles", "r03")

WORKLOADS = {
    "C2": dict(name="C2: Grid of Spheres (9), 1920x1080, sphere-tracing + BVH, Phong shader",
               preset=2, accel="BVH", width=1920, height=1080, shader="phong", golden="C2_grid_1080p_bvh_phong"),
    "C3": dict(name="C3: Dense Sphere Grid (125 spheres), 3840x2160, sphere-tracing + BVH, iteration-heatmap shader",
               preset=3, accel="BVH", width=3840, height=2160, shader="iteration-heatmap", golden="C3_dense_4k_bvh_iterheat"),
    "C5": dict(name="C5: synthetic 10000 random spheres (splitmix64 0x5EED5EED), 3840x2160, sphere-tracing + Octree, "
                    "iteration-heatmap shader",
               synthetic=10000, accel="Octree", width=3840, height=2160, shader="iteration-heatmap",
               golden="C5_random10k_4k_octree_iterheat"),
    # boxes / tori / rotated transforms (not BASELINE configs; same frame size as C3)
    "N3": dict(name="N3: Pyramid of Boxes (preset 9), 3840x2160, sphere-tracing + BVH, Phong shader",
               preset=9, accel="BVH", width=3840, height=2160, shader="phong"),
    "N3mixed": dict(name="N3: 40 synthetic mixed primitives (spheres, boxes, tori, rotated), 3840x2160, "
                         "sphere-tracing + BVH, Phong shader",
                    mixed=40, accel="BVH", width=3840, height=2160, shader="phong"),
    # SDF operators / Mandelbulb: the scene's trees compiled into the kernel at run time (rm_rtc.h; --opt specialise=0: the interpreter)
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


def host_cores():
    """Cores this process may use; 16 = the CPU share of a one-GPU box (the host itself reports every core of the node)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(16, os.cpu_count() or 1, aff))


def kernel_source_hash():
    """sha-256 (16 hex digits) of the kernel sources: a PMC file measured on other sources is stale."""
    csrc = os.path.join(ROOT, "cpu_raymarcher_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc(workload, options):
    """profiles/r03/pmc_<workload>.json (scripts/profile_r02.sh + scripts/make_pmc_json.py), or (None, reason)."""
    path = os.path.join(PROFILE_DIR, "pmc_%s.json" % workload)
    try:
        with open(path) as f:
            pj = json.load(f)
    except (OSError, ValueError):
        return None, "no PMC file for this workload"
    if pj.get("source_sha16") != kernel_source_hash():
        return None, "stale: %s was measured on kernel sources %s, the tree holds %s" % (
            os.path.relpath(path, ROOT), pj.get("source_sha16"), kernel_source_hash())
    if {k: int(v) for k, v in pj.get("options", {}).items()} != {k: int(v) for k, v in options.items()}:
        return None, "stale: %s was measured with options %s, this run uses %s" % (
            os.path.relpath(path, ROOT), pj.get("options"), options)
    return pj, os.path.relpath(path, ROOT)


def cpu_baseline_port(wl, frames=5):
    """The C port (oracle/rm_oracle.c) on all host cores: whole frames, one warm-up frame, rows handed out one at a
    time to a thread pool (ctypes drops the GIL) -- dynamic row scheduling, i.e. better balanced than the reference's
    contiguous tiles, so this is the stronger CPU number.  Frames are bounded to ~20 s of wall time."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    W, H = wl["width"], wl["height"]
    spheres = O.synthetic_spheres(wl["synthetic"]) if "synthetic" in wl else None
    prims = O.synthetic_mixed_prims(wl["mixed"]) if "mixed" in wl else None
    sc = O.OracleScene(preset=wl.get("preset"), accel=wl["accel"], spheres=spheres, prims=prims)
    cores = host_cores()
    # probe: is a whole frame affordable?  Otherwise every stride-th row, spread over the whole frame, and say so.
    t0 = time.time()
    probe_rows = list(range(H // 64, H, H // 8))[:8]
    for y in probe_rows:
        sc.render(W, H, y, y + 1)
    per_row = (time.time() - t0) / len(probe_rows)
    stride = max(1, int(per_row * H * (frames + 1) / cores / 20.0 + 0.999))
    rows = list(range(0, H, stride))

    def work(y):
        d, n, s, i = sc.render(W, H, y, y + 1)
        O.shade(wl["shader"], d, n, s, i, W, 1)
        return int(s.astype("int64").sum())

    times, total = [], 0
    with ThreadPoolExecutor(cores) as ex:
        sum(ex.map(work, rows))  # warm-up frame
        for _ in range(frames):
            t0 = time.time()
            total = sum(ex.map(work, rows))
            times.append(time.time() - t0)
    dt = sum(times) / len(times)
    fps = 1.0 / (dt * H / len(rows))
    what = ("%d whole %dx%d frames" % (frames, W, H)) if stride == 1 else \
        ("%d passes over every %d-th row of the %dx%d frame (%d rows, spread over the whole frame), scaled by H/rows"
         % (frames, stride, W, H, len(rows)))
    return {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%s after one warm-up pass, %.2f s each on %d threads, rows scheduled dynamically; oracle/rm_oracle.c "
                      "render + shade" % (what, dt, cores),
            "sample_avg_sdf_calls_per_pixel": total / (len(rows) * W)}


def cpu_baseline_js_pool(wl, frames=5, budget_s=25.0):
    """The reference's own policy (main.ts:318,444-449) with the JS restatement on this host's node: worker_threads,
    N = max(1, min(4, cores - 1)) workers, contiguous ceil(H/N)-row tiles, scene rebuilt per job, tiles transferred
    back, shade + diagnostics on the main thread; `frames` frames timed after one warm-up.  Bounded: when whole frames
    would exceed ~budget_s, every k-th row of every tile is rendered (spread over the whole frame) and the time scaled
    by k -- stated in `sample`.  None when node is absent or the workload has no JS description (synthetic scenes)."""
    import shutil
    import tempfile
    if shutil.which("node") is None or "preset" not in wl:
        return None
    W, H = wl["width"], wl["height"]
    cores = host_cores()
    n_workers = max(1, min(4, cores - 1))
    js = os.path.join(ROOT, "oracle", "rm_oracle.js")

    def run(cfg, timeout):
        with tempfile.TemporaryDirectory() as td:
            with open(os.path.join(td, "cfg.json"), "w") as f:
                json.dump(cfg, f)
            return json.loads(subprocess.check_output(["node", js, "pool", os.path.join(td, "cfg.json")], timeout=timeout))

    base = dict(preset=wl["preset"], accel=wl["accel"], width=W, height=H, shader=wl["shader"], cores=cores, workers=n_workers)
    try:
        probe = run(dict(base, frames=1, rowStride=64), 300)  # ~1.5 % of the rows: how long would a frame take?
        frame_s = probe["frame_ms"][0] * 1e-3 * 64
        k = max(1, int(frame_s * (frames + 1) / budget_s + 0.999))
        res = run(dict(base, frames=frames, rowStride=k), 600)
    except Exception as e:  # noqa: BLE001 -- a baseline leg must not take the bench line down
        return {"error": "%s: %s" % (type(e).__name__, e)}
    ms = sum(res["frame_ms"]) / len(res["frame_ms"])
    scale = H / res["sampled_rows"]
    return {"value": 1.0 / (ms * 1e-3 * scale), "unit": "frames/s", "cores": cores, "workers": n_workers,
            "kind": "port (JS restatement under the reference's worker policy)", "engine": res.get("engine"),
            "frame_ms_sampled": res["frame_ms"],
            "sample": ("%d whole frames" % frames if k == 1 else
                       "%d frames of every %d-th row of every worker's contiguous tile (%d of %d rows, spread over the "
                       "whole frame), time scaled by H/rows" % (frames, k, res["sampled_rows"], H)) +
                      " after one warm-up frame; N = max(1, min(4, cores - 1)) = %d worker_threads, ceil(H/N)-row tiles, "
                      "scene rebuilt per job, shade + diagnostics on the main thread (main.ts:318,444-449,493-548)" % n_workers,
            "sample_avg_sdf_calls_per_pixel": res["diagnostics"]["sum_sdf"] / (res["sampled_rows"] * W)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)   # eight rounds of the twelve frames in flight: ramp-up and drain are
    ap.add_argument("--warmup", type=int, default=16)  # ~3 % of the timed region
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--partition", default="interleaved", choices=["interleaved", "contiguous"])
    ap.add_argument("--stripe", type=int, default=16)
    ap.add_argument("--root-share", default="auto",
                    help="N > 1: rank 0's stripe share relative to the other ranks' (1.0 = equal deal), or 'auto': measured "
                         "before the warm-up (rank 0's reassembly + an equal shard's render time, no collectives)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--diagnostics", default="fused", choices=["fused", "reduce"],
                    help="fused: the render kernel accumulates the diagnostics of main.ts:528-548 itself (rm_render_attach_diagnostics); "
                         "reduce: two more launches per frame re-read the stored counters (round 2)")
    ap.add_argument("--no-verify", action="store_true", help="skip the SHA-256 check of the in-flight buffer sets after the timed region")
    ap.add_argument("--analytics-sweep", action="store_true",
                    help="rotate the camera by 0.015 rad of yaw per frame like the reference's Analytics view "
                         "(main.ts:438-441) and report the per-frame metric series (main.ts:550-566); N = 1 only")
    ap.add_argument("--opt", action="append", default=[], help="kernel option key=value (rm_set_option)")
    ap.add_argument("--tail-ramp", type=int, default=4,
                    help="N > 0 (one GPU, frames in flight): the last N frames of the timed burst launch with N - (frames after them) persistent "
                         "workgroups per CU instead of 1 (max 4): they find fewer and fewer other frames beside them.  0: off")
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
    phase = ["setup", 0]  # what this rank was doing, for the message of a failing collective (fail() below)

    def fail(exc):
        """A collective that fails or times out must end the run with a message and a non-zero exit code inside the
        driver's limit, not sit in a wait: rank, phase and frame to stderr, then leave without the interpreter's
        teardown (which would wait for the process group again).  Never re-exec: this process has touched the GPU."""
        sys.stderr.write("bench.py: rank %d of %d failed in phase '%s', frame %d: %s: %s\n"
                         % (rank, world, phase[0], phase[1], type(exc).__name__, exc))
        sys.stderr.flush()
        os._exit(3)

    if world > 1:
        import datetime
        # the watchdog of torch's NCCL / RCCL backend aborts the process when a collective exceeds the timeout
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        to = datetime.timedelta(seconds=int(os.environ.get("RM_BENCH_COLLECTIVE_TIMEOUT_S", "120")))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=to)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=to)
        except Exception as e:  # noqa: BLE001
            fail(e)
    if world > 1 and backend != "nccl":

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
    user_opts = {}
    for kv in args.opt:
        k, v = kv.split("=")
        user_opts[k] = int(v)
        ctx.set_option(k, int(v))
    # With frames in flight a launch runs best with few persistent workgroups (one per CU): the other frames' workgroups
    # fill the CUs, and each workgroup stages the scene tables once for four times as many tiles.
    S = max(1, args.frames_in_flight)
    in_flight_opts = {}  # option -> (value with frames in flight, library default used by serial launches)
    if S > 1:
        in_flight_opts["blocks_per_cu"] = (1, ctx.get_option("blocks_per_cu"))
        # longest-first item order (lpt) shortens the tail of a frame that runs ALONE; overlapping frames hide that tail
        # anyway and the extra sort kernel per launch costs ~1 %
        in_flight_opts["lpt"] = (0, ctx.get_option("lpt"))
        if world == 1:  # whole frames: 256-pixel work items of 8 x 32 pixels
            in_flight_opts["item_px"] = (256, ctx.get_option("item_px"))
            in_flight_opts["tile_w"] = (8, ctx.get_option("tile_w"))
        for k in list(in_flight_opts):
            if k in user_opts:
                del in_flight_opts[k]

    def apply_opts(in_flight):
        for k, (v, dflt) in in_flight_opts.items():
            ctx.set_option(k, v if in_flight else dflt)

    apply_opts(True)
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
    fused = args.diagnostics == "fused"
    tracer = R.SphereTracer()
    u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)  # noqa: E731
    i16 = lambda n: torch.zeros(n, dtype=torch.int16, device=dev)  # noqa: E731
    acc = torch.zeros(4, dtype=torch.int64, device=dev)
    ev_pairs = []
    root_balance = None

    # (the analytics sweep keeps frames in flight too: every frame has its own camera, launch parameters and
    # accumulator, and the per-frame series is read in frame order after the timed region)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    if world == 1:
        sets = []
        for _ in range(S):
            sets.append(dict(depth=u8(W * H), normal=u8(3 * W * H), rgba=u8(4 * W * H), sdf=i16(W * H), iters=i16(W * H),
                             acc=torch.zeros(4, dtype=torch.int64, device=dev)))
        acc = sets[0]["acc"]
        last_acc = [acc]
        frame_no = [0]
        series = []  # analytics sweep: one accumulator per frame, read after the timed region

        def step(timed, remaining=None):
            k = frame_no[0] % S
            frame_no[0] += 1
            b = sets[k]
            if remaining is not None and args.tail_ramp and "blocks_per_cu" in in_flight_opts:
                # the last frames of the burst find fewer and fewer other frames beside them: each brings more persistent
                # workgroups of its own (up to six: what a CU holds)
                ctx.set_option("blocks_per_cu", max(1, min(6, args.tail_ramp - remaining)) if remaining < args.tail_ramp else 1)
            if args.analytics_sweep:
                scene.camera.rotateCamera(0, 0.015)  # main.ts:438-441
            with torch.cuda.stream(streams[k]):
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                a = b["acc"]
                if args.analytics_sweep and timed:  # the per-frame metric series of main.ts:550-566: one accumulator per frame
                    a = torch.zeros(4, dtype=torch.int64, device=dev)
                    series.append((scene.camera.yaw, a))
                tracer.runRaymarcher(scene, b["depth"], b["normal"], b["sdf"], b["iters"], W, H, 0.0,
                                     shadedBuffer=b["rgba"], shader=wl["shader"], diagnostics=a if fused else None)
                if timed:
                    e1.record()
                    ev_pairs.append((e0, e1))
                if not fused:
                    ctx.reduce_counters_enqueue(b["sdf"], b["iters"], a)
                last_acc[0] = a

        def finish():
            pass
    else:
        # Only RGBA travels (north_star: "RCCL gather ... of the per-tile RGBA buffers").  The per-pixel counters stay on
        # the rank that produced them: each rank reduces its own rows (main.ts:528-548 is a sum / max / min, so partial
        # results combine exactly) into a 32-byte accumulator in the tail of its packed buffer, which rides along in the
        # same gather; rank 0 combines the N partial accumulators inside the kernel that rebuilds the frame.
        gather_counters = args.partition != "interleaved"  # the reference's contiguous partition keeps the three-section gather
        sections = ("rgba", "sdf", "iters") if gather_counters else ("rgba",)
        nbuf = max(2, S)

        def make_layout(weights):
            return D.FrameLayout(W, H, world, sections, args.partition, args.stripe, tail=0 if gather_counters else 32,
                                 weights=weights)

        # ---- rank 0's share.  Rank 0 renders its stripes AND rebuilds the frame (and runs the receiving side of the
        # gather); with an equal deal it is the slowest rank and sets the frame rate.  'auto' measures, with
        # no collective involved, (a) an equal shard's render + reduce and (b) the same plus rank 0's reassembly, both
        # with S frames in flight, and deals rank 0 the share that equalises the ranks (distributed.balanced_weights).
        weights = None
        if args.partition == "interleaved":
            if args.root_share != "auto":
                weights = [max(1, int(round(1000 * float(args.root_share))))] + [1000] * (world - 1)
                root_balance = {"mode": "given", "weights": weights}
            else:
                lay0 = make_layout(None)
                cal_sets = [dict(p=u8(lay0.nbytes), sdf=i16(lay0.cap * W), iters=i16(lay0.cap * W)) for _ in range(nbuf)]
                cal_px = W * sum(b - a for a, b in lay0.rows(rank))
                cal_asm = D.GpuFrameAssembler(lay0, dev, nbuf, ctx=ctx) if rank == 0 else None
                cal_acc = torch.zeros(4, dtype=torch.int64, device=dev)

                ra_extra = [None]  # the buffer set of the frame being enqueued: its rank-local sdfEval / iters
                ra = D.gpu_render_all(ctx, scene, W, H, wl["shader"], lay0, rank, extra=lambda packed: ra_extra[0], diag_in_tail=fused)

                def cal_frames(n, with_asm):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for f in range(n):
                        k = f % nbuf
                        b = cal_sets[k]
                        with torch.cuda.stream(streams[k % S]):
                            ra_extra[0] = b
                            ra(b["p"])
                            if not fused:
                                tail = b["p"][lay0.tail_offset:lay0.tail_offset + 32].view(torch.int64)
                                ctx.reduce_counters_enqueue(b["sdf"][:cal_px], b["iters"][:cal_px], tail)
                            if with_asm:
                                cal_asm.assemble(k, cal_acc)
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / n

                cal_frames(2 * nbuf, rank == 0)
                t_shard = cal_frames(4 * nbuf, False)
                t_root = cal_frames(4 * nbuf, True) if rank == 0 else 0.0
                t = torch.tensor([t_shard, t_root], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                tl = [torch.zeros_like(t) for _ in range(world)]
                dist.all_gather(tl, t)
                shard = sum(float(x[0]) for x in tl) / world
                over = max(0.0, float(tl[0][1]) - float(tl[0][0]))
                weights = D.balanced_weights(world, over, shard)
                root_balance = {"mode": "auto", "weights": weights, "equal_shard_ms": 1e3 * shard,
                                "root_reassembly_ms": 1e3 * over,
                                "note": "measured before the warm-up without collectives, %d frames in flight" % S}
                del cal_sets, cal_asm
        layout = make_layout(weights)
        render_rows = D.gpu_render_rows(ctx, scene, W, H, wl["shader"], layout)
        timed_flag = [False]
        my_px = W * sum(b - a for a, b in layout.rows(rank))
        local_counters = {}  # packed buffer -> this rank's sdfEval / iters for that buffer set

        def extra(packed):
            key = packed.data_ptr()
            if key not in local_counters:
                local_counters[key] = {"sdf": i16(layout.cap * W), "iters": i16(layout.cap * W)}
            return local_counters[key]

        def timed_call(fn, *a):
            if timed_flag[0]:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(*a)
                e1.record()
                ev_pairs.append((e0, e1))
            else:
                fn(*a)

        def timed_render_rows(a, b, local, packed):
            timed_call(render_rows, a, b, local, packed)

        render_all = D.gpu_render_all(ctx, scene, W, H, wl["shader"], layout, rank, extra=None if gather_counters else extra,
                                      diag_in_tail=fused and not gather_counters)

        def timed_render_all(packed):
            timed_call(render_all, packed)
            if not gather_counters and not fused:  # this rank's partial diagnostics, into the tail that travels with the gather
                c = extra(packed)
                tail = packed[layout.tail_offset:layout.tail_offset + 32].view(torch.int64)
                ctx.reduce_counters_enqueue(c["sdf"][:my_px], c["iters"][:my_px], tail)

        shr = D.ShardedFrameRenderer(layout, rank, world, timed_render_rows, u8, coll,
                                     render_all=timed_render_all if render_all else None,
                                     frames_in_flight=nbuf, streams=streams if S > 1 else None)
        asm = None
        if rank == 0:
            asm = D.GpuFrameAssembler(layout, dev, shr.nbuf, ctx=ctx)
            shr.recv = asm.gather_lists()
        # one accumulator per buffer set: frames in flight run concurrently on different streams
        accs = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(shr.nbuf)]
        acc = accs[0]

        def assemble(slot):
            with shr.on_stream(slot):
                if gather_counters:
                    frame = asm.assemble(slot)
                    ctx.reduce_counters_enqueue(frame["sdf"].view(torch.int16), frame["iters"].view(torch.int16), accs[slot])
                else:  # one kernel: stripes -> frame, partial accumulators -> accs[slot]
                    asm.assemble(slot, accs[slot])

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
    # stream is used once here (reported as setup_frames); the W warm-up steps follow.
    setup_frames = 0
    if S > args.warmup:
        setup_frames = S
        phase[0] = "set-up frames"
        try:
            for i in range(S):
                phase[1] = i
                step(False)
            finish()
            sync()
        except Exception as e:  # noqa: BLE001
            if world > 1:
                fail(e)
            raise
    host_enqueue = 0.0
    try:
        phase[0] = "warm-up"
        for i in range(args.warmup):
            phase[1] = i
            step(False)
        finish()
        sync()
        phase[0] = "timed region"
        t0 = time.perf_counter()
        for i in range(args.steps):
            phase[1] = i
            h0 = time.perf_counter()
            if world == 1:
                step(True, args.steps - 1 - i)
            else:
                step(True)
            host_enqueue += time.perf_counter() - h0
        phase[0] = "drain of the timed region"
        finish()
        sync()
        elapsed = time.perf_counter() - t0
    except Exception as e:  # noqa: BLE001 -- a failed collective / launch: say where, leave non-zero (fail() does not return)
        if world > 1:
            fail(e)
        raise
    apply_opts(True)
    host_enqueue_ms = [1e3 * host_enqueue / max(1, args.steps)]
    if world > 1:
        phase[0] = "all-reduce of the timings"
        try:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            he = torch.tensor(host_enqueue_ms, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            hl = [torch.zeros_like(he) for _ in range(world)]
            dist.all_gather(hl, he)
            host_enqueue_ms = [float(x.item()) for x in hl]
        except Exception as e:  # noqa: BLE001
            fail(e)
    kernel_in_flight = ctx.last_kernel()

    # The timed configuration checks itself: every buffer set that was in flight holds the last frame
    # rendered into it -- with the in-flight options, and for the last sets the tail-ramp values -- and all five buffers of
    # every set must hash to the committed golden fixtures of this workload (C = JS oracle agreement, tests/golden/).
    frames_verified, verify_note = None, None
    if world == 1 and not args.analytics_sweep and not args.no_verify and "golden" in wl:
        with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
            gold = json.load(f)[wl["golden"]]["sha256"]
        used = min(S, frame_no[0])
        frames_verified = 0
        for k in range(used):
            b = sets[k]
            bad = [name for name in ("depth", "normal", "sdf", "iters", "rgba")
                   if not (name == "rgba" and wl["shader"] == "phong")  # Math.pow: +-1 LSB contract, not hash-pinned
                   and hashlib.sha256(b[name].cpu().numpy().tobytes()).hexdigest() != gold[name]]
            if bad:
                raise SystemExit("bench.py: buffer set %d of the timed region differs from the golden fixture %s in %s"
                                 % (k, wl["golden"], ", ".join(bad)))
            frames_verified += 1
        verify_note = ("SHA-256 of depth, normal, sdfEval, iters%s of each of the %d buffer sets in flight against "
                       "tests/golden/golden.json[%s]" % ("" if wl["shader"] == "phong" else ", RGBA", used, wl["golden"]))
        # ... and the diagnostics the timed frames produced (fused: by the render kernel) against the fixture's
        with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
            gd = json.load(f)[wl["golden"]].get("diagnostics")
        for k in range(used):
            dd = ctx.decode_acc(sets[k]["acc"])
            if gd and any(dd[key] != gd[key] for key in ("total_sdf", "total_iters", "max_sdf", "min_sdf")):
                raise SystemExit("bench.py: diagnostics of buffer set %d, %s, differ from the golden fixture's %s" % (k, dd, gd))

    # dominant kernel: the render kernel.  kern_ms_in_flight = mean launch duration inside the timed region (HIP events
    # on the launch stream; with S > 1 it spans the other frames' work too).
    kern_ms_in_flight = sum(a.elapsed_time(b) for a, b in ev_pairs) / max(1, len(ev_pairs))
    n_launches = len(ev_pairs)
    if world == 1:
        px_per_launch = W * H
    else:
        rows_mine = layout.rows(rank)
        launches = 1 if args.partition == "interleaved" else max(1, len(rows_mine))
        px_per_launch = W * sum(b - a for a, b in rows_mine) / launches
    bytes_per_launch = ALG_BYTES_PER_PIXEL * px_per_launch

    # The launch running ALONE (library defaults, serial, HIP events around each launch) and the strictly serial frame
    # rate (render + reduce back to back on one stream), both right after the timed region.  N = 1 only.
    kern_ms, value_serial, kernel_alone, kern_ms_cold = kern_ms_in_flight, None, kernel_in_flight, None
    if world == 1 and not args.analytics_sweep:
        b = sets[0]
        apply_opts(False)
        ser = []
        with torch.cuda.stream(streams[0]):
            # A configuration that keeps coming back gets the wave loop compiled for it (option specialise_v2_after, ~2 s on the
            # host at that launch): untimed launches first, so that no timed one contains the compile.  The "first launch"
            # figure (no item costs yet) is then taken with the longest-first order off -- the same thing.
            for _ in range(max(0, ctx.get_option("specialise_v2_after"))):
                tracer.runRaymarcher(scene, b["depth"], b["normal"], b["sdf"], b["iters"], W, H, 0.0, shadedBuffer=b["rgba"], shader=wl["shader"])
            torch.cuda.synchronize()
            lpt_was = ctx.get_option("lpt")
            ctx.set_option("lpt", 0)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            tracer.runRaymarcher(scene, b["depth"], b["normal"], b["sdf"], b["iters"], W, H, 0.0, shadedBuffer=b["rgba"], shader=wl["shader"])
            c1.record()
            torch.cuda.synchronize()
            ctx.set_option("lpt", lpt_was)
            for _ in range(2 + 6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                tracer.runRaymarcher(scene, b["depth"], b["normal"], b["sdf"], b["iters"], W, H, 0.0,
                                     shadedBuffer=b["rgba"], shader=wl["shader"])
                e1.record()
                ser.append((e0, e1))
                torch.cuda.synchronize()
            kernel_alone = ctx.last_kernel()
            kern_ms = sum(a.elapsed_time(c) for a, c in ser[2:]) / len(ser[2:])
            kern_ms_cold = c0.elapsed_time(c1)  # no item costs: the queues' own order
            n_ser = max(5, min(20, args.steps))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_ser):
                tracer.runRaymarcher(scene, b["depth"], b["normal"], b["sdf"], b["iters"], W, H, 0.0,
                                     shadedBuffer=b["rgba"], shader=wl["shader"], diagnostics=b["acc"] if fused else None)
                if not fused:
                    ctx.reduce_counters_enqueue(b["sdf"], b["iters"], b["acc"])
            torch.cuda.synchronize()
            value_serial = n_ser / (time.perf_counter() - t0)
        apply_opts(True)
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    achieved_in_flight = bytes_per_launch * n_launches / elapsed / 1e9 if elapsed > 0 else 0.0

    # HBM traffic and the VALU mix of the dominant kernel: PMC counters cannot be read from inside this process; the
    # per-launch figures come from profiles/r03/pmc_<workload>.json, which scripts/profile_r02.sh measured with rocprofv3
    # on serial launches (library defaults) -- used only if measured on the sources now in the tree, N = 1.
    traffic, valu, pmc_note = None, None, None
    if world == 1:
        pj, pmc_note = load_pmc(args.workload, user_opts)
        if pj:
            traffic = pj.get("traffic_bytes")
            valu = pj.get("valu")
            if valu and valu.get("weighted_issue_floor_ms") and kern_ms > 0:
                valu = dict(valu, frac=valu["weighted_issue_floor_ms"] / kern_ms,
                            frac_in_flight=valu["weighted_issue_floor_ms"] / (1e3 * elapsed / args.steps))

    if rank == 0:
        d = ctx.decode_acc(last_acc[0] if world == 1 else acc)
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
                       "row-tile shard x%d (%s, stripe %d, rank-0 share weighted) + RCCL gather of RGBA and per-rank "
                       "diagnostics sums to rank 0" % (world, args.partition, args.stripe)},
            "setup_frames": setup_frames, "value_serial": value_serial, "diagnostics": args.diagnostics,
            "frames_verified": frames_verified, "frames_verified_how": verify_note,
            "host_enqueue_ms": host_enqueue_ms if world > 1 else host_enqueue_ms[0],
            "avg_sdf_calls_per_pixel": d["total_sdf"] / (W * H), "avg_iterations_per_pixel": d["total_iters"] / (W * H),
            "max_sdf_calls": d["max_sdf"], "min_sdf_calls": d["min_sdf"],
            "sphere_evals_per_s": d["total_sdf"] * fps,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "pmc_source": pmc_note,
                         "kernel": kernel_alone, "kernel_ms": kern_ms, "kernel_ms_first_launch": kern_ms_cold,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "basis": "the launch ALONE: algorithmic bytes per launch / mean duration of %s (HIP events on the "
                                  "launch stream)" % ("six serial launches of the same frame with the library's default options, right "
                                                      "after the timed region; by default (option lpt) a launch hands out its work "
                                                      "items longest-first using the item costs the previous launch recorded -- "
                                                      "kernel_ms_first_launch is the launch without such costs"
                                                      if world == 1 and not args.analytics_sweep else "the timed region's launches"),
                         "achieved_in_flight": achieved_in_flight, "frac_in_flight": achieved_in_flight / HBM_PEAK_GBPS,
                         "kernel_in_flight": kernel_in_flight, "kernel_ms_in_flight": kern_ms_in_flight,
                         "frames_in_flight": S, "in_flight_options": {k: v for k, (v, _) in in_flight_opts.items()},
                         "tail_ramp": (args.tail_ramp if (world == 1 and "blocks_per_cu" in in_flight_opts) else 0),
                         "basis_in_flight": "device level: algorithmic bytes per launch x %d launches / wall time of the "
                                            "timed region (launches overlap)" % n_launches,
                         "valu": valu,
                         "note": "FP64-VALU issue / divergence bound, not HBM bound: 12 B/pixel out, ~6e3 lane-instructions "
                                 "per pixel; `valu` prices the measured instruction mix with the measured issue costs "
                                 "(DESIGN.md 4)"},
        }
        if root_balance:
            out["config"]["root_balance"] = root_balance
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
            if "golden" in wl and not args.no_verify:  # ... and the committed fixture (C = JS oracle agreement, tests/golden/)
                with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
                    g = json.load(f)[wl["golden"]]
                ok = wl["shader"] == "phong" or hashlib.sha256(asm.frame["rgba"].cpu().numpy().tobytes()).hexdigest() == g["sha256"]["rgba"]
                gd = g.get("diagnostics")
                ok = ok and (not gd or all(d[key] == gd[key] for key in ("total_sdf", "total_iters", "max_sdf", "min_sdf")))
                out["frames_verified"] = 1 if ok else 0
                out["frames_verified_how"] = ("SHA-256 of the gathered, reassembled RGBA frame and the combined diagnostics of the last frame of the "
                                              "timed region against tests/golden/golden.json[%s]" % wl["golden"])
                if not ok:
                    raise SystemExit("bench.py: the gathered frame or its diagnostics differ from the golden fixture %s" % wl["golden"])
        if world == 1 and args.analytics_sweep:
            out["config"]["camera"] = {"pitch": 0.0, "yaw": "+0.015 rad per frame (analytics sweep)"}
            out["analytics_series"] = [
                {"yaw": round(y, 6), "avg_sdf_calls": ctx.decode_acc(a)["total_sdf"] / (W * H),
                 "avg_iterations": ctx.decode_acc(a)["total_iters"] / (W * H), "max_sdf_calls": ctx.decode_acc(a)["max_sdf"],
                 ("frame_ms" if S == 1 else "frame_ms_overlapped"): e0.elapsed_time(e1)} for (y, a), (e0, e1) in zip(series, ev_pairs)]
        if world == 1 and not args.no_cpu_baseline and not args.analytics_sweep:
            out["cpu_baseline"] = cpu_baseline_port(wl)
            js = cpu_baseline_js_pool(wl)
            if js:
                out["cpu_baseline"]["reference_policy_js"] = js
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
