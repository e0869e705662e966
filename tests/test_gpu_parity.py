"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on
the same inputs and against the committed golden fixtures at BASELINE.json's full sizes.

Bar (BASELINE.json north_star): integer counters bit-exact; RGBA within 1 LSB per channel.
In practice every buffer, Phong included, is bit-identical; the Phong test still states the
+-1 tolerance because Math.pow is not correctly rounded on either side."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ("depth", "normal", "sdf", "iters")


def gpu_render(rm, ctx, preset, accel, W, H, ang=(0.0, 0.0), rows=None, spheres=None, algorithm="sphere-tracer",
               overshoot=None, step=None, prims=None, time=0.0, nodes=None):
    sc = rm.Scene(accel, ctx=ctx)
    if nodes is not None:  # (node tuples, roots) as OracleScene.nodes() returns them
        sc.loadNodes(*nodes)
    elif prims is not None:  # (type, world_to_local, params) triples, e.g. OracleScene.prims()
        sc.loadPrims(prims)
    elif spheres is not None:
        sc.loadSpheres(spheres[:, :3], spheres[:, 3])
    else:
        sc.loadPreset(preset)
    sc.camera.setAngles(*ang)
    y0, y1 = rows if rows else (0, H)
    n = W * max(0, y1 - y0)
    bufs = (np.zeros(n, np.uint8), np.zeros(3 * n, np.uint8), np.zeros(n, np.uint16), np.zeros(n, np.uint16))
    rm.createRaymarcher(algorithm, overshoot, step).runRaymarcher(sc, *bufs, W, H, time, y0, y1)
    return bufs


def cpu_render(oracle, preset, accel, W, H, ang=(0.0, 0.0), rows=None, spheres=None, algorithm="sphere-tracer",
               overshoot=None, step=None, prims=None, time=0.0):
    sc = oracle.OracleScene(preset=preset, accel=accel, spheres=spheres, prims=prims)
    sc.set_angles(*ang)
    y0, y1 = rows if rows else (0, H)
    return sc.render(W, H, y0, y1, algorithm=algorithm, overshoot_factor=overshoot, step_size=step, time=time)


def assert_same(got, want, what):
    for name, g, w in zip(NAMES, got, want):
        bad = int((g != w).sum())
        assert bad == 0, "%s: %s differs in %d of %d entries" % (what, name, bad, w.size)


@pytest.fixture(params=[1, 0], ids=["specialised", "ahead-of-time"])
def specialise(request, gpu_ctx):
    """Small scenes -- expression forests, plain primitive lists, fewer than eight spheres -- run as the scene's run-time
    compiled kernel (rm_rtc.h; the default) or through the library's ahead-of-time kernels (the interpreter of rm_program.h for
    forests; option `specialise` = 0): both must equal the oracle."""
    gpu_ctx.set_option("specialise", request.param)
    yield request.param
    gpu_ctx.set_option("specialise", 1)


def test_device_hypot_is_v8_math_hypot(rm, gpu_ctx, oracle):
    rng = np.random.default_rng(11)
    xyz = (rng.standard_normal((300000, 3)) * 3).astype(np.float32)
    xyz[:2000] *= np.float32(1e-18)   # tiny magnitudes
    xyz[2000:4000, 0] = 0              # zeros in each slot
    xyz[4000:6000, 2] = 0
    xyz[6000:6100] = 0                 # max == 0 -> 0
    xyz[6100:8100] *= np.float32(1e15)
    xyz[8100:9000, 1] = xyz[8100:9000, 0]  # equal magnitudes
    got = gpu_ctx.selftest_hypot(xyz)
    L = oracle.lib()
    sel = np.r_[0:9000, rng.integers(9000, len(xyz), 40000)]
    want = np.array([L.ro_hypot3(float(a), float(b), float(c)) for a, b, c in xyz[sel]])
    assert np.array_equal(got[sel], want)


def test_shared_reciprocal_division_is_bit_identical(gpu_ctx):
    """v2's sphere SDF shares one reciprocal refinement between the three divisions of
    Math.hypot; on 3e8 generated triples it must never differ from the IEEE divisions."""
    assert gpu_ctx.selftest_fastdiv(0x1234, 300_000_000) == 0
    assert gpu_ctx.selftest_fastdiv(0xBEEF, 50_000_000) == 0


def test_ray_setup_division_is_the_ieee_division(gpu_ctx):
    """Ray set-up divides without the scale / fix-up bracket of the compiler's IEEE expansion where the operand ranges
    make it a no-op.  Exhaustive: every finite non-zero binary32 denominator of 1.0 / d, and every x / W with
    0 <= x < 65536, 1 <= W < 65536 (2^32 cases each)."""
    assert gpu_ctx.selftest_recip(0) == 0
    assert gpu_ctx.selftest_recip(1) == 0


def test_v1_and_v2_kernels_agree(rm, oracle):
    """Two independently structured kernels (divergent per-ray loops with native divisions vs
    the uniform wave loop with lists / cooperative fallback / shared reciprocal) must produce
    the same bytes, with every option combination."""
    ctx = rm.Context(0)
    sp = oracle.synthetic_spheres(3000)
    for preset, accel, spheres in [(3, "BVH", None), (3, "Octree", None), (3, "None", None), (None, "BVH", sp),
                                   (None, "Octree", sp)]:
        outs = []
        for opts in [dict(kernel=1), dict(kernel=2), dict(kernel=2, coop=0), dict(kernel=2, filter=0, coop=0),
                     dict(kernel=2, nodes_in_lds=0), dict(kernel=2, list_cap=2), dict(kernel=2, tile_w=64),
                     dict(kernel=2, grid=0), dict(kernel=2, refill=8), dict(kernel=2, refill=24, hw_xcd=0),
                     dict(kernel=2, refill=1, blocks_per_cu=1), dict(kernel=0), dict(kernel=1, recs=0),
                     dict(kernel=1, filter=0), dict(kernel=1, lut=0), dict(kernel=1, sub=0), dict(kernel=2, nn=0),
                     dict(kernel=2, item_px=64), dict(kernel=2, item_px=256, tile_w=32), dict(kernel=2, item_px=128, tile_w=64),
                     dict(kernel=2, static=75), dict(kernel=2, static=95, item_px=64),
                     dict(kernel=2, nn=1, coop=0), dict(kernel=2, uniform=0), dict(kernel=2, uniform=0, list_cap=2),
                     dict(kernel=2, rel=0), dict(kernel=2, rel=0, uniform=0), dict(kernel=2, rel=1, list_cap=2),
                     dict(kernel=2, item_px=256, tile_w=8, blocks_per_cu=2), dict(kernel=2, lds_kb=40), dict(kernel=2, lds_kb=16), dict(kernel=2, lds_kb=32),
                     dict(kernel=2, lds_kb=64, rel=0), dict(kernel=2, cull=0), dict(kernel=2, cull=0, rel=0),
                     dict(kernel=2, n0_batch=1), dict(kernel=2, n0_batch=16, refill=24), dict(kernel=2, n0_batch=8, uniform=0, cull=0),
                     dict(kernel=2, lpt=0), dict(kernel=2, lpt=1), dict(kernel=2, lpt=1, item_px=64, tile_w=32),
                     dict(kernel=2, multi_step=0), dict(kernel=2, multi_step=0, uniform=0, rel=0), dict(kernel=2, multi_step=1, n0_batch=8, refill=24),
                     dict(kernel=1, oct_lean=0), dict(kernel=1, oct_lean=0, v1_block=256, tile_w=16), dict(kernel=1, v1_block=128, tile_w=32)]:
            for k, v in dict(kernel=2, coop=1, filter=1, nodes_in_lds=1, list_cap=32, tile_w=8, grid=1, refill=64,
                             hw_xcd=1, blocks_per_cu=3, recs=1, lut=1, nn=2, sub=1, item_px=128, static=0, uniform=1, rel=1, lds_kb=0, cull=1,
                             n0_batch=64, lpt=1, oct_lean=1, v1_block=64, multi_step=1).items():
                ctx.set_option(k, v)
            for k, v in opts.items():
                ctx.set_option(k, v)
            outs.append(gpu_render(rm, ctx, preset, accel, 300, 170, (0.25, 0.6), spheres=spheres))
        for o in outs[1:]:
            assert_same(o, outs[0], "%s %s variants" % (preset, accel))
        assert_same(outs[0], cpu_render(oracle, preset, accel, 300, 170, (0.25, 0.6), spheres=spheres), "vs oracle")


def test_one_radius_scenes_rank_by_squared_distance(rm, oracle):
    """Scenes whose spheres share one radius take the v2 path that ranks leaf candidates by squared centre distance
    (option `uniform`): random clusters with overlaps, coincident centres (exact ties), a negative radius, and a
    scene that differs from uniform by one radius only (must take the general path)."""
    import numpy as np
    ctx = rm.Context(0)
    rng = np.random.default_rng(77)
    cases = []
    for n, r, spread in [(200, 0.3, 2.0), (64, 0.05, 0.6), (300, 0.5, 1.5), (40, -0.2, 1.0)]:
        c = rng.uniform(-spread, spread, size=(n, 3)).astype(np.float32)
        c[1] = c[0]                    # coincident centres
        c[3] = c[2] + np.float32(1e-7)  # and a near tie
        cases.append(np.concatenate([c, np.full((n, 1), r, np.float32)], axis=1).astype(np.float64))
    almost = cases[0].copy()
    almost[17, 3] = np.float64(np.float32(0.3000001))
    cases.append(almost)
    for sp in cases:
        for accel in ("BVH", "None"):
            outs = []
            for opts in (dict(kernel=2, uniform=1, rel=1), dict(kernel=2, uniform=0), dict(kernel=2, rel=0), dict(kernel=1)):
                for k, v in opts.items():
                    ctx.set_option(k, v)
                outs.append(gpu_render(rm, ctx, None, accel, 200, 120, (0.2, 0.5), spheres=sp))
            for o in outs[1:]:
                assert_same(o, outs[0], "uniform radius variants %s" % accel)
            assert_same(outs[0], cpu_render(oracle, None, accel, 200, 120, (0.2, 0.5), spheres=sp), "vs oracle")


def test_bundle_cull_matches_tree_walk(rm, oracle):
    """Whole 64-pixel batches find their hit BVH leaves by a bundle-frustum cull (option `cull`) instead of the tree
    walk: same bytes as the walk and as the oracle for frames with partial tiles, row sub-ranges, several tile shapes,
    cameras from every side, and boxes that contain the camera."""
    import numpy as np
    ctx = rm.Context(0)
    rng = np.random.default_rng(404)
    scenes = []
    c = rng.uniform(-2.5, 2.5, size=(150, 3))
    scenes.append(np.concatenate([c, rng.uniform(0.05, 0.5, size=(150, 1))], axis=1))
    big = np.concatenate([rng.uniform(-1.5, 1.5, size=(40, 3)), rng.uniform(0.1, 0.4, size=(40, 1))], axis=1)
    big[0] = (0.0, 0.0, 0.0, 7.0)   # encloses the camera and every other sphere
    big[1] = (0.0, 0.0, 4.5, 1.5)   # around the default camera position
    scenes.append(big)
    scenes.append(np.concatenate([rng.uniform(-0.2, 0.2, size=(30, 3)) * (1, 30, 1), np.full((30, 1), 0.15)], axis=1))  # a tall thin column
    # small spheres strung along the left, right and bottom edges of the default camera's view pyramid, starting 3 cm
    # from the camera (orbit radius 3, camera.ts:13): boxes that graze the cull's planes next to the apex
    s_ = np.concatenate([np.linspace(0.03, 0.3, 12), np.linspace(0.4, 4.0, 12)])
    edge = [np.stack([-s_, 0 * s_, 3.0 - s_], axis=1), np.stack([s_, 0.3 * s_, 3.0 - s_], axis=1),
            np.stack([0.2 * s_, -s_, 3.0 - s_], axis=1)]
    edge = np.concatenate(edge)
    scenes.append(np.concatenate([edge, np.full((len(edge), 1), 0.02)], axis=1))
    for sp in scenes:
        sp = sp.astype(np.float32).astype(np.float64)
        for (W, H, rows, ang) in [(77, 53, None, (0.0, 0.0)), (130, 70, (13, 59), (0.4, 2.2)), (96, 64, None, (-1.2, 4.0)),
                                  (64, 48, None, (1.5, 0.3))]:
            outs = []
            for opts in (dict(cull=1), dict(cull=0), dict(cull=1, tile_w=8, item_px=256), dict(cull=1, tile_w=32, item_px=64),
                         dict(cull=1, rel=0, uniform=0), dict(cull=1, refill=24), dict(cull=1, list_cap=2), dict(kernel=1)):
                for k, v in dict(kernel=2, cull=1, tile_w=16, item_px=128, rel=1, uniform=1, refill=64, list_cap=32).items():
                    ctx.set_option(k, v)
                for k, v in opts.items():
                    ctx.set_option(k, v)
                outs.append(gpu_render(rm, ctx, None, "BVH", W, H, ang, rows=rows, spheres=sp))
            for o in outs[1:]:
                assert_same(o, outs[0], "bundle cull variants %dx%d" % (W, H))
            assert_same(outs[0], cpu_render(oracle, None, "BVH", W, H, ang, rows=rows, spheres=sp), "vs oracle")


def test_frames_in_flight_on_separate_streams(rm):
    """bench.py keeps several frames in flight: one context, one HIP stream and buffer set per frame, different
    cameras.  Every frame must equal the same frame rendered alone (tile-counter ring, per-launch parameters)."""
    import torch
    W, H = 1280, 720
    dev = torch.device("cuda:0")
    ctx = rm.Context(0)
    ctx.set_option("static", 75)
    scene = rm.Scene("BVH", ctx=ctx)
    scene.loadPreset(3)
    tracer = rm.SphereTracer()
    yaws = [0.05 * k for k in range(8)]

    def buffers():
        return dict(depth=torch.zeros(W * H, dtype=torch.uint8, device=dev), normal=torch.zeros(3 * W * H, dtype=torch.uint8, device=dev),
                    sdf=torch.zeros(W * H, dtype=torch.int16, device=dev), iters=torch.zeros(W * H, dtype=torch.int16, device=dev),
                    rgba=torch.zeros(4 * W * H, dtype=torch.uint8, device=dev))

    def render(b, yaw):
        scene.camera.setAngles(0.1, yaw)
        tracer.runRaymarcher(scene, b["depth"], b["normal"], b["sdf"], b["iters"], W, H, 0.0, shadedBuffer=b["rgba"], shader="phong")

    alone = []
    for yaw in yaws:
        b = buffers()
        render(b, yaw)
        torch.cuda.synchronize()
        alone.append({k: v.clone() for k, v in b.items()})
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    sets = [buffers() for _ in yaws]
    for rep in range(3):  # every stream is reused while the others are busy
        for k, yaw in enumerate(yaws):
            with torch.cuda.stream(streams[k % 4]):
                render(sets[k], yaw)
    torch.cuda.synchronize()
    for k in range(len(yaws)):
        for name in sets[k]:
            assert torch.equal(sets[k][name], alone[k][name]), (k, name)
    ctx.close()


def test_lean_octree_kernel_and_its_camera_tables(rm, oracle):
    """render_kernel_oct (sphere scenes, octree with the cell table) against render_kernel<1, false, 0> and the oracle:
    cameras inside and outside the root cube (the all-primitive fallback), rays parallel to an axis (zero direction
    components: infinite 1 / d), more camera positions than the ring of origin-relative node tables holds, and frames
    in flight on several streams that share and replace those tables."""
    import torch
    ctx = rm.Context(0)
    sp = oracle.synthetic_spheres(1500)
    W, H = 200, 120
    for preset, spheres, angs in [(3, None, [(0.0, 0.0), (0.3, 0.7), (-1.2, 2.5), (1.5, 0.0)]), (None, sp, [(0.0, 0.0), (0.25, 0.6), (-0.4, 3.0)]),
                                  (2, None, [(0.0, 0.0), (0.2, -0.9)])]:
        for ang in angs:
            outs = {}
            for lean in (1, 0):
                ctx.set_option("oct_lean", lean)
                outs[lean] = gpu_render(rm, ctx, preset, "Octree", W, H, ang, spheres=spheres)
                assert ("render_kernel_oct" in ctx.last_kernel()) == bool(lean), ctx.last_kernel()
            assert_same(outs[1], outs[0], "lean vs generic octree kernel %s %s" % (preset, ang))
            assert_same(outs[1], cpu_render(oracle, preset, "Octree", W, H, ang, spheres=spheres), "lean octree kernel vs oracle %s %s" % (preset, ang))
    ctx.set_option("oct_lean", 1)
    # 40 camera positions > 32 table slots, four streams, every frame against the same frame rendered alone
    dev = torch.device("cuda:0")
    scene = rm.Scene("Octree", ctx=ctx)
    scene.loadSpheres(sp[:, :3], sp[:, 3])
    tracer = rm.SphereTracer()
    yaws = [0.16 * k for k in range(40)]

    def buffers():
        return [torch.zeros(W * H, dtype=torch.uint8, device=dev), torch.zeros(3 * W * H, dtype=torch.uint8, device=dev),
                torch.zeros(W * H, dtype=torch.int16, device=dev), torch.zeros(W * H, dtype=torch.int16, device=dev)]

    def render(b, yaw):
        scene.camera.setAngles(0.2, yaw)
        tracer.runRaymarcher(scene, *b, W, H, 0.0)

    alone = []
    for yaw in yaws:
        b = buffers()
        render(b, yaw)
        torch.cuda.synchronize()
        alone.append([v.clone() for v in b])
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    sets = [buffers() for _ in yaws]
    for rep in range(2):
        for k, yaw in enumerate(yaws):
            with torch.cuda.stream(streams[k % 4]):
                render(sets[k], yaw)
                render(sets[k], yaw)  # the same camera again: shares the table
    torch.cuda.synchronize()
    for k in range(len(yaws)):
        for a, b in zip(sets[k], alone[k]):
            assert torch.equal(a, b), k
    ctx.close()


def test_octree_camera_table_is_built_before_another_stream_reads_it(rm, oracle):
    """the per-camera octree table is built on the stream of the FIRST launch that sees a camera
    position; a launch with the same position on another stream must wait for that build (bench.py's C5 line keeps twelve
    frames of one camera in flight).  Stream A is kept busy so that its table build is still queued when streams B and C
    launch the same camera; every frame must equal the frame rendered alone."""
    import torch
    ctx = rm.Context(0)
    sp = oracle.synthetic_spheres(10000)
    W, H = 480, 270
    dev = torch.device("cuda:0")
    scene = rm.Scene("Octree", ctx=ctx)
    scene.loadSpheres(sp[:, :3], sp[:, 3])
    tracer = rm.SphereTracer()
    yaws = [0.37 * k + 0.05 for k in range(6)]

    def buffers(w=W, h=H):
        return [torch.zeros(w * h, dtype=torch.uint8, device=dev), torch.zeros(3 * w * h, dtype=torch.uint8, device=dev),
                torch.zeros(w * h, dtype=torch.int16, device=dev), torch.zeros(w * h, dtype=torch.int16, device=dev)]

    def render(b, yaw, w=W, h=H):
        scene.camera.setAngles(0.15, yaw)
        tracer.runRaymarcher(scene, *b, w, h, 0.0)
        assert "render_kernel_oct" in ctx.last_kernel(), ctx.last_kernel()

    alone = []
    for yaw in yaws:
        b = buffers()
        render(b, yaw)
        torch.cuda.synchronize()
        alone.append([v.clone() for v in b])
    ctx.close()
    # a fresh context: no table exists for any of these positions yet
    ctx = rm.Context(0)
    scene = rm.Scene("Octree", ctx=ctx)
    scene.loadSpheres(sp[:, :3], sp[:, 3])
    A, B, C = (torch.cuda.Stream(device=dev) for _ in range(3))
    busy = buffers(1920, 1080)
    sets = [[buffers() for _ in range(3)] for _ in yaws]
    for k, yaw in enumerate(yaws):
        with torch.cuda.stream(A):
            render(busy, yaw + 0.011, 1920, 1080)  # ~1 ms of work in front of the table build
            render(sets[k][0], yaw)               # builds the table for `yaw` on stream A
        with torch.cuda.stream(B):
            render(sets[k][1], yaw)               # same position, at once, on another stream
        with torch.cuda.stream(C):
            render(sets[k][2], yaw)
    torch.cuda.synchronize()
    for k in range(len(yaws)):
        for s in range(3):
            for a, b in zip(sets[k][s], alone[k]):
                assert torch.equal(a, b), (k, s)
    ctx.close()


def test_static_tile_share_on_a_frame_large_enough_to_use_it(rm):
    """The statically assigned part of the tile queues only exists when a frame has more items than one round
    over all waves: 1920x1080 gives two rounds.  Every share must produce the same bytes."""
    ctx = rm.Context(0)
    outs = []
    for opts in (dict(static=0), dict(static=75), dict(static=95), dict(static=75, item_px=64), dict(static=50, tile_w=32)):
        for k, v in dict(kernel=2, static=0, item_px=128, tile_w=16).items():
            ctx.set_option(k, v)
        for k, v in opts.items():
            ctx.set_option(k, v)
        outs.append(gpu_render(rm, ctx, 3, "BVH", 1920, 1080, (0.1, 0.3)))
    for o in outs[1:]:
        assert_same(o, outs[0], "static tile share")
    ctx.close()


@pytest.mark.parametrize("n,seed", [(30, 1), (120, 2), (300, 3), (500, 4)])
def test_random_sphere_scenes_with_candidate_lists(rm, oracle, n, seed):
    """Scenes of <= 512 spheres use the nearest-candidate grid for the all-primitive fallback: random
    centres and very unequal radii (the candidate bound involves the radius) against the oracle,
    with the lists forced on, forced off and in the v1 kernel."""
    rng = np.random.default_rng(seed)
    sp = np.zeros((n, 4))
    sp[:, :3] = rng.uniform(-1.4, 1.4, (n, 3)).astype(np.float32)
    sp[:, 3] = rng.choice([0.02, 0.05, 0.12, 0.3, 0.5], n) * rng.uniform(0.8, 1.2, n)
    ctx = rm.Context(0)
    want = cpu_render(oracle, None, "BVH", 240, 150, (0.35, -0.8), spheres=sp)
    for opts in (dict(kernel=2, nn=1), dict(kernel=2, nn=0), dict(kernel=2, nn=1, coop=0, filter=0), dict(kernel=1)):
        for k, v in dict(kernel=0, nn=2, coop=1, filter=1).items():
            ctx.set_option(k, v)
        for k, v in opts.items():
            ctx.set_option(k, v)
        assert_same(gpu_render(rm, ctx, None, "BVH", 240, 150, (0.35, -0.8), spheres=sp), want, "n=%d %s" % (n, opts))
    ctx.close()


@pytest.mark.parametrize("accel", ["None", "BVH", "Octree"])
def test_degenerate_radii(rm, gpu_ctx, oracle, accel):
    """Zero and negative radii are legal input (Sphere.localSdf only subtracts the radius): the conservative
    filter's error term must not shrink with them."""
    rng = np.random.default_rng(9)
    sp = np.zeros((60, 4))
    sp[:, :3] = rng.uniform(-1.2, 1.2, (60, 3)).astype(np.float32)
    sp[:, 3] = rng.choice([0.0, -0.3, -2.5, 0.2, 0.4, 1e-9], 60)
    got = gpu_render(rm, gpu_ctx, None, accel, 200, 130, (0.2, 0.4), spheres=sp)
    assert_same(got, cpu_render(oracle, None, accel, 200, 130, (0.2, 0.4), spheres=sp), "degenerate radii " + accel)


@pytest.mark.parametrize("accel", ["None", "BVH", "Octree"])
@pytest.mark.parametrize("preset", [0, 1, 2, 3, 4])
def test_every_sphere_preset_and_accel(rm, gpu_ctx, oracle, specialise, preset, accel):
    got = gpu_render(rm, gpu_ctx, preset, accel, 200, 150)
    if preset in (0, 1, 4) and not (accel == "Octree"):  # fewer than eight spheres: the one-ray-per-lane kernels, the scene's own by default
        assert gpu_ctx.last_kernel().startswith("rm_rtc_render<" if specialise else "render_kernel<"), gpu_ctx.last_kernel()
    assert_same(got, cpu_render(oracle, preset, accel, 200, 150), "preset %d %s" % (preset, accel))


@pytest.mark.parametrize("accel", ["None", "BVH", "Octree"])
@pytest.mark.parametrize("alg,overshoot,step", [("fixed-step", None, None), ("fixed-step", None, 0.03),
                                                 ("adaptive-step", None, None), ("adaptive-step-v2", None, None),
                                                 ("adaptive-step-v2", 1.8, None), ("adaptive-step-v3", None, None),
                                                 ("adaptive-step-v3", 1.05, None)])
def test_other_marchers(rm, gpu_ctx, oracle, specialise, accel, alg, overshoot, step):
    """The rest of the Algorithm plugin (raymarchWorker.ts:49-68): same accel prologue / skip
    protocol, different step rule; counters and G-buffers bit-exact against the oracle."""
    for preset, ang in ((3, (0.2, 0.5)), (1, (0.0, 0.0))):
        got = gpu_render(rm, gpu_ctx, preset, accel, 220, 140, ang, algorithm=alg, overshoot=overshoot, step=step)
        want = cpu_render(oracle, preset, accel, 220, 140, ang, algorithm=alg, overshoot=overshoot, step=step)
        assert_same(got, want, "%s %s preset %d" % (alg, accel, preset))


@pytest.mark.parametrize("accel,ang", [("BVH", (0.3, 0.7)), ("Octree", (-0.4, 2.1)), ("BVH", (1.2, -0.3)),
                                       ("Octree", (9.0, 3.3)), ("None", (-1.0, 5.0))])
def test_rotated_camera(rm, gpu_ctx, oracle, accel, ang):
    got = gpu_render(rm, gpu_ctx, 3, accel, 240, 180, ang)
    assert_same(got, cpu_render(oracle, 3, accel, 240, 180, ang), "dense grid %s %r" % (accel, ang))


def test_scene_get_distance_batch(rm, gpu_ctx, oracle):
    rng = np.random.default_rng(5)
    pts = rng.uniform(-2, 2, (3000, 3)).astype(np.float32)
    for preset, accel in [(3, "BVH"), (3, "Octree"), (3, "None"), (1, "BVH"), (1, "Octree")]:
        sc = rm.Scene(accel, ctx=gpu_ctx)
        sc.loadPreset(preset)
        d, c = sc.getDistances(pts)
        osc = oracle.OracleScene(preset=preset, accel=accel)
        for k in range(0, len(pts), 7):
            wd, wc = osc.distance(pts[k])
            assert d[k] == wd and c[k] == wc, (preset, accel, k)
        counter = {"count": 0}
        assert sc.getDistance(pts[0], counter) == osc.distance(pts[0])[0] and counter["count"] == osc.distance(pts[0])[1]


def test_tile_independence_and_ragged_tiles(rm, gpu_ctx, oracle):
    W, H = 333, 77  # not multiples of any tile shape
    full = gpu_render(rm, gpu_ctx, 3, "BVH", W, H, (0.2, 0.5))
    assert_same(full, cpu_render(oracle, 3, "BVH", W, H, (0.2, 0.5)), "ragged frame")
    parts = [gpu_render(rm, gpu_ctx, 3, "BVH", W, H, (0.2, 0.5), rows=r) for r in ((0, 1), (1, 40), (40, 77))]
    for k, name in enumerate(NAMES):
        assert np.array_equal(full[k], np.concatenate([p[k] for p in parts])), name
    empty = gpu_render(rm, gpu_ctx, 3, "BVH", W, H, rows=(5, 5))
    assert all(b.size == 0 for b in empty)
    one = gpu_render(rm, gpu_ctx, 0, "None", 1, 1)
    assert_same(one, cpu_render(oracle, 0, "None", 1, 1), "1x1 frame")


def test_tile_shapes_do_not_change_results(rm, oracle):
    ctx = rm.Context(0)
    want = cpu_render(oracle, 3, "Octree", 130, 70, (0.1, -0.9))
    for tw in (8, 16, 32, 64):
        ctx.set_option("tile_w", tw)
        assert_same(gpu_render(rm, ctx, 3, "Octree", 130, 70, (0.1, -0.9)), want, "tile_w %d" % tw)


def test_many_primitives_fallback_and_u16_wrap(rm, gpu_ctx, oracle):
    sp = oracle.synthetic_spheres(10000)
    # no acceleration: every step counts 10 000 -> wraps mod 65536 (raymarcher.ts:119)
    got = gpu_render(rm, gpu_ctx, None, "None", 32, 32, rows=(15, 17), spheres=sp)
    want = cpu_render(oracle, None, "None", 32, 32, rows=(15, 17), spheres=sp)
    assert_same(got, want, "10k none")
    assert np.any(want[3].astype(np.int64) * 10000 > 65535)
    # BVH with the all-primitive fallback (scene.ts:173), divergent rays
    got = gpu_render(rm, gpu_ctx, None, "BVH", 160, 90, (-0.2, 0.4), spheres=sp)
    assert_same(got, cpu_render(oracle, None, "BVH", 160, 90, (-0.2, 0.4), spheres=sp), "10k bvh")
    got = gpu_render(rm, gpu_ctx, None, "Octree", 320, 180, spheres=sp)
    assert_same(got, cpu_render(oracle, None, "Octree", 320, 180, spheres=sp), "10k octree")


def _oracle_diag(want):
    sdf, it = want[2].astype(np.int64), want[3].astype(np.int64)
    if sdf.size == 0:
        return {"total_sdf": 0, "total_iters": 0, "max_sdf": 0, "min_sdf": 0xFFFFFFFF}
    return {"total_sdf": int(sdf.sum()), "total_iters": int(it.sum()), "max_sdf": int(sdf.max()), "min_sdf": int(sdf.min())}


@pytest.mark.parametrize("case", [
    dict(preset=3, accel="BVH", size=(301, 173), ang=(0.2, 0.5)),                      # v2 headline kernel, ragged tiles
    dict(preset=3, accel="BVH", size=(301, 173), ang=(0.2, 0.5), opts=dict(refill=24, item_px=64, blocks_per_cu=1)),
    dict(preset=3, accel="None", size=(200, 120)),                                     # v2 without acceleration
    dict(preset=2, accel="BVH", size=(333, 97)),                                       # v1 (small scene)
    dict(preset=3, accel="Octree", size=(250, 141), ang=(0.3, 0.7)),                   # lean octree kernel
    dict(preset=3, accel="Octree", size=(250, 141), opts=dict(oct_lean=0, v1_block=256, tile_w=16)),  # four-wave v1 workgroups
    dict(preset=9, accel="BVH", size=(160, 90)),                                       # boxes (GEN 1)
    dict(preset=17, accel="BVH", size=(96, 64)),                                       # expression program (GEN 2)
    dict(preset=3, accel="BVH", size=(150, 80), alg="adaptive-step-v2"),               # another marcher
    dict(synthetic=10000, accel="None", size=(32, 32), rows=(15, 17)),                 # Uint16Array wrap: sums of the WRAPPED values
    dict(preset=3, accel="BVH", size=(64, 48), rows=(20, 20)),                         # no pixel at all: neutral elements
])
def test_fused_diagnostics_equal_the_reduction(rm, oracle, case):
    """rm_render_attach_diagnostics: the render kernels accumulate the diagnostics of main.ts:528-548 from the registers
    they store the counters from.  Against the oracle's counters (sum / max / min over the very bytes it stores) and against
    rm_reduce_counters on the buffers; also with the counter buffers absent, and twice in a row (the accumulator blocks
    and the tile-queue heads are left zeroed by each launch's last wave)."""
    import torch
    dev = torch.device("cuda:0")
    ctx = rm.Context(0)
    for k, v in case.get("opts", {}).items():
        ctx.set_option(k, v)
    W, H = case["size"]
    y0, y1 = case.get("rows", (0, H))
    sp = oracle.synthetic_spheres(case["synthetic"]) if "synthetic" in case else None
    sc = rm.Scene(case["accel"], ctx=ctx)
    if sp is not None:
        sc.loadSpheres(sp[:, :3], sp[:, 3])
    else:
        sc.loadPreset(case["preset"])
    sc.camera.setAngles(*case.get("ang", (0.0, 0.0)))
    n = W * max(0, y1 - y0)
    alg = case.get("alg", "sphere-tracer")
    tracer = rm.createRaymarcher(alg, None, None)
    want = cpu_render(oracle, case.get("preset"), case["accel"], W, H, case.get("ang", (0.0, 0.0)), rows=(y0, y1), spheres=sp, algorithm=alg)
    wd = _oracle_diag(want)
    for rep in range(2):
        d = torch.zeros(max(1, n), dtype=torch.uint8, device=dev)
        nr = torch.zeros(max(1, 3 * n), dtype=torch.uint8, device=dev)
        s16 = torch.zeros(max(1, n), dtype=torch.int16, device=dev)
        i16 = torch.zeros(max(1, n), dtype=torch.int16, device=dev)
        acc = torch.full((4,), -1, dtype=torch.int64, device=dev)  # garbage: the kernel must write all of it
        tracer.runRaymarcher(sc, d, nr, s16, i16, W, H, 0.0, y0, y1, diagnostics=acc)
        torch.cuda.synchronize()
        got = ctx.decode_acc(acc)
        assert got == wd, (rep, got, wd)
        if n:
            red = ctx.reduce_counters(s16[:n], i16[:n])
            assert {k: red[k] for k in wd} == wd
            assert np.array_equal(s16[:n].cpu().numpy().view(np.uint16), want[2]) and np.array_equal(i16[:n].cpu().numpy().view(np.uint16), want[3])
        # the counters need not be stored at all
        acc2 = torch.full((4,), -1, dtype=torch.int64, device=dev)
        rgba = torch.zeros(max(1, 4 * n), dtype=torch.uint8, device=dev)
        tracer.runRaymarcher(sc, None, None, None, None, W, H, 0.0, y0, y1, shadedBuffer=rgba, shader="iteration-heatmap", diagnostics=acc2)
        torch.cuda.synchronize()
        assert ctx.decode_acc(acc2) == wd
        if n:
            assert np.array_equal(rgba[:4 * n].cpu().numpy(), oracle.shade("iteration-heatmap", *want, W, y1 - y0))
    ctx.close()


def test_fused_diagnostics_of_stripes_and_frames_in_flight(rm, oracle):
    """The sharded entry points (one launch per rank's stripes, equal and weighted deals) and several frames in flight on
    several streams, each with its own accumulator: every accumulator equals the reduction of that launch's own counters."""
    import torch
    dev = torch.device("cuda:0")
    ctx = rm.Context(0)
    W, H, stripe, world = 320, 203, 7, 3
    sc = rm.Scene("BVH", ctx=ctx)
    sc.loadPreset(3)
    sc.camera.setAngles(0.1, 0.4)
    from cpu_raymarcher_amd import host
    job = host._job(sc, W, H, 0.0, 0, H, "sphere-tracer", None, None)
    total = {"total_sdf": 0, "total_iters": 0, "max_sdf": 0, "min_sdf": 0xFFFFFFFF}
    from cpu_raymarcher_amd.context import deal_stripes
    owner = deal_stripes(H, stripe, world, [500, 1000, 1300])
    for part in range(world):
        for mode in ("round-robin", "list"):
            if mode == "round-robin":
                rows = rm._native.lib().rm_stripe_rows(0, H, stripe, world, part)
            else:
                ids = [s_ for s_ in range(len(owner)) if owner[s_] == part]
                rows = sum(min(H, (i + 1) * stripe) - i * stripe for i in ids)
            s16 = torch.zeros(rows * W, dtype=torch.int16, device=dev)
            i16 = torch.zeros(rows * W, dtype=torch.int16, device=dev)
            acc = torch.full((4,), -1, dtype=torch.int64, device=dev)
            if mode == "round-robin":
                ctx.render_stripes(job, stripe, world, part, None, None, s16, i16, diag=acc)
            else:
                ctx.render_stripe_list(job, stripe, ids, None, None, s16, i16, diag=acc)
            torch.cuda.synchronize()
            red = ctx.reduce_counters(s16, i16)
            got = ctx.decode_acc(acc)
            assert got == {k: red[k] for k in got}, (part, mode)
            if mode == "list":
                total["total_sdf"] += got["total_sdf"]
                total["total_iters"] += got["total_iters"]
                total["max_sdf"] = max(total["max_sdf"], got["max_sdf"])
                total["min_sdf"] = min(total["min_sdf"], got["min_sdf"])
    assert total == _oracle_diag(cpu_render(oracle, 3, "BVH", W, H, (0.1, 0.4)))  # partial results combine exactly
    # frames in flight: 4 streams x 3 rounds, different cameras, one accumulator per launch
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    tracer = rm.SphereTracer()
    for k in ("blocks_per_cu", "item_px", "tile_w"):
        ctx.set_option(k, {"blocks_per_cu": 1, "item_px": 256, "tile_w": 8}[k])
    runs = []
    for f in range(12):
        with torch.cuda.stream(streams[f % 4]):
            sc.camera.setAngles(0.05 * f, 0.3 * f)
            s16 = torch.zeros(W * H, dtype=torch.int16, device=dev)
            i16 = torch.zeros(W * H, dtype=torch.int16, device=dev)
            acc = torch.full((4,), -1, dtype=torch.int64, device=dev)
            tracer.runRaymarcher(sc, None, None, s16, i16, W, H, 0.0, diagnostics=acc)
            runs.append((s16, i16, acc))
    torch.cuda.synchronize()
    for f, (s16, i16, acc) in enumerate(runs):
        red = ctx.reduce_counters(s16, i16)
        got = ctx.decode_acc(acc)
        assert got == {k: red[k] for k in got}, f
    ctx.close()


def test_empty_scene_and_bad_inputs(rm, gpu_ctx, oracle):
    empty = np.zeros((0, 4))
    for accel in ("None", "BVH", "Octree"):
        got = gpu_render(rm, gpu_ctx, None, accel, 40, 30, spheres=empty)
        assert_same(got, cpu_render(oracle, None, accel, 40, 30, spheres=empty), "empty scene " + accel)
    sc = rm.Scene("BVH", ctx=gpu_ctx)
    bufs = (np.zeros(16, np.uint8), np.zeros(48, np.uint8), np.zeros(16, np.uint16), np.zeros(16, np.uint16))
    sc.loadPreset(10)  # every preset is native
    with pytest.raises(rm.RmError):
        rm.SphereTracer().runRaymarcher(sc, *bufs, 4, 4, float("inf"))  # non-finite Job.time
    ident = np.eye(4, dtype=np.float32).ravel()
    moved = ident.copy()
    moved[12] = 0.25  # a Round takes its operand's transform: translated, every level needs a position slot (identity ones pass through)
    with pytest.raises(rm.RmUnsupported):  # deeper than the interpreter's position-slot file
        sc.loadNodes([(0, -1, -1, moved, [0.5])] + [(10, i, -1, None, [0.01]) for i in range(20)], [20])
    sc.camera.pitch = float("nan")
    with pytest.raises(rm.RmError):
        rm.SphereTracer().runRaymarcher(sc, *bufs, 4, 4)


@pytest.mark.parametrize("shader", ["normal", "phong", "sdf-heatmap", "iteration-heatmap", "unknown-name"])
def test_shading_models(rm, gpu_ctx, oracle, shader):
    W, H = 256, 160
    d, n, s, i = cpu_render(oracle, 3, "BVH", W, H, (0.3, 0.7))
    want = oracle.shade(shader, d, n, s, i, W, H)
    rgba = np.zeros(4 * W * H, np.uint8)
    rm.createShadingModelFromValue(shader, gpu_ctx).shade(rgba, d, n, s, i, W, H)
    diff = np.abs(rgba.astype(np.int16) - want.astype(np.int16))
    if shader == "phong":
        assert diff.max() <= 1  # tolerance of the contract: 1 LSB per channel
    else:
        assert diff.max() == 0  # integer shaders: bit-exact
    # synthetic G-buffers: every byte value, counters across the u16 range
    rng = np.random.default_rng(2)
    n2 = 1 << 16
    d2 = rng.integers(0, 256, n2).astype(np.uint8)
    d2[:300] = 255  # Phong background branch (dead for this marcher, still part of the model)
    nb2 = rng.integers(0, 256, 3 * n2).astype(np.uint8)
    s2 = np.arange(n2, dtype=np.uint16)
    i2 = rng.integers(0, 65536, n2).astype(np.uint16)
    want = oracle.shade(shader, d2, nb2, s2, i2, 256, 256)
    rgba = np.zeros(4 * n2, np.uint8)
    rm.createShadingModelFromValue(shader, gpu_ctx).shade(rgba, d2, nb2, s2, i2, 256, 256)
    diff = np.abs(rgba.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= (1 if shader == "phong" else 0)


def test_fused_render_and_shade_on_device_buffers(rm, gpu_ctx, oracle):
    import torch
    W, H = 192, 108
    dev = torch.device("cuda:0")
    sc = rm.Scene("BVH", ctx=gpu_ctx)
    sc.loadPreset(3)
    d = torch.zeros(W * H, dtype=torch.uint8, device=dev)
    nb = torch.zeros(3 * W * H, dtype=torch.uint8, device=dev)
    s = torch.zeros(W * H, dtype=torch.int16, device=dev)
    it = torch.zeros(W * H, dtype=torch.int16, device=dev)
    rg = torch.zeros(4 * W * H, dtype=torch.uint8, device=dev)
    rm.SphereTracer().runRaymarcher(sc, d, nb, s, it, W, H, 0.0, shadedBuffer=rg, shader="iteration-heatmap")
    diag = rm.diagnostics(gpu_ctx, s, it)
    torch.cuda.synchronize()
    want = cpu_render(oracle, 3, "BVH", W, H)
    got = (d.cpu().numpy(), nb.cpu().numpy(), s.cpu().numpy().view(np.uint16), it.cpu().numpy().view(np.uint16))
    assert_same(got, want, "device buffers")
    assert np.array_equal(rg.cpu().numpy(), oracle.shade("iteration-heatmap", *want, W, H))
    od = oracle.diagnostics(want[2], want[3])
    assert all(diag[k] == od[k] for k in od)
    # only RGBA requested (G-buffers NULL)
    rg2 = torch.zeros_like(rg)
    rm.SphereTracer().runRaymarcher(sc, None, None, None, None, W, H, 0.0, shadedBuffer=rg2, shader="iteration-heatmap")
    torch.cuda.synchronize()
    assert torch.equal(rg, rg2)


@pytest.mark.parametrize("world,stripe,accel,weights", [
    (2, 16, "BVH", None), (3, 4, "Octree", None), (8, 16, "BVH", None), (8, 7, "None", None),
    (8, 16, "BVH", [600, 1000, 1000, 1000, 1000, 1000, 1000, 1000]),  # rank 0 (the gather's root) gets a smaller share
    (8, 7, "BVH", [250, 1000, 900, 1000, 1100, 1000, 1000, 1000]), (3, 5, "Octree", [1, 3, 2])])
def test_striped_sharding_reassembles_the_frame(rm, gpu_ctx, oracle, world, stripe, accel, weights):
    """The multi-GPU path on one GPU: every 'rank' renders its interleaved stripes with ONE
    rm_render_stripes_device launch into the packed buffer the gather would move; rank 0's
    reassembly must reproduce the full frame (which must equal the oracle)."""
    import torch
    from cpu_raymarcher_amd import distributed as D
    W, H = 200, 131  # H is not a multiple of stripe * world
    dev = torch.device("cuda:0")
    sc = rm.Scene(accel, ctx=gpu_ctx)
    sc.loadPreset(3)
    sc.camera.setAngles(0.15, -0.4)
    sections = ("rgba", "sdf", "iters", "depth", "normal")
    layout = D.FrameLayout(W, H, world, sections, "interleaved", stripe, weights=weights)
    packed = []
    for rank in range(world):
        buf = torch.zeros(layout.nbytes, dtype=torch.uint8, device=dev)
        D.gpu_render_all(gpu_ctx, sc, W, H, "sdf-heatmap", layout, rank)(buf)
        if weights is None:
            assert sum(b - a for a, b in layout.rows(rank)) == \
                rm._native.lib().rm_stripe_rows(0, H, stripe, world, rank)
        packed.append(buf)
    assert sorted(r for rank in range(world) for r in layout.rows(rank)) == \
        [(a, min(a + stripe, H)) for a in range(0, H, stripe)]  # every stripe dealt exactly once
    if weights is not None:
        share = [sum(b - a for a, b in layout.rows(r)) / H for r in range(world)]
        assert all(abs(sh - w / sum(weights)) <= 1.5 * stripe / H for sh, w in zip(share, weights))
    torch.cuda.synchronize()
    frame = D.new_frame(layout, lambda n: torch.zeros(n, dtype=torch.uint8, device=dev))
    layout.scatter_into_frame(packed, frame)
    want = cpu_render(oracle, 3, accel, W, H, (0.15, -0.4))
    got = (frame["depth"].cpu().numpy(), frame["normal"].cpu().numpy(),
           frame["sdf"].cpu().numpy().view(np.uint16), frame["iters"].cpu().numpy().view(np.uint16))
    assert_same(got, want, "striped world=%d" % world)
    assert np.array_equal(frame["rgba"].cpu().numpy(), oracle.shade("sdf-heatmap", *want, W, H))
    # rank 0's device-side reassembly (what bench.py runs after the gather)
    asm = D.GpuFrameAssembler(layout, dev, 1, ctx=gpu_ctx)
    for rank in range(world):
        asm.recv2d[0][rank].copy_(packed[rank])
    fr = asm.assemble(0)
    for s in sections:
        assert torch.equal(fr[s], frame[s]), s
    # the per-range path (contiguous partition) gives the same bytes
    layout_c = D.FrameLayout(W, H, world, sections, "contiguous", stripe)
    packed_c = []
    for rank in range(world):
        buf = torch.zeros(layout_c.nbytes, dtype=torch.uint8, device=dev)
        rr = D.gpu_render_rows(gpu_ctx, sc, W, H, "sdf-heatmap", layout_c)
        local = 0
        for (a, b) in layout_c.rows(rank):
            rr(a, b, local, buf)
            local += b - a
        packed_c.append(buf)
    frame_c = D.new_frame(layout_c, lambda n: torch.zeros(n, dtype=torch.uint8, device=dev))
    layout_c.scatter_into_frame(packed_c, frame_c)
    asm_c = D.GpuFrameAssembler(layout_c, dev, 1, ctx=gpu_ctx)  # the native assembler serves the reference's partition too
    for rank in range(world):
        asm_c.recv2d[0][rank].copy_(packed_c[rank])
    fr_c = asm_c.assemble(0)
    for s in sections:
        assert torch.equal(frame[s], frame_c[s]), s
        assert torch.equal(fr_c[s], frame[s]), s


@pytest.mark.parametrize("name,stripe,weights", [
    ("C3_dense_4k_bvh_iterheat", 16, None),                                       # C4: the 8-way shard of the C3 frame
    ("C3_dense_4k_bvh_iterheat", 7, [700, 1000, 1000, 1000, 1000, 1000, 1000, 1000]),  # bundle culls across stripe edges
    ("C5_random10k_4k_octree_iterheat", 16, [800, 1000, 1000, 1000, 1000, 1000, 1000, 1000]),
])
def test_c4_c5_eight_way_shard_at_full_size(rm, gpu_ctx, oracle, golden, name, stripe, weights):
    """BASELINE.json C4 (Dense Grid 3840x2160 BVH sharded 8 ways) and the 8-GPU aspect of C5, on ONE GPU: each of the
    eight 'ranks' renders its stripes with one launch (rm_render_stripes_device / rm_render_stripe_list_device) into
    the packed buffer the gather would move, reduces its own counters into the tail of that buffer (as bench.py does),
    and rank 0's native fan-in (rm_assemble_frame_device) rebuilds the frame and combines the partial diagnostics.
    All five buffers must hash to the committed C3 / C5 fixtures and the combined diagnostics equal the fixture's."""
    import torch
    from cpu_raymarcher_amd import distributed as D
    g = golden[name]
    cfg = g["config"]
    W, H, world = cfg["width"], cfg["height"], 8
    dev = torch.device("cuda:0")
    sc = rm.Scene(cfg["accel"], ctx=gpu_ctx)
    if "synthetic" in cfg:
        sp = oracle.synthetic_spheres(cfg["synthetic"])
        sc.loadSpheres(sp[:, :3], sp[:, 3])
    else:
        sc.loadPreset(cfg["preset"])
    sections = ("rgba", "sdf", "iters", "depth", "normal")
    layout = D.FrameLayout(W, H, world, sections, "interleaved", stripe, tail=32, weights=weights)
    asm = D.GpuFrameAssembler(layout, dev, 1, ctx=gpu_ctx)
    for rank in range(world):
        buf = asm.recv2d[0][rank]
        D.gpu_render_all(gpu_ctx, sc, W, H, cfg["shader"], layout, rank)(buf)
        mine = W * sum(b - a for a, b in layout.rows(rank))
        gpu_ctx.reduce_counters_enqueue(layout.section(buf, "sdf").view(torch.int16)[:mine],
                                        layout.section(buf, "iters").view(torch.int16)[:mine],
                                        buf[layout.tail_offset:layout.tail_offset + 32].view(torch.int64))
    acc = torch.zeros(4, dtype=torch.int64, device=dev)
    fr = asm.assemble(0, acc)
    torch.cuda.synchronize()
    for key in ("depth", "normal", "sdf", "iters", "rgba"):
        assert hashlib.sha256(fr[key].cpu().numpy().tobytes()).hexdigest() == g["sha256"][key], (name, key)
    d = gpu_ctx.decode_acc(acc)
    for k in ("total_sdf", "total_iters", "max_sdf", "min_sdf"):
        assert d[k] == g["diagnostics"][k], (name, k, d[k], g["diagnostics"][k])


def test_python_host_validates_buffers(rm, gpu_ctx):
    """the C ABI takes raw pointers, so the host layer must refuse buffers of the wrong element size,
    too short, strided, or on the wrong side (host / device) instead of letting the kernel write out of bounds."""
    import torch
    W, H = 64, 32
    sc = rm.Scene("BVH", ctx=gpu_ctx)
    sc.loadPreset(3)
    ok = lambda: [np.zeros(W * H, np.uint8), np.zeros(3 * W * H, np.uint8), np.zeros(W * H, np.uint16), np.zeros(W * H, np.uint16)]  # noqa: E731
    tr = rm.SphereTracer()
    tr.runRaymarcher(sc, *ok(), W, H, 0.0)
    for k, bad in ((0, np.zeros(W * H - 1, np.uint8)), (1, np.zeros(W * H, np.uint8)), (2, np.zeros(W * H, np.uint8)),
                   (3, np.zeros(W * H, np.uint32)), (0, np.zeros(2 * W * H, np.uint8)[::2]), (2, list(range(W * H)))):
        b = ok()
        b[k] = bad
        with pytest.raises(ValueError):
            tr.runRaymarcher(sc, *b, W, H, 0.0)
    dev = torch.device("cuda:0")
    b = ok()
    b[0] = torch.zeros(W * H, dtype=torch.uint8, device=dev)  # mixed host / device
    with pytest.raises(ValueError):
        tr.runRaymarcher(sc, *b, W, H, 0.0)
    from cpu_raymarcher_amd.host import _job
    job = _job(sc, W, H, 0.0, 0, H, "sphere-tracer")
    small = torch.zeros(10, dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):  # an undersized CUDA tensor would be written out of bounds by store_pixel
        gpu_ctx.render_stripes(job, 4, 2, 0, None, None, None, None, rgba=small, shader=0)
    with pytest.raises(ValueError):
        gpu_ctx.render_stripes(job, 4, 2, 0, None, None, None, None, rgba=np.zeros(4 * W * H, np.uint8), shader=0)
    with pytest.raises(ValueError):
        gpu_ctx.render_stripe_list(job, 4, [0, 2], None, None, None, None, rgba=small, shader=0)
    with pytest.raises(ValueError):
        gpu_ctx.reduce_counters(np.zeros(8, np.uint16), np.zeros(9, np.uint16))
    with pytest.raises(ValueError):
        gpu_ctx.shade(0, W, H, *ok(), np.zeros(4 * W * H - 4, np.uint8))
    with pytest.raises(rm._native.RmError):  # the C ABI itself refuses a list that is not strictly increasing
        gpu_ctx.render_stripe_list(job, 4, [2, 2], None, None, None, None, rgba=torch.zeros(4 * W * 8, dtype=torch.uint8, device=dev))


def test_worker_fan_out_fan_in(rm, gpu_ctx, oracle):
    # main.ts:444-468 with 4 workers on one context; H not divisible by 4
    workers = [rm.RaymarchWorker(ctx=gpu_ctx) for _ in range(4)]
    W, H = 160, 90
    frame = rm.renderFrame(workers, W, H, 3, "Octree", pitch=0.1, yaw=0.2)
    assert_same(frame, cpu_render(oracle, 3, "Octree", W, H, (0.1, 0.2)), "4-worker frame")
    d = rm.diagnostics(gpu_ctx, frame[2], frame[3])
    od = oracle.diagnostics(frame[2], frame[3])
    assert all(d[k] == od[k] for k in od)
    assert d["average_sdf_calls"] == od["total_sdf"] / (W * H)


def test_golden_fixtures_at_baseline_sizes(rm, gpu_ctx, oracle, golden, golden_crops):
    """BASELINE.json configs at full size against the committed fixtures (sha-256 per buffer,
    counter sums, centre crops).  /root/reference is not needed or read."""
    for name, g in sorted(golden.items()):
        cfg = g["config"]
        W, H = cfg["width"], cfg["height"]
        spheres = oracle.synthetic_spheres(cfg["synthetic"]) if "synthetic" in cfg else None
        prims = None
        if "mixed" in cfg:  # the product receives the matrices the oracle's SceneManager restatement made
            prims = oracle.OracleScene(accel="None", prims=oracle.synthetic_mixed_prims(cfg["mixed"])).prims()
        got = gpu_render(rm, gpu_ctx, cfg.get("preset"), cfg["accel"], W, H, (cfg.get("pitch", 0.0), cfg.get("yaw", 0.0)),
                         spheres=spheres, algorithm=cfg.get("algorithm", "sphere-tracer"),
                         overshoot=cfg.get("overshootFactor"), step=cfg.get("stepSize"), prims=prims,
                         time=cfg.get("time", 0.0))
        rgba = np.zeros(4 * W * H, np.uint8)
        rm.createShadingModelFromValue(cfg["shader"], gpu_ctx).shade(rgba, *got, W, H)
        c = g["crop"]
        idx = (np.arange(c["y"], c["y"] + c["size"])[:, None] * W + np.arange(c["x"], c["x"] + c["size"])[None, :]).ravel()
        assert np.array_equal(got[2][idx], golden_crops[name + "/sdf"]), name + " sdf crop"
        assert np.array_equal(got[3][idx], golden_crops[name + "/iters"]), name + " iters crop"
        assert np.array_equal(got[0][idx], golden_crops[name + "/depth"]), name + " depth crop"
        dg = gpu_ctx.reduce_counters(got[2], got[3])
        for k, v in g["diagnostics"].items():
            assert dg[k] == v, (name, k, dg[k], v)
        for key, arr in zip(NAMES, got):
            assert hashlib.sha256(arr.tobytes()).hexdigest() == g["sha256"][key], (name, key)
        if cfg["shader"] == "phong":
            want = golden_crops[name + "/rgba"].astype(np.int16)
            assert np.abs(rgba.reshape(-1, 4)[idx].ravel().astype(np.int16) - want).max() <= 1
        else:
            assert hashlib.sha256(rgba.tobytes()).hexdigest() == g["sha256"]["rgba"], (name, "rgba")


def test_bench_in_flight_configuration_at_4k(rm, golden):
    """bench.py's TIMED configuration -- C3 at 3840x2160, twelve frames in flight on twelve streams, one
    persistent workgroup per CU and launch, 256-pixel items of 8 x 32, no longest-first sort, the tail ramp (2, 3, 4
    workgroups per CU for the last frames), diagnostics fused into the render kernel -- against the golden hashes of all
    five buffers and the fixture's diagnostics, for every buffer set."""
    import torch
    W, H = 3840, 2160
    dev = torch.device("cuda:0")
    ctx = rm.Context(0)
    for k, v in (("blocks_per_cu", 1), ("item_px", 256), ("tile_w", 8), ("lpt", 0)):
        ctx.set_option(k, v)
    scene = rm.Scene("BVH", ctx=ctx)
    scene.loadPreset(3)
    tracer = rm.SphereTracer()
    S, steps = 12, 20
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    sets = [dict(depth=torch.zeros(W * H, dtype=torch.uint8, device=dev), normal=torch.zeros(3 * W * H, dtype=torch.uint8, device=dev),
                 sdf=torch.zeros(W * H, dtype=torch.int16, device=dev), iters=torch.zeros(W * H, dtype=torch.int16, device=dev),
                 rgba=torch.zeros(4 * W * H, dtype=torch.uint8, device=dev), acc=torch.full((4,), -1, dtype=torch.int64, device=dev))
            for _ in range(S)]
    for i in range(steps):
        remaining = steps - 1 - i
        ctx.set_option("blocks_per_cu", max(1, min(4, 4 - remaining)) if remaining < 4 else 1)  # bench.py --tail-ramp 4
        b = sets[i % S]
        with torch.cuda.stream(streams[i % S]):
            tracer.runRaymarcher(scene, b["depth"], b["normal"], b["sdf"], b["iters"], W, H, 0.0, shadedBuffer=b["rgba"],
                                 shader="iteration-heatmap", diagnostics=b["acc"])
    torch.cuda.synchronize()
    assert "render_kernel_v2<2, true, true, false>" in ctx.last_kernel()
    g = golden["C3_dense_4k_bvh_iterheat"]
    for k, b in enumerate(sets):
        for name in ("depth", "normal", "sdf", "iters", "rgba"):
            assert hashlib.sha256(b[name].cpu().numpy().tobytes()).hexdigest() == g["sha256"][name], (k, name)
        d = ctx.decode_acc(b["acc"])
        assert all(d[key] == g["diagnostics"][key] for key in ("total_sdf", "total_iters", "max_sdf", "min_sdf")), (k, d)
    ctx.close()


def test_full_size_properties(rm, gpu_ctx):
    """Size-independent properties at 3840x2160: tile split == whole frame; iterations <= 100;
    a pixel with no iterations has no SDF calls; diagnostics equal a host recount."""
    W, H = 3840, 2160
    whole = gpu_render(rm, gpu_ctx, 3, "BVH", W, H)
    a = gpu_render(rm, gpu_ctx, 3, "BVH", W, H, rows=(0, 1000))
    b = gpu_render(rm, gpu_ctx, 3, "BVH", W, H, rows=(1000, 2160))
    for k, name in enumerate(NAMES):
        assert np.array_equal(whole[k], np.concatenate([a[k], b[k]])), name
    assert whole[3].max() <= 100
    assert np.all(whole[2][whole[3] == 0] == 0)
    assert np.all(whole[1].reshape(-1, 3)[whole[3] == 0] == 128)
    d = gpu_ctx.reduce_counters(whole[2], whole[3])
    assert d["total_sdf"] == int(whole[2].astype(np.int64).sum()) and d["total_iters"] == int(whole[3].astype(np.int64).sum())
    assert d["max_sdf"] == int(whole[2].max()) and d["min_sdf"] == int(whole[2].min())


def test_analytics_sweep_frames(rm, gpu_ctx, oracle):
    """the Analytics view rotates the camera by 0.015 rad of yaw per frame
    (main.ts:438-441); every frame of the sweep must match the oracle (host Math.sin/cos path)."""
    W, H = 160, 90
    sc = rm.Scene("BVH", ctx=gpu_ctx)
    sc.loadPreset(3)
    osc = oracle.OracleScene(preset=3, accel="BVH")
    yaw = 0.0
    for frame in range(12):
        sc.camera.rotateCamera(0, 0.015)
        yaw += 0.015
        n = W * H
        bufs = (np.zeros(n, np.uint8), np.zeros(3 * n, np.uint8), np.zeros(n, np.uint16), np.zeros(n, np.uint16))
        rm.SphereTracer().runRaymarcher(sc, *bufs, W, H, float(frame))
        osc.set_angles(0.0, yaw)
        assert sc.camera.yaw == yaw
        assert_same(bufs, osc.render(W, H), "sweep frame %d" % frame)


@pytest.mark.parametrize("accel", ["None", "BVH", "Octree"])
@pytest.mark.parametrize("preset", [5, 7, 8, 9])
def test_box_and_torus_presets(rm, gpu_ctx, oracle, specialise, preset, accel):
    """Torus (rotated), Cube, Sphere and Cube, Pyramid of Boxes -- general
    transformMat4, Box / Torus localSdf, bounds from the inverted matrix."""
    for alg, ang in (("sphere-tracer", (0.3, -0.8)), ("adaptive-step-v3", (-0.5, 2.4))):
        got = gpu_render(rm, gpu_ctx, preset, accel, 220, 140, ang, algorithm=alg)
        assert gpu_ctx.last_kernel().startswith("rm_rtc_render<" if specialise else "render_kernel<"), gpu_ctx.last_kernel()
        assert_same(got, cpu_render(oracle, preset, accel, 220, 140, ang, algorithm=alg), "preset %d %s %s" % (preset, accel, alg))


def test_mixed_rotated_primitives_and_make_transform(rm, gpu_ctx, oracle):
    desc = oracle.synthetic_mixed_prims(60, seed=11)
    osc = oracle.OracleScene(accel="None", prims=desc)
    triples = osc.prims()
    # rm_make_transform == SceneManager.getTransform as the oracle restates it
    for d, (t, m, par) in zip(desc, triples):
        mine = rm.make_transform(*d["pos"], rotation=d["rot"])
        assert np.array_equal(mine, m), d
    for accel in ("None", "BVH", "Octree"):
        got = gpu_render(rm, gpu_ctx, None, accel, 260, 150, (0.1, 0.9), prims=triples)
        assert_same(got, cpu_render(oracle, None, accel, 260, 150, (0.1, 0.9), prims=desc), "mixed60 " + accel)
    sc = rm.Scene("BVH", ctx=gpu_ctx)
    sc.loadPrims(triples)
    pts = np.random.default_rng(3).uniform(-2, 2, (500, 3)).astype(np.float32)
    d, c = sc.getDistances(pts)
    ob = oracle.OracleScene(accel="BVH", prims=desc)
    for k in range(0, 500, 5):
        assert (d[k], c[k]) == ob.distance(pts[k])


def test_torus_radii_of_either_sign_keep_the_primitive_filter_exact(rm, oracle):
    """the bounding sphere of the general-primitive filter used |major + minor| for a torus, which is not a
    geometric bound when a radius is negative (legal input: torus.ts:14-25 just subtracts them).  Tori with radii of either
    sign, filtered against unfiltered (the unfiltered loop is the reference's, oracle-checked by the tests above)."""
    ctx = rm.Context(0)
    desc = oracle.synthetic_mixed_prims(45, seed=5)
    triples = [list(t) for t in oracle.OracleScene(accel="None", prims=desc).prims()]
    k = 0
    for t in triples:
        if t[0] == 2:  # torus: (major, minor)
            t[2] = [(1.0, -0.5), (-1.0, 2.0), (-0.3, -0.2), (0.4, 0.1)][k % 4] + (0.0,)
            k += 1
    assert k >= 8
    for accel in ("None", "BVH"):
        outs = []
        for filt in (1, 0):
            ctx.set_option("filter", filt)
            outs.append(gpu_render(rm, ctx, None, accel, 220, 130, (0.2, 0.6), prims=[tuple(t) for t in triples]))
        assert_same(outs[0], outs[1], "tori with radii of either sign, filter on / off, " + accel)
    ctx.close()


# ---- SDF operators and the Mandelbulb ------------------------------------------

def _jsmath_cases(rng, n):
    trig = np.concatenate([rng.uniform(-40, 40, n // 2), rng.uniform(-8e5, 8e5, n // 8), rng.normal(0, 1e-3, n // 8),
                           (np.arange(n // 8) - n // 16) * np.pi / 2 * (1 + rng.normal(0, 1e-9, n // 8)),
                           # beyond 2^19*pi/2: the Payne-Hanek reduction of k_rem_pio2.c
                           rng.uniform(8e5, 1e7, n // 32), 10.0 ** rng.uniform(6, 300, n // 16) * rng.choice([-1, 1], n // 16),
                           2.0 ** rng.integers(20, 1023, n // 32).astype(float)])
    y = rng.normal(0, 1, n) * 10.0 ** rng.integers(-5, 5, n)
    x = rng.normal(0, 1, n) * 10.0 ** rng.integers(-5, 5, n)
    unit = np.concatenate([rng.uniform(-1.01, 1.01, n - 1000), 1 - 10.0 ** rng.uniform(-16, 0, 1000)])
    pos = np.concatenate([rng.uniform(-0.1, 4, n // 2), 10.0 ** rng.uniform(-320, 300, n // 2)])
    rnd = np.concatenate([rng.uniform(-100, 100, n - 9), [0.5, -0.5, 1.5, -1.5, 2.5, -2.5, -0.0, 0.49999999999999994, -0.2]])
    pw_x = np.concatenate([rng.uniform(0, 2.5, n // 2), 10.0 ** rng.uniform(-20, 20, n // 4), rng.uniform(-5, 5, n // 4)])
    pw_y = np.concatenate([rng.choice([7.0, 8.0, 2.0, 0.5, 3.0, -1.0], n // 4), rng.uniform(-10, 10, n // 4),
                           rng.uniform(-30, 30, n // 4), rng.integers(-9, 9, n // 4).astype(float)])
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 5e-324, 1e308, 2.0 ** -1022])
    sa, sb = [v.ravel() for v in np.meshgrid(special, special)]
    return {0: (np.r_[trig, special], None), 1: (np.r_[trig, special], None), 2: (np.r_[y, sa], np.r_[x, sb]),
            3: (np.r_[unit, special], None), 4: (np.r_[pos, special], None), 5: (np.r_[pw_x, sa], np.r_[pw_y, sb]),
            6: (np.r_[rnd, special], None), 7: (np.r_[y, special], None)}


def test_device_jsmath_is_bit_identical_to_the_oracle(gpu_ctx, oracle):
    """csrc/rm_jsmath.h (device) == oracle/ro_jsmath.h (which is pinned against node's Math.*)."""
    import ctypes
    L = oracle.lib()
    L.ro_jsmath_eval.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_long]
    for fn, (a, b) in _jsmath_cases(np.random.default_rng(21), 400000).items():
        a = np.ascontiguousarray(a)
        bb = np.zeros_like(a) if b is None else np.ascontiguousarray(b)
        want = np.zeros_like(a)
        L.ro_jsmath_eval(fn, a.ctypes.data, bb.ctypes.data, want.ctypes.data, len(a))
        got = gpu_ctx.selftest_jsmath(fn, a, b)
        same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (fn, int((~same).sum()), a[~same][:3], got[~same][:3], want[~same][:3])


def _assert_program_kernel(gpu_ctx, specialise, what):
    k = gpu_ctx.last_kernel()
    if specialise:
        assert k.startswith("rm_rtc_render<"), (what, k, gpu_ctx.rtc_status())
    else:
        assert k.startswith("render_kernel<") and k.split(">")[0].endswith((", 2", ", 3")), (what, k)


@pytest.mark.parametrize("preset", [6, 10, 11, 12, 13, 14, 15, 16, 17, 18])
def test_operator_and_mandelbulb_presets(rm, gpu_ctx, oracle, specialise, preset):
    W, H = (120, 80) if preset == 13 else (200, 120)
    for accel in ("None", "BVH", "Octree"):
        for alg, tm, ang in (("sphere-tracer", 0.0, (0.0, 0.0)), ("adaptive-step-v3", 1234.5, (0.3, 0.8)),
                             ("fixed-step", 77.0, (-0.4, 2.0))):
            if preset == 13 and alg == "fixed-step" and accel != "None":
                continue  # the Mandelbulb costs 80 escape iterations per evaluation on the oracle side
            got = gpu_render(rm, gpu_ctx, preset, accel, W, H, ang, algorithm=alg, time=tm)
            _assert_program_kernel(gpu_ctx, specialise, (preset, accel, alg))
            want = cpu_render(oracle, preset, accel, W, H, ang, algorithm=alg, time=tm)
            assert_same(got, want, "preset %d %s %s t=%g" % (preset, accel, alg, tm))
    done, failed, log = gpu_ctx.rtc_status()
    assert failed == 0, log


def _random_forest(rng, n_roots):
    """Nested dicts in the oracle's prims format: random operator trees over random leaves."""
    def leaf():
        kind = rng.choice(["sphere", "box", "torus"])
        d = {"type": str(kind), "pos": [float(np.float32(v)) for v in rng.uniform(-1.2, 1.2, 3)],
             "rot": [float(np.float32(v)) for v in rng.uniform(-3, 3, 3)] if rng.random() < 0.5 else None}
        if kind == "sphere":
            d["r"] = float(rng.uniform(0.1, 0.4))
        elif kind == "box":
            d["half"] = [float(np.float32(v)) for v in rng.uniform(0.05, 0.35, 3)]
        else:
            d["radius"] = float(rng.uniform(0.15, 0.4))
        return d

    def tree(depth):
        if depth == 0 or rng.random() < 0.25:
            return leaf()
        op = rng.choice(["round", "smoothUnion", "smoothSub", "twist", "anim", "repetition"],
                        p=[0.25, 0.3, 0.15, 0.15, 0.1, 0.05])
        if op == "round":
            return {"type": "round", "a": tree(depth - 1), "radius": float(rng.uniform(0.01, 0.15))}
        if op == "twist":
            return {"type": "twist", "a": tree(depth - 1), "amount": float(rng.uniform(0.5, 4))}
        if op == "anim":
            return {"type": "anim", "a": tree(depth - 1), "direction": [float(v) for v in rng.uniform(-1, 1, 3)],
                    "amplitude": float(rng.uniform(0.1, 0.6)), "speed": float(rng.uniform(0.001, 0.01))}
        if op == "repetition":
            return {"type": "repetition", "a": tree(depth - 1), "spacing": [float(np.float32(v)) for v in rng.uniform(2.5, 4, 3)]}
        return {"type": str(op), "a": tree(depth - 1), "b": tree(depth - 1), "k": float(rng.uniform(0.01, 0.3))}

    return [tree(4) for _ in range(n_roots)]


def test_random_expression_forests_through_rm_scene_from_nodes(rm, gpu_ctx, oracle, specialise):
    rng = np.random.default_rng(5)
    for trial, n_roots in enumerate((1, 3, 7)):
        forest = _random_forest(rng, n_roots)
        for accel in ("None", "BVH", "Octree"):
            osc = oracle.OracleScene(accel=accel, prims=forest)
            osc.set_angles(0.2, -0.7)
            want = osc.render(160, 100, time=500.0)
            got = gpu_render(rm, gpu_ctx, None, accel, 160, 100, (0.2, -0.7), nodes=osc.nodes(), time=500.0)
            _assert_program_kernel(gpu_ctx, specialise, (trial, accel))
            assert_same(got, want, "forest %d %s" % (trial, accel))
        sc = rm.Scene("BVH", ctx=gpu_ctx)
        osc = oracle.OracleScene(accel="BVH", prims=forest)
        sc.loadNodes(*osc.nodes())
        sc.updateTime(250.0)
        pts = rng.uniform(-2, 2, (300, 3)).astype(np.float32)
        d, c = sc.getDistances(pts)
        for k in range(0, 300, 3):
            wd, wc = osc.distance(pts[k], time=250.0)
            assert (d[k] == wd or (np.isnan(d[k]) and np.isnan(wd))) and c[k] == wc, (trial, k, d[k], wd)


def test_scene_kernels_are_cached_across_scene_switches(rm, gpu_ctx):
    """A preset menu goes back and forth: the second visit of a scene must not compile again (rm_rtc's process-wide cache)."""
    import time as _t
    gpu_ctx.set_option("specialise", 1)

    def visit(preset):
        t0 = _t.perf_counter()
        got = gpu_render(rm, gpu_ctx, preset, "BVH", 64, 48)
        assert gpu_ctx.last_kernel().startswith("rm_rtc_render<")
        return _t.perf_counter() - t0, got

    visit(17)
    visit(6)
    again, a = visit(17)
    _, b = visit(17)
    assert again < 0.5, "preset 17 took %.2f s on its second visit" % again
    assert all((x == y).all() for x, y in zip(a, b))


def test_v2_wave_loop_compiled_for_its_configuration(rm, oracle):
    """The v2 kernel compiled at run time with a launch configuration's parameters as literals (rm_v2_fields.h): same bytes as
    the library's own instantiation and as the oracle; the camera and the rows of a launch are not part of the configuration
    (no further compile), the frame size and the shader are."""
    ctx = rm.Context(0)
    tag = "[launch constants compiled in]"

    def render(preset, accel, W, H, ang, rows=None, spheres=None):
        return gpu_render(rm, ctx, preset, accel, W, H, ang, rows=rows, spheres=spheres)

    try:
        ctx.set_option("specialise_v2_after", 1)
        compiled0 = ctx.rtc_status()[0]
        for ang in ((0.0, 0.0), (0.3, 0.7), (-0.4, 2.1)):  # one configuration, three cameras
            got = render(3, "BVH", 320, 180, ang)
            assert tag in ctx.last_kernel(), (ctx.last_kernel(), ctx.rtc_status())
            assert_same(got, cpu_render(oracle, 3, "BVH", 320, 180, ang), "compiled v2, camera %s" % (ang,))
        assert ctx.rtc_status()[0] == compiled0 + 1, ctx.rtc_status()
        got = render(3, "BVH", 320, 180, (0.3, 0.7), rows=(40, 100))  # a band of rows: the same configuration
        assert tag in ctx.last_kernel() and ctx.rtc_status()[0] == compiled0 + 1
        assert_same(got, cpu_render(oracle, 3, "BVH", 320, 180, (0.3, 0.7), rows=(40, 100)), "compiled v2, a band of rows")
        got = render(3, "None", 200, 120, (0.2, 0.5))  # another configuration (no acceleration, another frame size)
        assert tag in ctx.last_kernel() and ctx.rtc_status()[:2] == (1, 0), ctx.rtc_status()  # (a new scene build starts the context's list afresh)
        assert_same(got, cpu_render(oracle, 3, "None", 200, 120, (0.2, 0.5)), "compiled v2, no acceleration")
        # a small tree (20 spheres: the origin-relative instantiation, whose loops unroll over literal counts): compiled with or
        # without the counts or refused -- whichever, the bytes are the oracle's
        rng = np.random.default_rng(21)
        sp = np.concatenate([rng.uniform(-1, 1, (20, 3)), rng.uniform(0.1, 0.3, (20, 1))], axis=1).astype(np.float32)
        got = render(None, "BVH", 240, 160, (0.1, 0.3), spheres=sp)
        assert "render_kernel_v2<" in ctx.last_kernel()
        assert_same(got, cpu_render(oracle, None, "BVH", 240, 160, (0.1, 0.3), spheres=sp), "compiled v2, small tree")
        assert ctx.rtc_status()[1] == 0 or "refused" in ctx.rtc_status()[2], ctx.rtc_status()
        ctx.set_option("specialise_v2_after", 0)
        got0 = render(3, "BVH", 320, 180, (0.3, 0.7))
        assert tag not in ctx.last_kernel()
        ctx.set_option("specialise_v2_after", 1)
        assert_same(render(3, "BVH", 320, 180, (0.3, 0.7)), got0, "compiled v2 against the library's instantiation")
    finally:
        ctx.close()


def test_background_compile_never_waits_and_takes_over(rm, gpu_ctx, oracle):
    """`specialise` = 2: the first renders of a new scene run in the ahead-of-time kernel while a background thread compiles; the
    scene's own kernel takes over when it is ready; every frame equals the oracle's."""
    import time as _t
    rng = np.random.default_rng(77)  # a forest no other test builds: not in the process-wide cache
    forest = _plain_forest(rng, 2, 3, (3e-4, 7e-3))
    osc = oracle.OracleScene(accel="BVH", prims=forest)
    osc.set_angles(0.1, 0.4)
    want = osc.render(160, 100)
    gpu_ctx.set_option("specialise", 2)
    try:
        t0 = _t.perf_counter()
        got = gpu_render(rm, gpu_ctx, None, "BVH", 160, 100, (0.1, 0.4), nodes=osc.nodes())
        first = _t.perf_counter() - t0
        assert gpu_ctx.last_kernel().startswith("render_kernel<"), gpu_ctx.last_kernel()
        assert_same(got, want, "while the scene's kernel compiles")
        assert first < 1.0, "the first render waited %.2f s" % first
        sc = rm.Scene("BVH", ctx=gpu_ctx)
        sc.loadNodes(*osc.nodes())
        sc.camera.setAngles(0.1, 0.4)
        deadline = _t.perf_counter() + 60
        while _t.perf_counter() < deadline:
            bufs = (np.zeros(16000, np.uint8), np.zeros(48000, np.uint8), np.zeros(16000, np.uint16), np.zeros(16000, np.uint16))
            rm.createRaymarcher("sphere-tracer", None, None).runRaymarcher(sc, *bufs, 160, 100, 0.0, 0, 100)
            if gpu_ctx.last_kernel().startswith("rm_rtc_render<"):
                break
            _t.sleep(0.2)
        assert gpu_ctx.last_kernel().startswith("rm_rtc_render<"), gpu_ctx.rtc_status()
        assert_same(bufs, want, "after the take-over")
    finally:
        gpu_ctx.set_option("specialise", 1)


def _plain_forest(rng, n_roots, depth, k_range):
    """Spheres, boxes and tori (half of them rotated) under Round / SmoothUnion / SmoothSubtraction only: the trees whose
    specialised code prunes operands by binary32 intervals (csrc/rm_rtc.cpp)."""
    def leaf():
        kind = rng.choice(["sphere", "box", "torus"], p=[0.3, 0.5, 0.2])
        d = {"type": str(kind), "pos": [float(np.float32(v)) for v in rng.uniform(-0.9, 0.9, 3)],
             "rot": [float(np.float32(v)) for v in rng.uniform(-3, 3, 3)] if rng.random() < 0.5 else None}
        if kind == "sphere":
            d["r"] = float(rng.uniform(0.1, 0.4))
        elif kind == "box":
            d["half"] = [float(np.float32(v)) for v in rng.uniform(0.02, 0.4, 3)]
        else:
            d["radius"] = float(rng.uniform(0.15, 0.4))
        return d

    def tree(dep):
        if dep == 0 or (dep < depth and rng.random() < 0.15):
            return leaf()
        op = rng.choice(["round", "smoothUnion", "smoothSub"], p=[0.2, 0.6, 0.2]) if dep < depth else "smoothUnion"
        if op == "round":
            return {"type": "round", "a": tree(dep - 1), "radius": float(rng.uniform(0.005, 0.1))}
        return {"type": str(op), "a": tree(dep - 1), "b": tree(dep - 1), "k": float(np.exp(rng.uniform(*np.log(k_range))))}

    return [tree(depth) for _ in range(n_roots)]


@pytest.mark.parametrize("seed,n_roots,depth,k_range", [(11, 1, 4, (2e-5, 2e-3)), (12, 2, 3, (1e-3, 0.05)), (13, 1, 5, (1e-4, 0.3)), (14, 3, 3, (1e-5, 1e-4))])
def test_pruned_smooth_trees_equal_the_oracle(rm, gpu_ctx, oracle, seed, n_roots, depth, k_range):
    """Exact pruning (rm_rtc.cpp): narrow and wide blends, unions and subtractions, rotated leaves -- renders and point
    queries (surface points and blend seams among them) equal the oracle's, and equal the unpruned specialised code."""
    rng = np.random.default_rng(seed)
    forest = _plain_forest(rng, n_roots, depth, k_range)
    osc = oracle.OracleScene(accel="BVH", prims=forest)
    gpu_ctx.set_option("specialise", 1)
    sc = rm.Scene("BVH", ctx=gpu_ctx)
    sc.loadNodes(*osc.nodes())
    assert "bounded subtree" in gpu_ctx.rtc_source(), "the forest was meant to exercise the pruning code"
    for accel in ("BVH", "None"):
        o2 = oracle.OracleScene(accel=accel, prims=forest)
        o2.set_angles(0.25, 0.9)
        want = o2.render(200, 120)
        got = gpu_render(rm, gpu_ctx, None, accel, 200, 120, (0.25, 0.9), nodes=o2.nodes())
        assert gpu_ctx.last_kernel().startswith("rm_rtc_render<")
        assert_same(got, want, "plain forest %d %s" % (seed, accel))
    # point queries: random points, then points walked onto the surface (where neighbouring leaves are closest to each other)
    sc = rm.Scene("BVH", ctx=gpu_ctx)
    sc.loadNodes(*osc.nodes())
    pts = rng.uniform(-1.5, 1.5, (4000, 3)).astype(np.float32)
    d, c = sc.getDistances(pts)
    for _ in range(6):  # crude sphere tracing along a random direction towards the surface
        dirs = rng.normal(size=pts.shape)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        pts = (pts + dirs * np.clip(d, -1, 1)[:, None] * 0.9).astype(np.float32)
        d, c = sc.getDistances(pts)
    gpu_ctx.set_option("prune", 0)
    try:
        sc2 = rm.Scene("BVH", ctx=gpu_ctx)
        sc2.loadNodes(*osc.nodes())
        assert "bounded subtree" not in gpu_ctx.rtc_source()
        d2, c2 = sc2.getDistances(pts)
    finally:
        gpu_ctx.set_option("prune", 1)
    assert (d.view(np.uint64) == d2.view(np.uint64)).all() and (c == c2).all(), int((d.view(np.uint64) != d2.view(np.uint64)).sum())
    for k in range(0, len(pts), 40):
        wd, wc = osc.distance(pts[k])
        assert d[k] == wd and c[k] == wc, (seed, k, d[k], wd)


@pytest.fixture
def sqrt_length(oracle, gpu_ctx):
    """vec3.length = Math.sqrt(x*x + y*y + z*z) on both sides, restored afterwards (the oracle's switch is global)."""
    oracle.lib().ro_set_length_mode(1)
    gpu_ctx.set_option("length", 1)
    yield
    gpu_ctx.set_option("length", 0)
    oracle.lib().ro_set_length_mode(0)


@pytest.mark.parametrize("case", [
    dict(preset=3, accel="BVH", W=3840, H=2160, rows=(1040, 1120)),     # C3 band through the grid's centre
    dict(preset=3, accel="BVH", W=640, H=360, ang=(0.3, 0.7)),            # whole frame, rotated camera
    dict(synthetic=10000, accel="Octree", W=3840, H=2160, rows=(1060, 1100)),  # C5 band
    dict(preset=3, accel="None", W=320, H=180),
    dict(preset=9, accel="BVH", W=320, H=200, ang=(0.3, -0.8)),           # boxes: box.ts:26,33 use vec3.length too
    dict(preset=17, accel="Octree", W=240, H=160, ang=(0.2, 0.5)),        # smooth unions: vec3.distance in the bounds
    dict(preset=13, accel="BVH", W=96, H=64, time=1000.0),                # Mandelbulb: r = vec3.length(z)
    dict(preset=2, accel="BVH", W=320, H=180, algorithm="adaptive-step-v3"),
])
def test_vec3_length_sqrt_mode(rm, gpu_ctx, oracle, sqrt_length, case):
    """SURVEY Appendix B / VERDICT r1 #5: gl-matrix 3.4.4 is not available offline and may compute vec3.length as
    Math.sqrt(x*x + y*y + z*z) instead of Math.hypot; the product carries the switch (option `length`), and in that
    mode it must equal the oracle in that mode -- v2, v1, general primitives and the interpreter."""
    W, H = case["W"], case["H"]
    sp = oracle.synthetic_spheres(case["synthetic"]) if "synthetic" in case else None
    kw = dict(rows=case.get("rows"), spheres=sp, algorithm=case.get("algorithm", "sphere-tracer"), time=case.get("time", 0.0))
    got = gpu_render(rm, gpu_ctx, case.get("preset"), case["accel"], W, H, case.get("ang", (0.0, 0.0)), **kw)
    want = cpu_render(oracle, case.get("preset"), case["accel"], W, H, case.get("ang", (0.0, 0.0)), **kw)
    assert_same(got, want, "length=sqrt %s" % case)
    assert "[length=sqrt]" in gpu_ctx.last_kernel()


def test_vec3_length_modes_differ_and_switch_back(rm, gpu_ctx, oracle):
    """The two formulas are different functions (a point where they differ in the last bit of the distance), the
    switch really changes what the device computes, and switching back restores the default results."""
    rng = np.random.default_rng(3)
    pts = (rng.standard_normal((20000, 3)) * 1.2).astype(np.float32)
    sc = rm.Scene("None", ctx=gpu_ctx)
    sc.loadPreset(3)
    d_hypot, _ = sc.getDistances(pts)
    gpu_ctx.set_option("length", 1)
    try:
        assert gpu_ctx.get_option("length") == 1
        d_sqrt, _ = sc.getDistances(pts)
    finally:
        gpu_ctx.set_option("length", 0)
    d_again, _ = sc.getDistances(pts)
    assert np.array_equal(d_hypot, d_again)
    diff = d_hypot != d_sqrt
    assert 0 < diff.sum() < len(pts)  # they differ on a fraction of the points ...
    assert np.abs(d_hypot - d_sqrt).max() < 1e-15  # ... by an ulp of the distance
    L = oracle.lib()
    osc = oracle.OracleScene(preset=3, accel="None")
    L.ro_set_length_mode(1)
    try:
        want = np.array([osc.distance(p)[0] for p in pts[:3000]])
    finally:
        L.ro_set_length_mode(0)
    assert np.array_equal(d_sqrt[:3000], want)



def test_longest_first_item_order_is_only_an_order(rm, oracle):
    """Option `lpt`: a launch hands out its work items longest-first from the costs the PREVIOUS launch recorded.  Any
    permutation must give the same bytes: the same frame repeated (feedback from an identical frame), a moving camera
    (stale costs), a change of frame size and of row range (feedback restarts), row stripes, and frames in flight on
    several streams (a launch sorts while the previous one is still writing its costs)."""
    import torch
    ctx = rm.Context(0)
    ctx.set_option("kernel", 2)
    dev = torch.device("cuda:0")

    def render(W, H, ang, rows=None, lpt=1):
        ctx.set_option("lpt", lpt)
        return gpu_render(rm, ctx, 3, "BVH", W, H, ang, rows=rows)

    ref = {}
    for lpt in (0, 1, 1, 1):  # the third and fourth launches use the costs of an identical frame
        got = render(640, 360, (0.1, 0.3), lpt=lpt)
        ref.setdefault("a", got)
        assert_same(got, ref["a"], "repeated frame, lpt=%d" % lpt)
    assert_same(ref["a"], cpu_render(oracle, 3, "BVH", 640, 360, (0.1, 0.3)), "vs oracle")
    for k in range(6):  # moving camera: costs one frame stale
        ang = (0.1 + 0.05 * k, 0.3 + 0.2 * k)
        assert_same(render(640, 360, ang, lpt=1), render(640, 360, ang, lpt=0), "moving camera %d" % k)
    for (W, H, rows) in ((333, 77, None), (640, 360, (100, 300)), (640, 360, (0, 17)), (1920, 1080, None), (640, 360, None)):
        assert_same(render(W, H, (0.2, 0.5), rows, lpt=1), render(W, H, (0.2, 0.5), rows, lpt=0), "geometry change %dx%d %r" % (W, H, rows))
    # frames in flight: 6 streams x 4 rounds, alternating two cameras; every buffer set must hold its own frame
    ctx.set_option("lpt", 1)
    W, H = 1280, 720
    scene = rm.Scene("BVH", ctx=ctx)
    scene.loadPreset(3)
    tracer = rm.SphereTracer()
    streams = [torch.cuda.Stream(device=dev) for _ in range(6)]
    sets = [dict(d=torch.zeros(W * H, dtype=torch.uint8, device=dev), n=torch.zeros(3 * W * H, dtype=torch.uint8, device=dev),
                 s=torch.zeros(W * H, dtype=torch.int16, device=dev), i=torch.zeros(W * H, dtype=torch.int16, device=dev)) for _ in range(6)]
    for rnd in range(4):
        for k in range(6):
            scene.camera.setAngles(0.1 * (k % 2), 0.4 * (k % 2))
            with torch.cuda.stream(streams[k]):
                tracer.runRaymarcher(scene, sets[k]["d"], sets[k]["n"], sets[k]["s"], sets[k]["i"], W, H, 0.0)
    torch.cuda.synchronize()
    for k in range(2, 6):
        for name in sets[k]:
            assert torch.equal(sets[k][name], sets[k % 2][name]), (k, name)
    ctx.set_option("lpt", 0)
    for k in range(2):
        scene.camera.setAngles(0.1 * k, 0.4 * k)
        b = {n: torch.zeros_like(v) for n, v in sets[k].items()}
        tracer.runRaymarcher(scene, b["d"], b["n"], b["s"], b["i"], W, H, 0.0)
        torch.cuda.synchronize()
        for name in b:
            assert torch.equal(b[name], sets[k][name]), (k, name)
    ctx.close()
