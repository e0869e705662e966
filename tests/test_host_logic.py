"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol
include/rm_raymarch.h declares, and the host logic (presets, camera, BVH/Octree build,
partition, string defaulting, error codes) agrees with the oracle.  No compute calls."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(rm):
    from cpu_raymarcher_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "rm_raymarch.h")).read()
    declared = set(re.findall(r"RM_API\s+[\w\s\*]+?\b(rm_\w+)\s*\(", hdr))
    assert len(declared) >= 20
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    lib = C.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in N.lib().rm_version()


def test_struct_layouts_match_header(rm):
    from cpu_raymarcher_amd import _native as N
    assert C.sizeof(N.rm_job) == 72 and N.rm_job.camera_pitch.offset == 24 and N.rm_job.step_size.offset == 64
    assert C.sizeof(N.rm_scene_info) == 72
    assert C.sizeof(N.rm_diagnostics) == 32
    assert C.sizeof(N.rm_node) == 128 and N.rm_node.params.offset == 80 and C.sizeof(N.rm_prim) == 104


def test_string_defaulting_rules(rm):
    from cpu_raymarcher_amd import _native as N
    L = N.lib()
    # raymarchWorker.ts:50-68: unknown -> sphere tracer
    assert [L.rm_algorithm_from_string(s.encode()) for s in
            ("sphere-tracer", "fixed-step", "adaptive-step", "adaptive-step-v2", "adaptive-step-v3", "bogus", "")] == \
        [0, 1, 2, 3, 4, 0, 0]
    # scene.ts:32-36: anything else -> None (case-sensitive)
    assert [L.rm_accel_from_string(s.encode()) for s in ("None", "Octree", "BVH", "bvh", "")] == [0, 1, 2, 0, 0]
    # main.ts:33-45
    assert [L.rm_shader_from_string(s.encode()) for s in ("normal", "phong", "sdf-heatmap", "iteration-heatmap", "x")] \
        == [0, 1, 2, 3, 0]
    assert L.rm_preset_count() == 19


def test_partition_rows_matches_main_ts(rm):
    # main.ts:444-449: r = ceil(H / N); [min(i r, H), min((i+1) r, H))
    for H in (1, 7, 128, 1080, 2160):
        for n in (1, 2, 3, 4, 7, 8, 16):
            r = math.ceil(H / n)
            rows = [rm.partition_rows(H, n, i) for i in range(n)]
            assert rows == [(min(i * r, H), min((i + 1) * r, H)) for i in range(n)]
            assert sum(b - a for a, b in rows) == H


def test_host_only_context_builds_scenes_like_the_oracle(rm, oracle):
    ctx = rm.Context(None)
    for preset in range(19):  # 6 and 10-18: operator trees / Mandelbulb (bounds from the overridden getters)
        for accel, name in ((0, "None"), (1, "Octree"), (2, "BVH")):
            ctx.scene_from_preset(preset, accel)
            info = ctx.scene_info()
            osc = oracle.OracleScene(preset=preset, accel=name)
            st = osc.stats()
            assert info["n_prims"] == st["n"]
            assert (info["bvh_nodes"], info["bvh_leaves"], info["bvh_depth"]) == \
                (st["bvh_nodes"], st["bvh_leaves"], st["bvh_depth"])
            assert (info["oct_nodes"], info["oct_leaves"], info["oct_empty_leaves"], info["oct_max_leaf_prims"]) == \
                (st["oct_nodes"], st["oct_leaves"], st["oct_empty"], st["oct_maxleafprims"])
            rb = osc.root_bounds()
            if rb is not None:
                assert np.array_equal(np.float32(info["root_min"] + info["root_max"]), rb)
    sp = oracle.synthetic_spheres(3000)
    for accel, name in ((1, "Octree"), (2, "BVH")):
        ctx.scene_from_spheres(sp[:, :3], sp[:, 3], accel)
        info = ctx.scene_info()
        st = oracle.OracleScene(spheres=sp, accel=name).stats()
        assert (info["bvh_nodes"], info["bvh_leaves"], info["bvh_depth"], info["oct_nodes"], info["oct_leaves"],
                info["oct_empty_leaves"], info["oct_max_leaf_prims"]) == \
            (st["bvh_nodes"], st["bvh_leaves"], st["bvh_depth"], st["oct_nodes"], st["oct_leaves"], st["oct_empty"],
             st["oct_maxleafprims"])


def test_camera_matches_oracle(rm, oracle):
    sc = oracle.OracleScene(preset=0, accel="None")
    rng = np.random.default_rng(3)
    for pitch, yaw in [(0, 0), (0.3, 0.7), (-2.0, 9.0), (math.pi / 2, -math.pi)] + list(rng.uniform(-4, 4, (200, 2))):
        sc.set_angles(pitch, yaw)
        rot, org = sc.camera()
        r2, o2 = rm.camera_from_angles(pitch, yaw)
        assert np.array_equal(rot, r2) and np.array_equal(org, o2)


def test_error_codes_without_a_device(rm):
    from cpu_raymarcher_amd import _native as N
    ctx = rm.Context(None)
    ctx.scene_from_preset(99, 2)  # scene.ts:39 clamps to preset 18 ("67")
    assert ctx.scene_info()["n_prims"] == 2
    ident = np.eye(4, dtype=np.float32).ravel()
    moved = ident.copy()
    moved[12] = 0.25  # a Round takes its operand's transform (round.ts): translated, every level needs a position slot of its own
    chain = [(0, -1, -1, moved, [0.5])] + [(10, i, -1, None, [0.01]) for i in range(20)]  # 20 nested Round operators
    with pytest.raises(rm.RmUnsupported):
        ctx.scene_from_nodes(chain, [len(chain) - 1], 0)
    ctx.scene_from_nodes(chain[:10], [9], 2)
    assert ctx.scene_info()["n_prims"] == 1
    ctx.scene_from_nodes([(0, -1, -1, ident, [0.5])] + [(10, i, -1, None, [0.01]) for i in range(20)], [20], 0)  # identity transforms pass the point through: no slot per level
    assert ctx.scene_info()["n_prims"] == 1
    with pytest.raises(rm.RmError):  # operands must precede their operator
        ctx.scene_from_nodes([(10, 1, -1, None, [0.1]), (0, -1, -1, ident, [0.5])], [0], 0)
    with pytest.raises(rm.RmError):
        ctx.scene_from_nodes([(42, -1, -1, ident, [0.5])], [0], 0)
    ctx.scene_from_preset(-5, 2)  # clamps to preset 0
    assert ctx.scene_info()["n_prims"] == 1
    scene = rm.Scene("BVH", ctx=ctx)
    scene.loadPreset(3)
    buf = [np.zeros(16, np.uint8), np.zeros(48, np.uint8), np.zeros(16, np.uint16), np.zeros(16, np.uint16)]
    with pytest.raises(rm.RmError) as e:  # there is no CPU render path
        rm.SphereTracer().runRaymarcher(scene, *buf, 4, 4)
    assert e.value.code == N.RM_E_NO_DEVICE
    with pytest.raises(rm.RmError):
        ctx.scene_from_spheres(np.array([[0, 0, float("nan")]]), np.array([1.0]), 0)
    with pytest.raises(rm.RmError):
        ctx.set_option("tile_w", 12)
    ctx.set_option("tile_w", 16)
    assert ctx.get_option("tile_w") == 16
    # the round-2 knobs: defaults, accepted values, rejected values, unknown keys
    assert (ctx.get_option("oct_lean"), ctx.get_option("v1_block"), ctx.get_option("v1_lists"), ctx.get_option("lpt")) == (1, 64, 1, 1)
    # the round-3 knobs
    assert (ctx.get_option("multi_step"), ctx.get_option("blocks_per_cu"), ctx.get_option("lds_kb"), ctx.get_option("lds_fill")) == (1, 6, 0, 0)
    ctx.set_option("static", 75)  # accepted, ignored since round 3
    for bad in (8, 65):
        with pytest.raises(rm.RmError):
            ctx.set_option("lds_kb", bad)
    for bad in (0, 32, 96, 512):
        with pytest.raises(rm.RmError):
            ctx.set_option("v1_block", bad)
    for good in (128, 256, 64):
        ctx.set_option("v1_block", good)
        assert ctx.get_option("v1_block") == good
    ctx.set_option("oct_lean", 0)
    assert ctx.get_option("oct_lean") == 0
    ctx.set_option("oct_lean", 7)  # any non-zero value switches it on
    assert ctx.get_option("oct_lean") == 1
    with pytest.raises(rm.RmError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(rm.RmError):
        ctx.get_option("no_such_option")


def test_camera_class_mirrors_reference(rm):
    cam = rm.Camera()
    cam.setAngles(9.0, 1.0)
    assert cam.getAngles() == [math.pi / 2, 1.0]
    cam.rotateCamera(-0.5, 0.015)
    assert cam.getAngles() == [math.pi / 2 - 0.5, 1.015]
    assert isinstance(rm.createRaymarcher("nope"), rm.SphereTracer)
    assert isinstance(rm.createRaymarcher("fixed-step", stepSize=0.1), rm.FixedStep)
    assert isinstance(rm.createShadingModelFromValue("nope"), rm.NormalModel)
    assert isinstance(rm.createShadingModelFromValue("phong"), rm.PhongModel)


def test_synthetic_scene_definitions_agree_with_the_oracles_copy(rm, oracle):
    from cpu_raymarcher_amd import synthetic as S
    assert np.array_equal(S.synthetic_spheres(500), oracle.synthetic_spheres(500))
    assert S.synthetic_mixed_prims(30, seed=11) == oracle.synthetic_mixed_prims(30, seed=11)
    triples = S.mixed_prims_as_triples(S.synthetic_mixed_prims(12), rm.make_transform)
    want = oracle.OracleScene(accel="None", prims=oracle.synthetic_mixed_prims(12)).prims()
    for (t, m, par), (wt, wm, wpar) in zip(triples, want):
        assert t == wt and np.array_equal(m, wm) and list(par) == list(wpar[:len(par)])
