"""CPU tests of the oracle itself: hand-derivable known answers from the reference source
 and the committed golden fixtures.  No GPU."""
import hashlib
import math
import shutil
import subprocess

import numpy as np
import pytest


def test_c1_centre_pixel_kat(oracle):
    # raymarcher.ts:73-106 by hand: dir (0,0,-1); step 1 dist 1.5; step 2 dist 0 < eps -> iters 2;
    # normal costs 4 -> sdfEval 6; depth 1.5 -> u8c 2 (ties to even); n ~ (-3.3e-5,-3.3e-5,.01) -> (127,127,255)
    sc = oracle.OracleScene(preset=0, accel="None")
    d, n, s, i = sc.render(256, 256)
    px = 128 * 256 + 128
    assert (d[px], s[px], i[px]) == (2, 6, 2)
    assert tuple(n[3 * px:3 * px + 3]) == (127, 127, 255)
    rgba = oracle.shade("normal", d, n, s, i, 256, 256)
    assert tuple(rgba[4 * px:4 * px + 4]) == (127, 127, 255, 255)
    heat = oracle.shade("iteration-heatmap", d, n, s, i, 256, 256)
    assert tuple(heat[4 * px:4 * px + 4]) == (20, 255, 0, 255)  # 2*5=10 -> R 20, G min(492,255)


def test_c1_corner_pixel_kat(oracle):
    # ray misses (closest approach sqrt(6) > 1.5): zero normal -> (128,128,128) (127.5 ties to even),
    # one primitive and no normal pass -> sdfEval == iters, depth byte >= 10
    sc = oracle.OracleScene(preset=0, accel="None")
    d, n, s, i = sc.render(256, 256)
    assert tuple(n[0:3]) == (128, 128, 128)
    assert s[0] == i[0] and d[0] >= 10


def test_bvh_root_miss_kat(oracle):
    # sphereTracer.ts:38-40 + raymarcher.ts:98-99: no interval -> exactly MAX_DIST, 0 iterations, 0 SDF calls
    sc = oracle.OracleScene(preset=2, accel="BVH")
    d, n, s, i = sc.render(64, 64)
    assert (d[0], s[0], i[0]) == (10, 0, 0) and tuple(n[0:3]) == (128, 128, 128)


def test_bvh_shape_kat(oracle):
    # bvh.ts:44-92 on the 125-sphere grid: 64 leaves / 127 nodes / depth 6; root = f32(+-1.2 +- 0.225)
    sc = oracle.OracleScene(preset=3, accel="BVH")
    st = sc.stats()
    assert (st["bvh_nodes"], st["bvh_leaves"], st["bvh_depth"]) == (127, 64, 6)
    r = 0.15 * 1 * 1.5
    lo = np.float32(np.float64(np.float32(0 * 0.6 - 1.2)) - r)
    hi = np.float32(np.float64(np.float32(4 * 0.6 - 1.2)) + r)
    rb = sc.root_bounds()
    assert np.all(rb[:3] == lo) and np.all(rb[3:] == hi)
    st9 = oracle.OracleScene(preset=2, accel="BVH").stats()
    assert (st9["bvh_nodes"], st9["bvh_leaves"]) == (9, 5)
    # leaf count recurrence L(n) = 1 for n <= 2 else L(floor(n/2)) + L(ceil(n/2))
    def L(n):
        return 1 if n <= 2 else L(n // 2) + L(n - n // 2)
    sp = oracle.synthetic_spheres(1000)
    st = oracle.OracleScene(spheres=sp, accel="BVH").stats()
    assert st["bvh_leaves"] == L(1000) and st["bvh_nodes"] == 2 * L(1000) - 1


def test_js_number_semantics(oracle):
    L = oracle.lib()
    # Uint8ClampedArray store: NaN -> 0, clamp, ties to even (Appendix A.3)
    for x, want in [(0.5, 0), (1.5, 2), (2.5, 2), (127.5, 128), (254.5, 254), (254.51, 255), (-3, 0), (300, 255),
                    (float("nan"), 0), (254.49999, 254), (1e-9, 0)]:
        assert L.ro_u8clamp(x) == want, x
    # Math.hypot scaled-Kahan form: exact on 3-4-12 -> 13 and scale invariance by powers of two
    assert L.ro_hypot3(3.0, 4.0, 12.0) == 13.0
    assert L.ro_hypot3(0.0, 0.0, 0.0) == 0.0
    a = L.ro_hypot3(0.1, 0.2, 0.3)
    assert L.ro_hypot3(0.1 * 2 ** 40, 0.2 * 2 ** 40, 0.3 * 2 ** 40) == a * 2 ** 40
    assert abs(a - math.sqrt(0.14)) < 1e-15


def test_default_camera(oracle):
    sc = oracle.OracleScene(preset=0, accel="None")
    rot, org = sc.camera()
    assert np.array_equal(rot, np.eye(3, dtype=np.float32).ravel())
    assert tuple(org) == (0.0, 0.0, 3.0)
    sc.set_angles(5.0, 0.0)  # pitch clamps to pi/2 (camera.ts:59)
    rot2, _ = sc.camera()
    sc.set_angles(math.pi / 2, 0.0)
    rot3, _ = sc.camera()
    assert np.array_equal(rot2, rot3)


def test_tile_independence_and_wrap(oracle):
    # tile-local idx, full-frame u,v (raymarcher.ts:72-76,83): any row split is exact
    sc = oracle.OracleScene(preset=3, accel="Octree")
    sc.set_angles(0.3, 0.7)
    full = sc.render(96, 80)
    a = sc.render(96, 80, 0, 33)
    b = sc.render(96, 80, 33, 80)
    for f, x, y in zip(full, a, b):
        assert np.array_equal(f, np.concatenate([x, y]))
    # Uint16Array += wraps mod 65536 (raymarcher.ts:119): 10k primitives, no acceleration, >= 7 steps
    sp = oracle.synthetic_spheres(10000)
    sc = oracle.OracleScene(spheres=sp, accel="None")
    d, n, s, i = sc.render(16, 16, 8, 9)
    assert np.any((i.astype(np.int64) * 10000 > 65535))
    extra = (s.astype(np.int64) - i.astype(np.int64) * 10000) % 65536  # + 4 evaluations when a normal is taken
    assert np.all((extra == 0) | (extra == 40000))


def test_empty_and_degenerate_inputs(oracle):
    sc = oracle.OracleScene(spheres=np.zeros((0, 4)), accel="BVH")
    d, n, s, i = sc.render(8, 8)
    assert np.all(s == 0)
    sc = oracle.OracleScene(preset=0, accel="None")
    d, n, s, i = sc.render(8, 8, 5, 5)  # empty tile
    assert d.size == 0
    assert oracle.OracleScene(preset=99, accel="None").stats()["n"] == 2  # index clamps to the last preset ("67")
    assert np.array_equal(sc.render(8, 8, algorithm="no-such-marcher")[2], sc.render(8, 8)[2])  # default branch


def test_golden_small_cases(oracle, golden):
    """The committed fixtures (made from C == JS agreement) still match the C oracle."""
    for name, g in golden.items():
        cfg = g["config"]
        if cfg["width"] * cfg["height"] > 700 * 400:
            continue  # full-size cases are checked on the GPU box against the HIP path
        spheres = oracle.synthetic_spheres(cfg["synthetic"]) if "synthetic" in cfg else None
        prims = oracle.synthetic_mixed_prims(cfg["mixed"]) if "mixed" in cfg else None
        sc = oracle.OracleScene(preset=cfg.get("preset"), accel=cfg["accel"], spheres=spheres, prims=prims)
        sc.set_angles(cfg.get("pitch", 0.0), cfg.get("yaw", 0.0))
        d, n, s, i = sc.render(cfg["width"], cfg["height"], algorithm=cfg.get("algorithm", "sphere-tracer"),
                               overshoot_factor=cfg.get("overshootFactor"), step_size=cfg.get("stepSize"),
                               time=cfg.get("time", 0.0))
        rgba = oracle.shade(cfg["shader"], d, n, s, i, cfg["width"], cfg["height"])
        for key, arr in zip(("depth", "normal", "sdf", "iters", "rgba"), (d, n, s, i, rgba)):
            assert hashlib.sha256(arr.tobytes()).hexdigest() == g["sha256"][key], (name, key)
        dg = oracle.diagnostics(s, i)
        assert dg == g["diagnostics"], name


@pytest.mark.skipif(shutil.which("node") is None, reason="node (JS engine) not present")
def test_c_oracle_matches_js_restatement(oracle, tmp_path):
    """Second restatement on a real JS engine agrees byte for byte (small case; the fixtures
    carry the large ones)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = dict(preset=3, accel="BVH", width=96, height=64, shader="phong", pitch=0.25, yaw=-0.6)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    subprocess.check_output(["node", os.path.join(root, "oracle", "rm_oracle.js"), "render",
                             str(tmp_path / "cfg.json"), str(tmp_path / "out")])
    sc = oracle.OracleScene(preset=3, accel="BVH")
    sc.set_angles(0.25, -0.6)
    d, n, s, i = sc.render(96, 64)
    rgba = oracle.shade("phong", d, n, s, i, 96, 64)
    for name, arr in (("depth", d), ("normal", n), ("sdf", s), ("iters", i), ("rgba", rgba)):
        js = np.fromfile(str(tmp_path / "out" / (name + ".bin")), dtype=arr.dtype)
        assert np.array_equal(js, arr), name
    # Math.hypot and Math.sin/cos (camera) of the engine vs the C restatement
    rng = np.random.default_rng(7)
    trip = rng.standard_normal((20000, 3)).astype(np.float32).astype(np.float64)
    trip.tofile(str(tmp_path / "h.f64"))
    subprocess.check_call(["node", os.path.join(root, "oracle", "rm_oracle.js"), "hypot", str(tmp_path / "h.f64"),
                           str(tmp_path / "h.out")])
    js = np.fromfile(str(tmp_path / "h.out"), dtype=np.float64)
    L = oracle.lib()
    mine = np.array([L.ro_hypot3(*t) for t in trip])
    assert np.array_equal(js, mine)
    ang = rng.uniform(-7, 7, (4000, 2))
    ang.tofile(str(tmp_path / "a.f64"))
    subprocess.check_call(["node", os.path.join(root, "oracle", "rm_oracle.js"), "camera", str(tmp_path / "a.f64"),
                           str(tmp_path / "a.out")])
    js = np.fromfile(str(tmp_path / "a.out"), dtype=np.float32).reshape(-1, 12)
    sc = oracle.OracleScene(preset=0, accel="None")
    for k in range(len(ang)):
        sc.set_angles(ang[k, 0], ang[k, 1])
        rot, org = sc.camera()
        assert np.array_equal(js[k, :9], rot) and np.array_equal(js[k, 9:], org), ang[k]


def _jsmath(oracle, fn, a, b=None):
    import ctypes
    a = np.ascontiguousarray(a, np.float64)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, np.float64)
    out = np.zeros_like(a)
    L = oracle.lib()
    L.ro_jsmath_eval.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_long]
    L.ro_jsmath_eval(fn, a.ctypes.data, b.ctypes.data, out.ctypes.data, len(a))
    return out


def _jsmath_inputs(rng, n):
    trig = np.concatenate([rng.uniform(-40, 40, n // 2), rng.uniform(-8e5, 8e5, n // 8), rng.normal(0, 1e-3, n // 8),
                           (np.arange(n // 8) - n // 16) * np.pi / 2 * (1 + rng.normal(0, 1e-9, n // 8)),
                           # beyond 2^19*pi/2: the Payne-Hanek reduction of k_rem_pio2.c
                           rng.uniform(8e5, 1e7, n // 32), 10.0 ** rng.uniform(6, 300, n // 16) * rng.choice([-1, 1], n // 16),
                           2.0 ** rng.integers(20, 1023, n // 32).astype(float)])
    y = rng.normal(0, 1, n) * 10.0 ** rng.integers(-5, 5, n)
    x = rng.normal(0, 1, n) * 10.0 ** rng.integers(-5, 5, n)
    unit = np.concatenate([rng.uniform(-1, 1, n - 1000), 1 - 10.0 ** rng.uniform(-16, 0, 1000)])
    pos = np.concatenate([rng.uniform(0, 4, n // 2), 10.0 ** rng.uniform(-300, 300, n // 2)])
    rnd = np.concatenate([rng.uniform(-100, 100, n - 9), [0.5, -0.5, 1.5, -1.5, 2.5, -2.5, -0.0, 0.49999999999999994, -0.2]])
    return {0: (trig, None), 1: (trig, None), 2: (y, x), 3: (unit, None), 4: (pos, None), 6: (rnd, None), 7: (y, None)}


@pytest.mark.skipif(shutil.which("node") is None, reason="node (JS engine) not present")
def test_jsmath_equals_node(oracle, tmp_path):
    """oracle/ro_jsmath.h (fdlibm restated) == the engine's Math.sin/cos/atan2/asin/log/round/atan, bit for bit.
    Math.pow is engine-version dependent (node 12's differs from the fdlibm e_pow.c current V8 uses): the
    C and the JS port of e_pow.c must agree with each other instead, and stay within 1 ulp of the exact value."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    js = os.path.join(root, "oracle", "rm_oracle.js")
    rng = np.random.default_rng(1)
    n = 200000

    def node(fn, a, b):
        a.tofile(str(tmp_path / "a.f64"))
        (np.zeros_like(a) if b is None else b).tofile(str(tmp_path / "b.f64"))
        subprocess.check_call(["node", js, "jsmath", str(fn), str(tmp_path / "a.f64"), str(tmp_path / "b.f64"),
                               str(tmp_path / "o.f64")])
        return np.fromfile(str(tmp_path / "o.f64"))

    for fn, (a, b) in _jsmath_inputs(rng, n).items():
        mine, ref = _jsmath(oracle, fn, a, b), node(fn, a, b)
        same = (mine.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(mine) & np.isnan(ref))
        assert same.all(), (fn, int((~same).sum()))
    xs = np.concatenate([rng.uniform(0, 2.5, n // 2), 10.0 ** rng.uniform(-20, 20, n // 4), rng.uniform(-5, 5, n // 4)])
    ys = np.concatenate([rng.choice([7.0, 8.0, 2.0, 0.5, 3.0, -1.0], n // 4), rng.uniform(-10, 10, n // 4),
                         rng.uniform(-30, 30, n // 4), rng.integers(-9, 9, n // 4).astype(float)])
    mine, ref = _jsmath(oracle, 5, xs, ys), node(8, xs, ys)
    same = (mine.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(mine) & np.isnan(ref))
    assert same.all(), int((~same).sum())


def test_jsmath_known_answers(oracle):
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    f = lambda fn, a, b=None: _jsmath(oracle, fn, [a], None if b is None else [b])[0]  # noqa: E731
    assert f(0, 0.0) == 0.0 and f(1, 0.0) == 1.0 and f(4, 1.0) == 0.0 and f(3, 0.0) == 0.0
    assert f(3, 1.0) == math.pi / 2 and f(2, 1.0, 0.0) == math.pi / 2 and f(2, 0.0, -1.0) == math.pi
    assert f(5, 2.0, 7.0) == 128.0 and f(5, 1.5, 2.0) == 2.25 and f(5, 9.0, 0.5) == 3.0 and f(5, 3.0, 0.0) == 1.0
    assert math.isnan(f(5, -8.0, 1.0 / 3)) and f(5, -2.0, 3.0) == -8.0 and math.isnan(f(3, 1.5)) and math.isnan(f(4, -1.0))
    # Math.round: ties toward +inf, [-0.5, -0] -> -0
    for x, want in ((0.5, 1.0), (1.5, 2.0), (2.5, 3.0), (-1.5, -1.0), (-2.5, -2.0), (0.49999999999999994, 0.0), (-0.6, -1.0)):
        assert f(6, x) == want, x
    assert math.copysign(1.0, f(6, -0.5)) == -1.0 and math.copysign(1.0, f(6, -0.2)) == -1.0
    # accuracy of the restated kernels against 60-digit arithmetic (fdlibm: < 1 ulp)
    rng = np.random.default_rng(5)
    xs, ys = rng.uniform(0.01, 2.5, 3000), rng.choice([7.0, 8.0, 2.5, -3.3], 3000)
    got = _jsmath(oracle, 5, xs, ys)
    for x, y, r in zip(xs, ys, got):
        exact = (Decimal(float(y)) * Decimal(float(x)).ln()).exp()
        assert abs(Decimal(float(r)) - exact) < Decimal(math.ulp(r)), (x, y)
    got = _jsmath(oracle, 4, xs)
    for x, r in zip(xs, got):
        assert abs(Decimal(float(r)) - Decimal(float(x)).ln()) < Decimal(math.ulp(r)), x


def test_operator_known_answers(oracle):
    """Hand-derived from primitive_operations/*.ts at points where every step is exact in binary32/64."""
    d = lambda prims, p, time=0.0: oracle.OracleScene(accel="None", prims=prims).distance(p, time)[0]  # noqa: E731
    box = {"type": "box", "pos": (0, 0, 0), "half": (0.5, 0.5, 0.5)}
    sph = {"type": "sphere", "pos": (0, 0, 0), "r": 0.5}
    assert d([box], (2, 0, 0)) == 1.5
    assert d([{"type": "round", "a": box, "radius": 0.25}], (2, 0, 0)) == 1.25          # round.ts:24
    # smoothUnion.ts:31-34 with d1 = 1.5 (box), d2 = 1.5 (sphere), k = 4 * 0.25: h = 1, min - 1*1*0.25/1
    assert d([{"type": "smoothUnion", "a": box, "b": sph, "k": 0.25}], (2, 0, 0)) == 1.25
    # smoothSubstraction.ts:30-33: h = max(1 - |1.5 + 1.5|, 0) = 0 -> max(d1, -d2) = 1.5
    assert d([{"type": "smoothSub", "a": box, "b": sph, "k": 0.25}], (2, 0, 0)) == 1.5
    # repetition.ts:22-24 spacing 1: q = p - round(p) = (0.25, 0, 0) -> |q| - r = -0.25
    assert d([{"type": "repetition", "a": sph, "spacing": (1, 1, 1)}], (3.25, 2, -4)) == -0.25
    # twist.ts:23-33 at y = 0: cos 0 = 1, sin 0 = 0 -> untouched point
    assert d([{"type": "twist", "a": box, "amount": 10.0}], (2, 0, 0)) == 1.5
    # animatedTranslate.ts:36: time * speed = 0 -> offset 0; the operand re-applies its transform (identity here)
    anim = {"type": "anim", "a": sph, "direction": (0, 2, 0), "amplitude": 2.0, "speed": 0.5}
    assert d([anim], (0, 2, 0)) == 1.5
    # ... and with sin(time*speed)*amplitude = 2 sin(pi/2 rounded) = 2: the sphere sits at y = 2
    assert abs(d([anim], (0, 2, 0), time=math.pi) - (-0.5)) < 1e-15
    # mandelbulb.ts:46-48: |p| > 2 in local space (world * 0.5) breaks at once: 0.5 * log(r) * r / 1
    mb = {"type": "mandelbulb", "pos": (0, 0, 0), "power": 8, "iterations": 9, "animate": False, "speed": 0.0}
    assert d([mb], (8, 0, 0)) == 0.5 * math.log(4.0) * 4.0
