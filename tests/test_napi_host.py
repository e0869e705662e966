"""The reference's host language is TypeScript/JS: these tests drive the C ABI from node through
the N-API addon (native/rm_addon.cc) and the JS mirror of the reference's classes
(native/host/raymarcher.js), i.e. the integration INTEGRATION.md describes."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
HAVE_HDR = os.path.exists("/usr/include/node/node_api.h")
pytestmark = pytest.mark.skipif(NODE is None or not HAVE_HDR, reason="node / node_api.h not present")


@pytest.fixture(scope="module")
def addon(rm):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "native")])
    return os.path.join(ROOT, "native", "build", "rm_addon.node")


def test_addon_loads_and_exports(addon):
    out = subprocess.check_output([NODE, "-e", "const a=require(%r);console.log(JSON.stringify([Object.keys(a).sort(),"
                                   "a.version(),a.create(-1)]))" % addon])
    keys, version, rc = json.loads(out)
    assert keys == ["create", "diagnostics", "lastError", "renderTile", "shade", "version"]
    assert "gfx950" in version and rc == 0  # host-only context: no GPU needed


def test_host_only_context_refuses_to_render(addon):
    js = ("const a=require(%r);a.create(-1);const r=a.renderTile({width:4,height:4,yStart:0,yEnd:4,camera:{pitch:0,yaw:0},"
          "algorithm:'sphere-tracer',scenePresetIndex:0,accelerationStructure:'BVH'},new Uint8ClampedArray(16),"
          "new Uint8ClampedArray(48),new Uint16Array(16),new Uint16Array(16));console.log(r, a.lastError())" % addon)
    out = subprocess.check_output([NODE, "-e", js]).decode()
    assert out.startswith("-3 ")  # RM_E_NO_DEVICE: there is no CPU render path


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(preset=0, accel="None", width=256, height=256, shader="normal"),
    dict(preset=3, accel="BVH", width=320, height=200, shader="iteration-heatmap", pitch=0.3, yaw=0.7),
    dict(preset=3, accel="Octree", width=200, height=120, shader="phong", yStart=17, yEnd=93),
    dict(preset=2, accel="BVH", width=160, height=90, shader="sdf-heatmap", algorithm="no-such-marcher"),
    dict(preset=3, accel="BVH", width=160, height=90, shader="sdf-heatmap", algorithm="adaptive-step-v3", overshootFactor=1.4),
    dict(preset=3, accel="Octree", width=160, height=90, shader="normal", algorithm="fixed-step"),
    dict(preset=5, accel="BVH", width=160, height=90, shader="phong", pitch=0.4, yaw=0.9),  # rotated torus
    dict(preset=12, accel="None", width=160, height=90, shader="phong", time=2500.0),        # AnimatedTranslate: Job.time
    dict(preset=16, accel="Octree", width=160, height=90, shader="normal", pitch=0.2, yaw=0.5),  # Round(Twist(Box))
    dict(preset=13, accel="BVH", width=96, height=64, shader="iteration-heatmap", time=1000.0),  # Mandelbulb
])
def test_node_worker_matches_oracle(addon, oracle, tmp_path, cfg):
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    out = subprocess.check_output([NODE, os.path.join(ROOT, "native", "render_cli.js"), str(tmp_path / "cfg.json"),
                                   str(tmp_path / "out")])
    diag = json.loads(out)
    W, H = cfg["width"], cfg["height"]
    y0, y1 = cfg.get("yStart", 0), cfg.get("yEnd", H)
    sc = oracle.OracleScene(preset=cfg["preset"], accel=cfg["accel"])
    sc.set_angles(cfg.get("pitch", 0.0), cfg.get("yaw", 0.0))
    d, n, s, i = sc.render(W, H, y0, y1, algorithm=cfg.get("algorithm", "sphere-tracer"),
                           overshoot_factor=cfg.get("overshootFactor"), step_size=cfg.get("stepSize"),
                           time=cfg.get("time", 0.0))
    rgba = oracle.shade(cfg["shader"], d, n, s, i, W, y1 - y0)
    for name, arr in (("depth", d), ("normal", n), ("sdf", s), ("iters", i), ("rgba", rgba)):
        got = np.fromfile(str(tmp_path / "out" / (name + ".bin")), dtype=arr.dtype)
        if name == "rgba" and cfg["shader"] == "phong":
            assert np.abs(got.astype(np.int16) - arr.astype(np.int16)).max() <= 1
        else:
            assert np.array_equal(got, arr), name
    od = oracle.diagnostics(s, i)
    assert (diag["totalSDFCalls"], diag["maxSDFCalls"], diag["minSDFCalls"], diag["totalIterations"]) == \
        (od["total_sdf"], od["max_sdf"], od["min_sdf"], od["total_iters"])


@pytest.mark.gpu
def test_node_worker_reports_errors(addon, tmp_path):
    cfg = dict(preset=6, accel="BVH", width=16, height=0, shader="normal")  # height 0: u, v would divide by zero
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    p = subprocess.run([NODE, os.path.join(ROOT, "native", "render_cli.js"), str(tmp_path / "cfg.json"),
                        str(tmp_path / "out")], stdout=subprocess.PIPE)
    assert p.returncode == 1 and json.loads(p.stdout)["code"] == -1  # RM_E_INVALID surfaces as an Error with .code


def test_worker_pool_owns_one_context_per_worker(addon, tmp_path):
    """the addon kept one process-global ctx, so a worker's create() destroyed the ctx another
    worker was using.  Now the ctx is per napi_env.  Four worker_threads on host-only contexts, every second job
    re-creating the worker's own ctx while the others are in renderTile: every status must be RM_E_NO_DEVICE
    (a destroyed foreign ctx would crash or report RM_E_INVALID)."""
    cfg = dict(width=64, height=48, workers=4, frames=8, device=-1, preset=3, accel="BVH", recreate=True)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    out = subprocess.check_output([NODE, os.path.join(ROOT, "native", "pool_cli.js"), str(tmp_path / "cfg.json")], timeout=120)
    res = json.loads(out)
    assert res["statuses"] == [-3] * (4 * 8)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(preset=3, accel="BVH", width=320, height=203, workers=4, frames=3, pitch=0.1, yaw=0.3),
    dict(preset=3, accel="Octree", width=200, height=120, workers=3, frames=2, recreate=True),
])
def test_worker_pool_renders_concurrently(addon, oracle, tmp_path, cfg):
    """The reference's pool (main.ts:318-321, fan-out/fan-in :444-468) over worker_threads: N workers render their
    ceil(H/N)-row tiles at the same time, each through its own rm_ctx on the same GPU; the gathered frame of the
    last frame (yaw advanced by 0.015 per frame, main.ts:438-441) equals the oracle's."""
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    out = subprocess.check_output([NODE, os.path.join(ROOT, "native", "pool_cli.js"), str(tmp_path / "cfg.json"),
                                   str(tmp_path / "out")], timeout=300)
    res = json.loads(out)
    assert res["statuses"] == [0] * (cfg["workers"] * cfg["frames"])
    W, H = cfg["width"], cfg["height"]
    sc = oracle.OracleScene(preset=cfg["preset"], accel=cfg["accel"])
    sc.set_angles(cfg.get("pitch", 0.0), cfg.get("yaw", 0.0) + 0.015 * (cfg["frames"] - 1))
    d, n, s, i = sc.render(W, H)
    for name, arr in (("depth", d), ("normal", n), ("sdf", s), ("iters", i)):
        got = np.fromfile(str(tmp_path / "out" / (name + ".bin")), dtype=arr.dtype)
        assert np.array_equal(got, arr), name
