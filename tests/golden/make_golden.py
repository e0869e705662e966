#!/usr/bin/env python3
"""Generates tests/golden/golden.json (+ golden_crops.npz) for the BASELINE.json configs.

The reference ships no golden vectors and cannot run here (TypeScript without
a transpiler, gl-matrix not vendored), so the fixtures are produced by the build's two
independent restatements, which must agree byte for byte before anything is written:
  * oracle/rm_oracle.c  (C, -ffp-contract=off)            -- full size, every config
  * oracle/rm_oracle.js (plain JS on the real engine, node) -- full size where affordable,
    otherwise row bands of the full-size frame plus a reduced-size frame
PARITY UNPINNED against the reference itself; the hand-derivable known answers are in
tests/test_oracle_kat.py.  Run from the repo root in the build container (needs node):
    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
THREADS = max(1, (os.cpu_count() or 2) - 1)

# name -> config.  "js": "full" | list of (yStart, yEnd) bands; "js_reduced": (W, H) or None
CASES = {
    "C1_sphere_256_none_normal": dict(preset=0, accel="None", width=256, height=256, shader="normal", js="full"),
    "C2_grid_1080p_bvh_phong": dict(preset=2, accel="BVH", width=1920, height=1080, shader="phong", js="full"),
    "C3_dense_4k_bvh_iterheat": dict(preset=3, accel="BVH", width=3840, height=2160, shader="iteration-heatmap",
                                     js=[(0, 4), (700, 712), (1078, 1090), (2156, 2160)], js_reduced=(960, 540)),
    "C3r_dense_540p_bvh_rotated_sdfheat": dict(preset=3, accel="BVH", width=960, height=540, pitch=0.3, yaw=0.7,
                                               shader="sdf-heatmap", js="full"),
    "C3o_dense_540p_octree_phong": dict(preset=3, accel="Octree", width=960, height=540, shader="phong", js="full"),
    "C5_random10k_4k_octree_iterheat": dict(synthetic=10000, accel="Octree", width=3840, height=2160,
                                            shader="iteration-heatmap", js=[(0, 2), (1079, 1082), (1500, 1502)],
                                            js_reduced=(480, 270)),
    "C5b_random10k_270p_bvh_wrap": dict(synthetic=10000, accel="BVH", width=480, height=270, pitch=-0.2, yaw=0.4,
                                        shader="sdf-heatmap", js=[(100, 104), (134, 137)]),
    "P1_random7_360p_octree_rot": dict(preset=1, accel="Octree", width=640, height=360, pitch=0.9, yaw=-2.3,
                                       shader="phong", js="full"),
    "P4_atom_360p_bvh": dict(preset=4, accel="BVH", width=640, height=360, pitch=-1.7, yaw=0.1, shader="normal", js="full"),
    # the four other marchers
    "M1_fixed_360p_bvh": dict(preset=3, accel="BVH", width=640, height=360, shader="sdf-heatmap", algorithm="fixed-step",
                              stepSize=0.05, js="full"),
    "M2_adaptive_360p_octree": dict(preset=3, accel="Octree", width=640, height=360, shader="iteration-heatmap",
                                    algorithm="adaptive-step", pitch=0.2, yaw=1.1, js="full"),
    "M3_v2_360p_bvh_rot": dict(preset=3, accel="BVH", width=640, height=360, shader="phong", algorithm="adaptive-step-v2",
                               overshootFactor=1.5, pitch=0.3, yaw=0.7, js="full"),
    "M4_v3_360p_none": dict(preset=1, accel="None", width=640, height=360, shader="normal", algorithm="adaptive-step-v3",
                            js="full"),
    "M5_v3_270p_octree10k": dict(synthetic=10000, accel="Octree", width=480, height=270, shader="iteration-heatmap",
                                 algorithm="adaptive-step-v3", overshootFactor=1.35, js="full"),
    # boxes, tori, rotated transforms
    "N3_torus_360p_bvh": dict(preset=5, accel="BVH", width=640, height=360, shader="phong", pitch=0.4, yaw=0.9, js="full"),
    "N3_cube_360p_none_v2march": dict(preset=7, accel="None", width=640, height=360, shader="normal",
                                      algorithm="adaptive-step-v2", pitch=-0.3, yaw=0.5, js="full"),
    "N3_sphere_and_cube_360p_octree": dict(preset=8, accel="Octree", width=640, height=360, shader="sdf-heatmap", js="full"),
    "N3_pyramid_360p_bvh": dict(preset=9, accel="BVH", width=640, height=360, shader="iteration-heatmap", pitch=0.5,
                                yaw=-1.0, js="full"),
    "N3_mixed40_360p_bvh": dict(mixed=40, accel="BVH", width=640, height=360, shader="phong", pitch=0.2, yaw=0.3, js="full"),
    "N3_mixed40_360p_octree": dict(mixed=40, accel="Octree", width=640, height=360, shader="normal", pitch=-0.6,
                                   yaw=2.0, js="full"),
    # SDF operators and the Mandelbulb; `time` feeds the animated primitives
    "N4_rounded_box_360p_bvh": dict(preset=6, accel="BVH", width=640, height=360, shader="phong", pitch=0.4, yaw=0.6, js="full"),
    "N4_smooth_union_360p_none": dict(preset=10, accel="None", width=640, height=360, shader="normal", pitch=0.5, yaw=-0.4,
                                      js="full"),
    "N4_smooth_sub_360p_octree_v3": dict(preset=11, accel="Octree", width=640, height=360, shader="sdf-heatmap",
                                         algorithm="adaptive-step-v3", pitch=0.3, yaw=0.8, js="full"),
    "N4_smooth_union_anim_360p_none_t2500": dict(preset=12, accel="None", width=640, height=360, shader="phong", time=2500.0,
                                                 js="full"),
    "N4_mandelbulb_270p_bvh_t1000": dict(preset=13, accel="BVH", width=480, height=270, shader="iteration-heatmap",
                                         time=1000.0, pitch=0.2, yaw=0.5, js="full"),
    "N4_twisted_torus_360p_bvh": dict(preset=14, accel="BVH", width=640, height=360, shader="phong", pitch=0.6, yaw=1.2,
                                      js="full"),
    "N4_infinite_spheres_360p_octree": dict(preset=15, accel="Octree", width=640, height=360, shader="iteration-heatmap",
                                            pitch=0.1, yaw=0.2, js="full"),
    "N4_screw_360p_none_adaptive": dict(preset=16, accel="None", width=640, height=360, shader="normal",
                                        algorithm="adaptive-step", pitch=-0.2, yaw=0.9, js="full"),
    "N4_chicken_360p_bvh": dict(preset=17, accel="BVH", width=640, height=360, shader="phong", pitch=0.3, yaw=2.4, js="full"),
    "N4_67_360p_octree": dict(preset=18, accel="Octree", width=640, height=360, shader="sdf-heatmap", js="full"),
    "M6_fixed_default_270p_none": dict(preset=2, accel="None", width=480, height=270, shader="sdf-heatmap",
                                       algorithm="fixed-step", js="full"),
}
CROP = 64  # crop side, centred


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def c_render(cfg, y0=None, y1=None, width=None, height=None):
    W = width or cfg["width"]
    H = height or cfg["height"]
    y0 = 0 if y0 is None else y0
    y1 = H if y1 is None else y1
    spheres = O.synthetic_spheres(cfg["synthetic"]) if "synthetic" in cfg else None
    prims = O.synthetic_mixed_prims(cfg["mixed"]) if "mixed" in cfg else None
    sc = O.OracleScene(preset=cfg.get("preset"), accel=cfg["accel"], spheres=spheres, prims=prims)
    sc.set_angles(cfg.get("pitch", 0.0), cfg.get("yaw", 0.0))
    bands = []
    step = max(1, (y1 - y0 + 4 * THREADS - 1) // (4 * THREADS))
    for a in range(y0, y1, step):
        bands.append((a, min(y1, a + step)))
    with ThreadPoolExecutor(THREADS) as ex:
        parts = list(ex.map(lambda b: sc.render(W, H, b[0], b[1], algorithm=cfg.get("algorithm", "sphere-tracer"),
                                                overshoot_factor=cfg.get("overshootFactor"),
                                                step_size=cfg.get("stepSize"), time=cfg.get("time", 0.0)), bands))
    d, n, s, i = (np.concatenate([p[k] for p in parts]) for k in range(4))
    rgba = O.shade(cfg["shader"], d, n, s, i, W, y1 - y0)
    return d, n, s, i, rgba


def js_render(cfg, y0=None, y1=None, width=None, height=None):
    W = width or cfg["width"]
    H = height or cfg["height"]
    with tempfile.TemporaryDirectory() as td:
        j = dict(accel=cfg["accel"], width=W, height=H, shader=cfg["shader"], pitch=cfg.get("pitch", 0.0),
                 yaw=cfg.get("yaw", 0.0))
        if y0 is not None:
            j["yStart"], j["yEnd"] = y0, y1
        if "synthetic" in cfg:
            sp = O.synthetic_spheres(cfg["synthetic"])
            sp.astype(np.float64).tofile(os.path.join(td, "spheres.f64"))
            j["spheres_file"] = os.path.join(td, "spheres.f64")
        elif "mixed" in cfg:
            j["prims"] = O.synthetic_mixed_prims(cfg["mixed"])
        else:
            j["preset"] = cfg["preset"]
        for k in ("algorithm", "overshootFactor", "stepSize", "time"):
            if k in cfg:
                j[k] = cfg[k]
        with open(os.path.join(td, "cfg.json"), "w") as f:
            json.dump(j, f)
        out = subprocess.check_output(["node", os.path.join(ROOT, "oracle", "rm_oracle.js"), "render",
                                       os.path.join(td, "cfg.json"), os.path.join(td, "out")])
        st = json.loads(out)
        bufs = tuple(np.fromfile(os.path.join(td, "out", n + ".bin"), dtype=dt) for n, dt in
                     [("depth", np.uint8), ("normal", np.uint8), ("sdf", np.uint16), ("iters", np.uint16),
                      ("rgba", np.uint8)])
    return bufs, st


def agree(cbufs, jbufs, what):
    for name, a, b in zip(("depth", "normal", "sdf", "iters", "rgba"), cbufs, jbufs):
        if not np.array_equal(a, b):
            raise SystemExit("C and JS restatements DISAGREE on %s (%s): %d mismatches" % (what, name, int((a != b).sum())))


def main():
    only = sys.argv[1:]
    out_path = os.path.join(HERE, "golden.json")
    golden = json.load(open(out_path)) if (only and os.path.exists(out_path)) else {}
    crops_path = os.path.join(HERE, "golden_crops.npz")
    crops = dict(np.load(crops_path)) if (only and os.path.exists(crops_path)) else {}
    engine = None
    for name, cfg in CASES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        W, H = cfg["width"], cfg["height"]
        full = c_render(cfg)
        tc = time.time() - t0
        d, n, s, i, rgba = full
        checks = []
        if cfg["js"] == "full":
            jb, st = js_render(cfg)
            agree(full, jb, name + " full frame")
            checks.append("full frame")
        else:
            for (a, b) in cfg["js"]:
                jb, st = js_render(cfg, a, b)
                band = (d[a * W:b * W], n[3 * a * W:3 * b * W], s[a * W:b * W], i[a * W:b * W], rgba[4 * a * W:4 * b * W])
                agree(band, jb, "%s rows [%d,%d)" % (name, a, b))
                checks.append("rows [%d,%d)" % (a, b))
            if cfg.get("js_reduced"):
                rw, rh = cfg["js_reduced"]
                cb = c_render(cfg, width=rw, height=rh)
                jb, st = js_render(cfg, width=rw, height=rh)
                agree(cb, jb, "%s reduced %dx%d" % (name, rw, rh))
                checks.append("reduced %dx%d full frame" % (rw, rh))
        engine = st["engine"]
        diag = O.diagnostics(s, i)
        cx, cy = (W - CROP) // 2, (H - CROP) // 2
        idx = (np.arange(cy, cy + CROP)[:, None] * W + np.arange(cx, cx + CROP)[None, :]).ravel()
        crops[name + "/sdf"] = s[idx]
        crops[name + "/iters"] = i[idx]
        crops[name + "/depth"] = d[idx]
        crops[name + "/normal"] = n.reshape(-1, 3)[idx].ravel()
        crops[name + "/rgba"] = rgba.reshape(-1, 4)[idx].ravel()
        golden[name] = dict(
            config={k: v for k, v in cfg.items() if k not in ("js", "js_reduced")},
            sha256=dict(depth=sha(d), normal=sha(n), sdf=sha(s), iters=sha(i), rgba=sha(rgba)),
            diagnostics=diag, crop=dict(x=cx, y=cy, size=CROP),
            js_agreement=checks, js_engine=engine, oracle_seconds=round(tc, 2), oracle_threads=THREADS)
        print("%-40s C %.1fs  avg sdf %.2f  avg it %.2f  max sdf %d  js: %s" %
              (name, tc, diag["total_sdf"] / (W * H), diag["total_iters"] / (W * H), diag["max_sdf"], ", ".join(checks)),
              flush=True)
    with open(out_path, "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    np.savez_compressed(crops_path, **crops)
    print("wrote", out_path, crops_path)


if __name__ == "__main__":
    main()
