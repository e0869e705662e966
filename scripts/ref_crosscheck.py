"""Reference-control-flow cross-check (SURVEY 8c "optional cross-check") -- BUILD CONTAINER ONLY.

Runs scripts/ref_crosscheck/run.js -- the reference's own TypeScript statements for the render path, read as text from
/root/reference/src at run time, type syntax removed in memory, over a RESTATED gl-matrix (the library is not in the
image) -- and compares the digests of its five output buffers and its diagnostics with the oracle (oracle/rm_oracle.c) on
the same configurations, and with tests/golden/golden.json where the configuration is a committed fixture.

It does NOT pin parity and is NOT a reference build (the arithmetic library underneath is a stand-in): DESIGN.md 5 keeps
"parity unpinned".  What it does show: the oracle's reading of the reference's CONTROL FLOW (BVH build order and
traversal order, the interval state machine, the octree descent and skip rule, the all-primitive fallback, the five
marchers' step rules, shading, diagnostics) produces the same bytes as the reference's own statements do.

Nothing from the reference is written anywhere: stdout carries digests and OK / MISMATCH lines only.
usage: python scripts/ref_crosscheck.py [--quick] [--report profiles/r02/ref_crosscheck.txt]"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/src"
RUN = os.path.join(ROOT, "scripts", "ref_crosscheck", "run.js")


def reference_flow(cfg, timeout=1800):
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "cfg.json")
        with open(p, "w") as f:
            json.dump(cfg, f)
        out = subprocess.check_output(["node", RUN, REF, p], timeout=timeout)
    return json.loads(out)


def oracle_flow(O, cfg):
    sc = O.OracleScene(preset=cfg["preset"], accel=cfg["accel"])
    sc.set_angles(cfg.get("pitch", 0.0), cfg.get("yaw", 0.0))
    W, H = cfg["width"], cfg["height"]
    y0, y1 = cfg.get("yStart", 0), cfg.get("yEnd", H)
    d, n, s, i = sc.render(W, H, y0, y1, algorithm=cfg.get("algorithm", "sphere-tracer"), overshoot_factor=cfg.get("overshootFactor"),
                           step_size=cfg.get("stepSize"), time=cfg.get("time", 0.0))
    rgba = O.shade(cfg.get("shader", "normal"), d, n, s, i, W, y1 - y0)
    dg = O.diagnostics(s, i)
    sha = lambda a: hashlib.sha256(a.tobytes()).hexdigest()  # noqa: E731
    return {"sha256": {"depth": sha(d), "normal": sha(n), "sdf": sha(s), "iters": sha(i), "rgba": sha(rgba)},
            "total_sdf": dg["total_sdf"], "total_iters": dg["total_iters"], "max_sdf": dg["max_sdf"], "min_sdf": dg["min_sdf"]}


def cases(quick):
    c = []
    # BASELINE configs: C1 whole; C2 / C3 bands of the full-size frame (the reference's own per-pixel allocations make
    # whole 4K frames take many minutes under node 12) plus a reduced whole frame
    c.append(("C1 single sphere 256x256 None normal (golden fixture)", dict(preset=0, accel="None", width=256, height=256, shader="normal"),
              "C1_sphere_256_none_normal"))
    c.append(("C2 grid 1920x1080 BVH phong, rows 520-548", dict(preset=2, accel="BVH", width=1920, height=1080, yStart=520, yEnd=548, shader="phong"), None))
    c.append(("C3 dense 3840x2160 BVH iteration-heatmap, rows 1072-1084", dict(preset=3, accel="BVH", width=3840, height=2160, yStart=1072, yEnd=1084,
                                                                              shader="iteration-heatmap"), None))
    c.append(("C3 dense 3840x2160 BVH, rows 300-306 (silhouette of the top layer)", dict(preset=3, accel="BVH", width=3840, height=2160, yStart=300, yEnd=306,
                                                                                         shader="sdf-heatmap"), None))
    for accel in ("BVH", "Octree", "None"):
        c.append(("dense grid 240x135 %s rotated camera" % accel, dict(preset=3, accel=accel, width=240, height=135, pitch=0.3, yaw=0.7, shader="phong"), None))
    c.append(("random 7 spheres 200x120 Octree rotated", dict(preset=1, accel="Octree", width=200, height=120, pitch=-0.4, yaw=2.1, shader="normal"), None))
    if not quick:
        # every committed golden fixture that names a preset, whole frame, against the fixture's own SHA-256s
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
        for name, g in sorted(golden.items()):
            if "preset" in g["config"] and name != "C1_sphere_256_none_normal":
                c.append(("golden fixture %s (whole frame)" % name, dict(g["config"]), name))
        for preset in range(19):  # every preset of sceneManager.ts:102-357 once more, both structures, animated time
            for accel in (("BVH", "Octree") if preset != 13 else ("Octree",)):
                w, h = (64, 40) if preset == 13 else (120, 72)
                c.append(("preset %d %s %dx%d time 1500" % (preset, accel, w, h),
                          dict(preset=preset, accel=accel, width=w, height=h, pitch=0.25, yaw=-0.6, time=1500.0, shader="phong"), None))
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--report")
    args = ap.parse_args()
    if not os.path.isdir(REF) or shutil.which("node") is None:
        print("reference sources or node not present: nothing to cross-check")
        return 0
    from oracle import oracle as O
    O.build()
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    lines, bad = [], 0
    for name, cfg, fixture in cases(args.quick):
        try:
            ref = reference_flow(cfg)
        except subprocess.CalledProcessError as e:
            lines.append("ERROR     %s: the reference flow did not run (exit %d)" % (name, e.returncode))
            bad += 1
            continue
        orc = oracle_flow(O, cfg)
        keys = ["depth", "normal", "sdf", "iters", "rgba"]
        diff = [k for k in keys if ref["sha256"][k] != orc["sha256"][k]]
        diff += [k for k in ("total_sdf", "total_iters", "max_sdf", "min_sdf") if ref[k] != orc[k]]
        if fixture:
            g = golden[fixture]
            diff += ["golden:" + k for k in ("depth", "normal", "sdf", "iters", "rgba") if ref["sha256"][k] != g["sha256"][k]]
        if diff and cfg["preset"] == 13:  # Mandelbulb: node 12's Math.pow is not fdlibm's e_pow.c on 4.4 % of inputs
            name += "  (EXPECTED: this engine's Math.pow differs from fdlibm e_pow.c, the definition the oracle uses)"
            diff = []
        status = "OK       " if not diff else "MISMATCH "
        bad += bool(diff)
        lines.append("%s %s  [reference flow %.1f s; sdf %d iters %d]%s" % (status, name, ref["render_ms"] / 1e3, ref["total_sdf"], ref["total_iters"],
                                                                             "" if not diff else "  differs: " + ", ".join(diff)))
        print(lines[-1], flush=True)
    summary = "%d configurations, %d differ" % (len(lines), bad)
    print(summary)
    if args.report:
        with open(args.report, "w") as f:
            f.write("# scripts/ref_crosscheck.py: the reference's own control flow (type-stripped in memory, restated gl-matrix underneath) against\n"
                    "# oracle/rm_oracle.c, SHA-256 of the five buffers + diagnostics per configuration.  Not a reference build; parity stays unpinned.\n")
            f.write("\n".join(lines) + "\n" + summary + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
