"""Reproducer of round 1's wrong normals (DESIGN.md "Toolchain note"): build the library with
-DRM_REPRO_GENERAL_BRANCH (list_min chooses the primitive representation with a RUN-TIME `if (P.general)` inside the
sphere instantiation), point RM_HIP_LIB at it and run this: every marcher x accel on sphere presets against the oracle,
with a description of the pixels that differ.
usage: RM_HIP_LIB=$PWD/cpu_raymarcher_amd/librm_hip_repro.so python scripts/repro_general_branch.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cpu_raymarcher_amd as R
from oracle import oracle as O

W, H = 220, 140
ctx = R.Context(0)
bad_total = 0
for accel in ("None", "BVH", "Octree"):
    for alg in ("sphere-tracer", "fixed-step", "adaptive-step", "adaptive-step-v2", "adaptive-step-v3"):
        for preset, ang in ((3, (0.2, 0.5)), (1, (0.0, 0.0))):
            sc = R.Scene(accel, ctx=ctx)
            sc.loadPreset(preset)
            sc.camera.setAngles(*ang)
            n = W * H
            got = (np.zeros(n, np.uint8), np.zeros(3 * n, np.uint8), np.zeros(n, np.uint16), np.zeros(n, np.uint16))
            R.createRaymarcher(alg).runRaymarcher(sc, *got, W, H, 0.0)
            osc = O.OracleScene(preset=preset, accel=accel)
            osc.set_angles(*ang)
            want = osc.render(W, H, algorithm=alg)
            diffs = [int((g != w).sum()) for g, w in zip(got, want)]
            if any(diffs):
                bad_total += 1
                nb = (got[1].reshape(-1, 3) != want[1].reshape(-1, 3)).any(1)
                idx = np.nonzero(nb)[0]
                zero = int((got[1].reshape(-1, 3)[idx] == 128).all(1).sum())
                print("%-8s %-17s preset %d: depth %d normal %d sdf %d iters %d differ | %d pixels with a wrong normal, %d of them (128,128,128); "
                      "their depth bytes got %s want %s; sdf got-want %s; lanes (x mod 64) %s"
                      % (accel, alg, preset, *diffs, len(idx), zero, got[0][idx][:6], want[0][idx][:6],
                         (got[2][idx].astype(int) - want[2][idx].astype(int))[:6], (idx % W % 64)[:12]))
print("kernel of the last launch:", ctx.last_kernel())
print("cases that differ from the oracle:", bad_total)
