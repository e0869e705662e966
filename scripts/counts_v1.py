"""Diagnostic: how often each part of the v1 (one ray per lane) octree path executes in one C5 frame.  Needs the -DRM_COUNTS
build: make -C cpu_raymarcher_amd/csrc EXTRA=-DRM_COUNTS OUT=../librm_hip_counts.so ; RM_HIP_LIB=.../librm_hip_counts.so.
Per event: wave-level executions, lanes active in them, lane utilisation, per-pixel figures.
usage: RM_HIP_LIB=cpu_raymarcher_amd/librm_hip_counts.so python scripts/counts_v1.py [k=v ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cpu_raymarcher_amd as R
from cpu_raymarcher_amd import _native as N
from cpu_raymarcher_amd.synthetic import synthetic_spheres

W, H = 3840, 2160
ctx = R.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
sc = R.Scene("Octree", ctx=ctx)
sp = synthetic_spheres(10000)
sc.loadSpheres(sp[:, :3], sp[:, 3])
dev = torch.device("cuda:0")
d = torch.zeros(W * H, dtype=torch.uint8, device=dev)
nb = torch.zeros(3 * W * H, dtype=torch.uint8, device=dev)
s = torch.zeros(W * H, dtype=torch.int16, device=dev)
it = torch.zeros(W * H, dtype=torch.int16, device=dev)
rg = torch.zeros(4 * W * H, dtype=torch.uint8, device=dev)
out = np.zeros(32, np.uint64)
N.lib().rm_debug_read_counts(ctx._h, out.ctypes.data_as(C.c_void_p))
R.SphereTracer().runRaymarcher(sc, d, nb, s, it, W, H, 0.0, shadedBuffer=rg, shader="iteration-heatmap")
torch.cuda.synchronize()
N.lib().rm_debug_read_counts(ctx._h, out.ctypes.data_as(C.c_void_p))
print(ctx.last_kernel())
names = ["march-loop iteration", "skip taken", "leaf evaluation (prims > 0)", "skip chained inside the same empty leaf", "(bvh_next list entry)", "(prologue node visit)",
         "scan trip: sub-cell candidate", "scan trip: 4 records of the full list", "exact evaluation", "near-tie rescan", "-",
         "empty node: minDistance", "outside the cube: all primitives", "getNormal (4 samples)", "wave start", "scan trip: full-list tail record"]
print("%-40s %12s %14s %6s %10s %10s" % ("event", "wave execs", "lanes", "util", "lanes/px", "execs/wave"))
waves = max(int(out[14]), 1)
for i, n in enumerate(names):
    w, l = int(out[i]), int(out[i + 16])
    print("%-40s %12d %14d %5.1f%% %10.2f %10.2f" % (n, w, l, 100.0 * l / (64.0 * w) if w else 0.0, l / float(W * H), w / float(waves)))
