"""Diagnostic: how often each part of the v2 wave loop executes in one C3 frame (needs a -DRM_COUNTS build:
make -B -C cpu_raymarcher_amd/csrc EXTRA=-DRM_COUNTS).  Per event: wave-level executions, lanes active in them, lane
utilisation; per pixel figures.  usage: python scripts/counts.py [k=v ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cpu_raymarcher_amd as R
from cpu_raymarcher_amd import _native as N

W, H = 3840, 2160
ctx = R.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
sc = R.Scene("BVH", ctx=ctx)
sc.loadPreset(3)
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
names = ["wave-loop iteration", "R refill section", "A march-step bookkeeping pass", "A bvh_next call", "A bvh_next list entry",
         "R prologue node visit", "B getDistance", "B cell leaf-list entry", "B leaf sphere scan", "B nn-list sphere scan",
         "B exact evaluation", "B near-tie redo", "B cooperative fallback ray", "B fallback not served by nn", "A normal sample",
         "M in-round march step"]
print("%-32s %12s %14s %6s %10s" % ("event", "wave execs", "lanes", "util", "lanes/px"))
for i, n in enumerate(names):
    w, l = int(out[i]), int(out[i + 16])
    print("%-32s %12d %14d %5.1f%% %10.2f" % (n, w, l, 100.0 * l / (64.0 * w) if w else 0.0, l / float(W * H)))
