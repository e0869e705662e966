"""Diagnostic: per-section cycle shares of the v2 wave loop (needs a -DRM_STAMPS build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cpu_raymarcher_amd as R
from cpu_raymarcher_amd import _native as N
W, H = 3840, 2160
ctx = R.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
sc = R.Scene("BVH", ctx=ctx); sc.loadPreset(3)
dev = torch.device("cuda:0")
d = torch.zeros(W*H, dtype=torch.uint8, device=dev); nb = torch.zeros(3*W*H, dtype=torch.uint8, device=dev)
s = torch.zeros(W*H, dtype=torch.int16, device=dev); it = torch.zeros(W*H, dtype=torch.int16, device=dev)
rg = torch.zeros(4*W*H, dtype=torch.uint8, device=dev)
out = np.zeros(8, np.uint64)
N.lib().rm_debug_read_stamps(ctx._h, out.ctypes.data_as(C.c_void_p))
R.SphereTracer().runRaymarcher(sc, d, nb, s, it, W, H, 0.0, shadedBuffer=rg, shader="iteration-heatmap")
torch.cuda.synchronize()
N.lib().rm_debug_read_stamps(ctx._h, out.ctypes.data_as(C.c_void_p))
names = ["R refill+ray setup", "A bookkeeping", "B query+leaf eval", "C consume", "B fallback", "R prologue", "-", "loop top"]
tot = float(out.sum())
for n, v in zip(names, out):
    print("%-22s %14d  %5.1f %%" % (n, v, 100.0 * float(v) / tot if tot else 0))
