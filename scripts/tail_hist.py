"""Diagnostic (needs a -DRM_STAMPS build, RM_HIP_LIB=...): when do the waves of ONE C3 frame finish?  32 buckets of 64 us
from the first wave's start.  usage: python scripts/tail_hist.py [k=v ...]"""
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
bufs = [torch.zeros(n * W * H, dtype=torch.uint8, device=dev) for n in (1, 3, 2, 2, 4)]
def frame():
    R.SphereTracer().runRaymarcher(sc, bufs[0], bufs[1], bufs[2].view(torch.int16), bufs[3].view(torch.int16), W, H, 0.0, shadedBuffer=bufs[4], shader="iteration-heatmap")
    torch.cuda.synchronize()
frame()
st = np.zeros(8, np.uint64); cn = np.zeros(32, np.uint64)
import torch
# stamps[7] holds the minimum start time: reset it to "infinity" through the library's own read-and-clear, then set by hand
N.lib().rm_debug_read_stamps(ctx._h, st.ctypes.data_as(C.c_void_p)); N.lib().rm_debug_read_counts(ctx._h, cn.ctypes.data_as(C.c_void_p))
for rep in range(3):
    frame()
    N.lib().rm_debug_read_stamps(ctx._h, st.ctypes.data_as(C.c_void_p)); N.lib().rm_debug_read_counts(ctx._h, cn.ctypes.data_as(C.c_void_p))
    tot = int(cn.sum())
    print("frame %d: %d waves; finished by bucket (64 us each):" % (rep, tot), " ".join("%d" % int(v) for v in cn[:32]))
