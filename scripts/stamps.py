"""Diagnostic: per-section cycle shares of the v2 wave loop (needs a -DRM_STAMPS build: make -C cpu_raymarcher_amd/csrc
EXTRA=-DRM_STAMPS OUT=../librm_hip_stamps.so; RM_HIP_LIB=.../librm_hip_stamps.so).  A frame alone: the shares include the
latency a lone frame cannot hide.  `inflight=S`: S frames on S streams with bench.py's in-flight options, where the kernel
runs at its issue rate, so a section's share of the waves' time is close to its share of the issue slots.
usage: python scripts/stamps.py [inflight=12] [k=v ...]"""
import ctypes as C, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cpu_raymarcher_amd as R
from cpu_raymarcher_amd import _native as N
W, H = 3840, 2160
ctx = R.Context(0)
S = 1
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "inflight":
        S = int(v)
    else:
        ctx.set_option(k, int(v))
if S > 1:
    for k, v in (("blocks_per_cu", 1), ("lpt", 0), ("item_px", 256), ("tile_w", 8)):
        ctx.set_option(k, v)
sc = R.Scene("BVH", ctx=ctx); sc.loadPreset(3)
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
sets = [[torch.zeros(n * W * H, dtype=torch.uint8, device=dev) for n in (1, 3, 2, 2, 4)] for _ in range(S)]
def frames(n):
    for i in range(n):
        b = sets[i % S]
        with torch.cuda.stream(streams[i % S]):
            R.SphereTracer().runRaymarcher(sc, b[0], b[1], b[2].view(torch.int16), b[3].view(torch.int16), W, H, 0.0, shadedBuffer=b[4], shader="iteration-heatmap")
    torch.cuda.synchronize()
out = np.zeros(8, np.uint64)
frames(2 * S)
N.lib().rm_debug_read_stamps(ctx._h, out.ctypes.data_as(C.c_void_p))
frames(4 * S if S > 1 else 1)
N.lib().rm_debug_read_stamps(ctx._h, out.ctypes.data_as(C.c_void_p))
names = ["R refill+ray setup", "A bookkeeping", "B query+leaf eval", "C consume + N", "B fallback", "R prologue", None, "loop top"]
tot = float(sum(int(v) for i, v in enumerate(out) if names[i]))
print("%d frame(s) in flight" % S)
for n, v in zip(names, out):
    if n:
        print("%-22s %16d  %5.1f %%" % (n, v, 100.0 * float(v) / tot if tot else 0))
