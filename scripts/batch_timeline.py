"""Diagnostic (EXTRA="-DRM_STAMPS -DRM_STAMPS_LOG" build): how fast do the waves of ONE C3 frame get through their batches, over the
life of the kernel?  Batches started per 100 us window (first 2048 waves), alone and with S frames in flight.
usage: python scripts/batch_timeline.py [inflight=S] [k=v ...]"""
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
log = np.zeros(2048 * 96, np.uint32)
frames(3 * S)
N.lib().rm_debug_read_batch_log(ctx._h, log.ctypes.data_as(C.c_void_p))
frames(S)  # with S > 1 the S kernels overwrite each other's entries: the log then shows whichever wrote last (same wave index, other frame)
N.check(ctx._h, N.lib().rm_debug_read_batch_log(ctx._h, log.ctypes.data_as(C.c_void_p)))
L = log.reshape(2048, 96).astype(np.int64)
have = L > 0
t0 = L[have].min()
d = np.diff(np.where(have, L, 0), axis=1)
ok = have[:, 1:] & have[:, :-1]
dur = d[ok] / 100.0
print("%d frame(s) in flight: %d batch intervals of %d waves; batch duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f" %
      (S, ok.sum(), have.any(axis=1).sum(), dur.mean(), np.percentile(dur, 10), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99)))
rel = (L[have] - t0) / 100.0
h, edges = np.histogram(rel, bins=np.arange(0, rel.max() + 100, 100))
print("batches started per 100 us:", " ".join(str(int(v)) for v in h))
