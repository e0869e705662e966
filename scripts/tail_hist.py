"""Diagnostic (needs a -DRM_STAMPS build, RM_HIP_LIB=...): when do the waves of ONE C3 frame start and finish?
usage: python scripts/tail_hist.py [k=v ...]"""
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
t = np.zeros(3 * 8192, np.uint64)
for rep in range(4):
    frame()
    N.check(ctx._h, N.lib().rm_debug_read_wave_times(ctx._h, t.ctypes.data_as(C.c_void_p)))
    st, en, ent = t[:8192].astype(np.int64), t[8192:2 * 8192].astype(np.int64), t[2 * 8192:].astype(np.int64)
    ok = (st > 0) & (en > 0) & (ent > 0)
    t0 = ent[ok].min()
    print("   staging (kernel entry -> wave loop): mean %.1f us, max %.1f us; last entry %.1f us after the first" % (((st[ok] - ent[ok]) / 100.0).mean(), ((st[ok] - ent[ok]) / 100.0).max(), (ent[ok].max() - t0) / 100.0))
    s_us, e_us = (st[ok] - t0) / 100.0, (en[ok] - t0) / 100.0
    print("frame %d: %d waves; kernel %.0f us from first start to last end; starts: p50 %.0f p99 %.0f max %.0f us; ends: p1 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f us; mean lifetime %.0f us"
          % (rep, ok.sum(), e_us.max(), np.percentile(s_us, 50), np.percentile(s_us, 99), s_us.max(), np.percentile(e_us, 1), np.percentile(e_us, 10),
             np.percentile(e_us, 50), np.percentile(e_us, 90), np.percentile(e_us, 99), e_us.max(), (e_us - s_us).mean()))
    cn = np.zeros(32, np.uint64)
    N.lib().rm_debug_read_counts(ctx._h, cn.ctypes.data_as(C.c_void_p))
    if int(cn[31]):
        print("   queue claims: %d, mean round trip of the atomic %.2f us (RM_STAMPS_CLAIM build)" % (int(cn[31]), float(cn[30]) / float(cn[31]) / 100.0))
    h, _ = np.histogram(e_us, bins=20, range=(0, e_us.max()))
    print("   ends per twentieth of the kernel:", " ".join(str(int(v)) for v in h))
