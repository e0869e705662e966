"""Diagnostic: the distribution of the item durations a lone C3 frame records for the longest-first order (option lpt).
usage: python scripts/lpt_costs.py [k=v ...]"""
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
b = [torch.zeros(n * W * H, dtype=torch.uint8, device=dev) for n in (1, 3, 2, 2, 4)]
for rep in range(3):
    R.SphereTracer().runRaymarcher(sc, b[0], b[1], b[2].view(torch.int16), b[3].view(torch.int16), W, H, 0.0, shadedBuffer=b[4], shader="iteration-heatmap")
    torch.cuda.synchronize()
    out = np.zeros(64 * 4096, np.uint8)
    N.check(ctx._h, N.lib().rm_debug_read_lpt_costs(ctx._h, out.ctypes.data_as(C.c_void_p), out.size))
    c = out.reshape(64, 4096).astype(np.float64) * 10.24
    nz = c[c > 0]
    print("frame %d: %d items recorded; duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; sum %.1f ms of wave time; items >= 4x mean: %d, >= 2x: %d"
          % (rep, nz.size, nz.mean(), np.percentile(nz, 50), np.percentile(nz, 90), np.percentile(nz, 99), nz.max(), nz.sum() / 1e3,
             (nz >= 4 * nz.mean()).sum(), (nz >= 2 * nz.mean()).sum()))
    h, edges = np.histogram(nz, bins=[0, 20, 40, 80, 160, 320, 640, 1000, 1500, 2700])
    print("   histogram (us):", " ".join("%d-%d:%d" % (edges[i], edges[i + 1], h[i]) for i in range(len(h))))
# where are the slowest items?  (queue, slot) -> tile: rm_render_v2.hip queue_entry with perm = identity
item_px = ctx.get_option("item_px"); tw = ctx.get_option("tile_w"); th = item_px // tw
tiles_x = (W + tw - 1) // tw; tiles_y = (H + th - 1) // th
def rows_of(q):
    x, s_ = q // 8, q % 8
    Rx = (tiles_y - x + 7) >> 3
    return (Rx - s_ + 7) // 8 if Rx > s_ else 0
slow = []
for q in range(64):
    R_ = rows_of(q)
    for slot in range(R_ * tiles_x):
        v = c[q, slot]
        if v >= 400:
            qi, col = divmod(slot, tiles_x)
            mid = R_ >> 1
            j = mid - ((qi + 1) >> 1) if (qi & 1) else mid + (qi >> 1)
            row = (q // 8) + 8 * ((q % 8) + 8 * j)
            slow.append((v, col * tw, row * th))
slow.sort(reverse=True)
print("   %d items >= 400 us; the slowest (us, x, y of the tile's corner):" % len(slow), [(int(v), x, y) for v, x, y in slow[:24]])
xs = np.array([x for _, x, _ in slow]); ys = np.array([y for _, _, y in slow])
if len(slow):
    print("   x histogram of slow items (16 bins over the width):", np.histogram(xs, bins=16, range=(0, W))[0].tolist())
    print("   y histogram (16 bins over the height):", np.histogram(ys, bins=16, range=(0, H))[0].tolist())
