"""Ad-hoc GPU probe: parity of the HIP path vs the C oracle on a matrix of configs, plus
first timings.  (The curated version lives in tests/test_gpu_parity.py.)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cpu_raymarcher_amd as R
from oracle import oracle as O

ctx = R.Context(0)
rng = np.random.default_rng(1)

# 1. hypot selftest
xyz = (rng.standard_normal((400000, 3)) * 2).astype(np.float32)
xyz[:1000] *= 1e-20
xyz[1000:2000, 1] = 0
xyz[2000:2100] = 0
got = ctx.selftest_hypot(xyz)
L = O.lib()
ref = np.array([L.ro_hypot3(float(a), float(b), float(c)) for a, b, c in xyz[:60000]])
print("hypot mismatches (60000):", int((got[:60000] != ref).sum()))

def cmp(preset, accel, W, H, ang=(0, 0), spheres=None, rows=None):
    sc = R.Scene(accel, ctx=ctx)
    if spheres is not None:
        sc.loadSpheres(spheres[:, :3], spheres[:, 3])
    else:
        sc.loadPreset(preset)
    sc.camera.setAngles(*ang)
    y0, y1 = rows if rows else (0, H)
    n = W * (y1 - y0)
    d = np.zeros(n, np.uint8); nb = np.zeros(3 * n, np.uint8); s = np.zeros(n, np.uint16); it = np.zeros(n, np.uint16)
    t0 = time.time()
    R.SphereTracer().runRaymarcher(sc, d, nb, s, it, W, H, 0.0, y0, y1)
    tg = time.time() - t0
    t0 = time.time()
    osc = O.OracleScene(preset=preset, accel=accel, spheres=spheres)
    osc.set_angles(*ang)
    rd, rn, rs, ri = osc.render(W, H, y0, y1)
    tc = time.time() - t0
    res = dict(depth=int((d != rd).sum()), normal=int((nb != rn).sum()), sdf=int((s != rs).sum()), iters=int((it != ri).sum()))
    for sh in R.SHADERS:
        rg = np.zeros(4 * n, np.uint8)
        R.createShadingModelFromValue(sh, ctx).shade(rg, d, nb, s, it, W, y1 - y0)
        rr = O.shade(sh, rd, rn, rs, ri, W, y1 - y0)
        diff = np.abs(rg.astype(int) - rr.astype(int))
        res[sh] = (int((diff != 0).sum()), int(diff.max()))
    dg = ctx.reduce_counters(s, it); do = O.diagnostics(rs, ri)
    res['diag'] = all(dg[k] == do[k] for k in do)
    print(preset if spheres is None else 'synthetic%d' % len(spheres), accel, W, H, ang, rows, res, 'gpu %.3fs cpu %.2fs avg sdf %.1f' % (tg, tc, rs.mean()), flush=True)
    return res

FULL = '--timing-only' not in sys.argv
for preset in (range(5) if FULL else []):
    for accel in ("None", "BVH", "Octree"):
        cmp(preset, accel, 200, 150)
if FULL:
    cmp(3, "BVH", 320, 240, (0.3, 0.7))
    cmp(3, "Octree", 320, 240, (-0.4, 2.1))
    cmp(3, "BVH", 320, 240, (1.2, -0.3), rows=(37, 111))
    cmp(2, "BVH", 480, 270)
    sp = O.synthetic_spheres(2000)
    cmp(None, "Octree", 256, 144, spheres=sp)
    cmp(None, "BVH", 256, 144, spheres=sp)
    cmp(None, "None", 64, 36, spheres=sp)

# timing at full size, device buffers
import torch
W, H = 3840, 2160
sc = R.Scene("BVH", ctx=ctx); sc.loadPreset(3)
dev = torch.device("cuda:0")
d = torch.zeros(W * H, dtype=torch.uint8, device=dev); nb = torch.zeros(3 * W * H, dtype=torch.uint8, device=dev)
s = torch.zeros(W * H, dtype=torch.int16, device=dev); it = torch.zeros(W * H, dtype=torch.int16, device=dev)
rg = torch.zeros(4 * W * H, dtype=torch.uint8, device=dev)
for tw in (8, 16, 32, 64):
    ctx.set_option("tile_w", tw)
    tr = R.SphereTracer()
    tr.runRaymarcher(sc, d, nb, s, it, W, H, 0.0, shadedBuffer=rg, shader="iteration-heatmap")
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        tr.runRaymarcher(sc, d, nb, s, it, W, H, 0.0, shadedBuffer=rg, shader="iteration-heatmap")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    dg = ctx.reduce_counters(s, it)
    print("C3 4K BVH tile_w=%d: %.3f ms/frame  %.1f fps  avg sdf %.2f avg it %.2f" % (tw, ms, 1000 / ms, dg['total_sdf'] / (W * H), dg['total_iters'] / (W * H)), flush=True)
