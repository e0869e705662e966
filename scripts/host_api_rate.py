import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import cpu_raymarcher_amd as R
W, H = 3840, 2160
scene = R.Scene("BVH"); scene.loadPreset(3)
depth = np.zeros(W*H, np.uint8); normal = np.zeros(W*H*3, np.uint8); sdf = np.zeros(W*H, np.uint16); iters = np.zeros(W*H, np.uint16)
rgba = np.zeros(W*H*4, np.uint8)
tr = R.SphereTracer(); sh = R.createShadingModelFromValue("iteration-heatmap", scene.ctx)
for _ in range(2): tr.runRaymarcher(scene, depth, normal, sdf, iters, W, H, 0.0)
t0 = time.perf_counter(); n = 10
for _ in range(n): tr.runRaymarcher(scene, depth, normal, sdf, iters, W, H, 0.0)
t1 = time.perf_counter()
for _ in range(n): sh.shade(rgba, depth, normal, sdf, iters, W, H)
t2 = time.perf_counter()
print("rm_render_tile (host numpy buffers, 8 B/px back over PCIe): %.2f ms/frame = %.1f frames/s" % (1e3*(t1-t0)/n, n/(t1-t0)))
print("rm_shade (host buffers in and out): %.2f ms/frame" % (1e3*(t2-t1)/n))
print("sum sdf", int(sdf.astype(np.int64).sum()))
