"""How does the lone frame's time scale with its size (same scene, same camera: the image is stretched, reference
raymarcher.ts:73,83)?  T(H) = a * H + b separates the throughput-bound part from the fixed part (ramp + tail).
usage: python scripts/size_scaling.py [k=v ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cpu_raymarcher_amd as R
dev = torch.device("cuda:0")
ctx = R.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
sc = R.Scene("BVH", ctx=ctx); sc.loadPreset(3)
W = 3840
res = []
for H in (540, 1080, 2160, 4320, 8640):
    b = [torch.zeros(n * W * H, dtype=torch.uint8, device=dev) for n in (1, 3, 2, 2, 4)]
    acc = torch.zeros(4, dtype=torch.int64, device=dev)
    ts = []
    for i in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        R.SphereTracer().runRaymarcher(sc, b[0], b[1], b[2].view(torch.int16), b[3].view(torch.int16), W, H, 0.0, shadedBuffer=b[4], shader="iteration-heatmap", diagnostics=acc)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sum(ts[3:]) / len(ts[3:])
    res.append((H, t))
    print("%dx%d: %.3f ms  (%.3f ms per 2160 rows)  avg sdf %.3f  %s" % (W, H, t, t * 2160 / H, ctx.decode_acc(acc)["total_sdf"] / (W * H), ctx.last_kernel()), flush=True)
    del b
(h1, t1), (h2, t2) = res[2], res[3]
a = (t2 - t1) / (h2 - h1)
print("between 2160 and 4320 rows: %.3f ms per 2160 rows + %.3f ms fixed" % (a * 2160, t1 - a * h1))
