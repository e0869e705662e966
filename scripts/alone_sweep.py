"""The C3 frame ALONE (serial launches, HIP events) under option sets.  usage: python scripts/alone_sweep.py "k=v,k=v" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cpu_raymarcher_amd as R
W, H = 3840, 2160
dev = torch.device("cuda:0")
ctx = R.Context(0)
sc = R.Scene("BVH", ctx=ctx); sc.loadPreset(3)
b = [torch.zeros(n * W * H, dtype=torch.uint8, device=dev) for n in (1, 3, 2, 2, 4)]
acc = torch.zeros(4, dtype=torch.int64, device=dev)
defaults = {}
for var in sys.argv[1:] or [""]:
    for k, v in defaults.items():
        ctx.set_option(k, v)
    for kv in filter(None, var.split(",")):
        k, v = kv.split("=")
        defaults.setdefault(k, ctx.get_option(k))
        ctx.set_option(k, int(v))
    ts = []
    for i in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        R.SphereTracer().runRaymarcher(sc, b[0], b[1], b[2].view(torch.int16), b[3].view(torch.int16), W, H, 0.0, shadedBuffer=b[4], shader="iteration-heatmap", diagnostics=acc)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%-60s alone %.3f ms (min %.3f)  avg sdf %.3f" % (var or "(defaults)", sum(ts[3:]) / len(ts[3:]), min(ts[3:]), ctx.decode_acc(acc)["total_sdf"] / (W * H)), flush=True)
