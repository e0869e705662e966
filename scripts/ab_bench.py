"""A/B timing of kernel variants / options on one GPU, with an equality check of every
variant's five buffers against the first one (hash on device buffers copied to host).
usage: python scripts/ab_bench.py [C3|C2|C5|C3o|C5b] ["k=v,k=v" ...]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cpu_raymarcher_amd as R

WL = {
    "C2": dict(preset=2, accel="BVH", W=1920, H=1080, shader="phong"),
    "C3": dict(preset=3, accel="BVH", W=3840, H=2160, shader="iteration-heatmap"),
    "C3r": dict(preset=3, accel="BVH", W=3840, H=2160, shader="iteration-heatmap", ang=(0.3, 0.7)),
    "C3o": dict(preset=3, accel="Octree", W=3840, H=2160, shader="iteration-heatmap"),
    "C3n": dict(preset=3, accel="None", W=1920, H=1080, shader="iteration-heatmap"),
    "C5": dict(synthetic=10000, accel="Octree", W=3840, H=2160, shader="iteration-heatmap"),
    "C5b": dict(synthetic=10000, accel="BVH", W=960, H=540, shader="iteration-heatmap"),
}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    variants = sys.argv[2:] or ["kernel=1", "kernel=2"]
    wl = WL[name]
    W, H = wl["W"], wl["H"]
    dev = torch.device("cuda:0")
    ctx = R.Context(0)
    sc = R.Scene(wl["accel"], ctx=ctx)
    if "synthetic" in wl:
        from cpu_raymarcher_amd.synthetic import synthetic_spheres
        sp = synthetic_spheres(wl["synthetic"])
        sc.loadSpheres(sp[:, :3], sp[:, 3])
    else:
        sc.loadPreset(wl["preset"])
    sc.camera.setAngles(*wl.get("ang", (0.0, 0.0)))
    d = torch.zeros(W * H, dtype=torch.uint8, device=dev)
    nb = torch.zeros(3 * W * H, dtype=torch.uint8, device=dev)
    s = torch.zeros(W * H, dtype=torch.int16, device=dev)
    it = torch.zeros(W * H, dtype=torch.int16, device=dev)
    rg = torch.zeros(4 * W * H, dtype=torch.uint8, device=dev)
    tr = R.SphereTracer()
    ref = None
    defaults = {k: ctx.get_option(k) for k in ("kernel", "tile_w", "filter", "nodes_in_lds", "list_cap", "coop", "grid",
                                               "blocks_per_cu", "refill", "hw_xcd", "item_px", "nn")}
    for v in variants:
        for k, val in defaults.items():
            ctx.set_option(k, val)
        for kv in v.split(","):
            if kv:
                k, val = kv.split("=")
                ctx.set_option(k, int(val))
        for b in (d, nb, s, it, rg):
            b.zero_()
        tr.runRaymarcher(sc, d, nb, s, it, W, H, 0.0, shadedBuffer=rg, shader=wl["shader"])
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for b in (d, nb, s, it, rg):
            h.update(b.cpu().numpy().tobytes())
        hh = h.hexdigest()[:16]
        if ref is None:
            ref = hh
        n = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            tr.runRaymarcher(sc, d, nb, s, it, W, H, 0.0, shadedBuffer=rg, shader=wl["shader"])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        dg = ctx.reduce_counters(s, it)
        print("%-4s %-40s %8.3f ms  %7.1f fps  hash %s %s  avg sdf %.2f it %.2f" %
              (name, v, ms, 1000 / ms, hh, "OK" if hh == ref else "MISMATCH", dg["total_sdf"] / (W * H),
               dg["total_iters"] / (W * H)), flush=True)


if __name__ == "__main__":
    main()
