"""Quick kernel iteration check on one GPU: for each workload, the five buffers' SHA-256 against the committed golden
fixtures (tests/golden/golden.json: C2, C3, C5 at BASELINE sizes), the render kernel alone (serial launches, HIP
events) and the frame rate with S frames in flight (render + diagnostics, bench.py's in-flight options).
usage: python scripts/kbench.py [C3 C2 C5 ...] [S=12] [frames=96] [opt=value ...]"""
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import cpu_raymarcher_amd as R

WL = {
    "C2": dict(golden="C2_grid_1080p_bvh_phong", preset=2, accel="BVH", W=1920, H=1080, shader="phong"),
    "C3": dict(golden="C3_dense_4k_bvh_iterheat", preset=3, accel="BVH", W=3840, H=2160, shader="iteration-heatmap"),
    "C3r": dict(preset=3, accel="BVH", W=3840, H=2160, shader="iteration-heatmap", ang=(0.3, 0.7)),
    "C3o": dict(preset=3, accel="Octree", W=3840, H=2160, shader="iteration-heatmap"),
    "C5": dict(golden="C5_random10k_4k_octree_iterheat", synthetic=10000, accel="Octree", W=3840, H=2160, shader="iteration-heatmap"),
    "C5b": dict(synthetic=10000, accel="BVH", W=960, H=540, shader="iteration-heatmap"),
    "N3mixed": dict(mixed=40, accel="BVH", W=3840, H=2160, shader="phong"),
    "N4chicken": dict(preset=17, accel="BVH", W=3840, H=2160, shader="phong"),
    "N4screw": dict(preset=16, accel="BVH", W=3840, H=2160, shader="phong"),
    "N4mandel": dict(preset=13, accel="BVH", W=1920, H=1080, shader="phong"),
    "P0": dict(preset=0, accel="BVH", W=3840, H=2160, shader="phong"),
    "P1": dict(preset=1, accel="BVH", W=3840, H=2160, shader="phong"),
    "P4": dict(preset=4, accel="BVH", W=3840, H=2160, shader="phong"),
    "P8": dict(preset=8, accel="BVH", W=3840, H=2160, shader="phong"),
    "P9": dict(preset=9, accel="BVH", W=3840, H=2160, shader="phong"),
    "P9oct": dict(preset=9, accel="Octree", W=3840, H=2160, shader="phong"),
    "P0none": dict(preset=0, accel="None", W=3840, H=2160, shader="phong"),
    "N4screwNone": dict(preset=16, accel="None", W=3840, H=2160, shader="phong"),
    "N4screwOct": dict(preset=16, accel="Octree", W=3840, H=2160, shader="phong"),
    "N4sixty7": dict(preset=18, accel="BVH", W=3840, H=2160, shader="phong"),
    "N4smooth": dict(preset=11, accel="BVH", W=3840, H=2160, shader="phong"),
}


def main():
    names = [a for a in sys.argv[1:] if "=" not in a] or ["C3"]
    kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    S = int(kv.pop("S", 12))
    want_hash = kv.pop("hash", None)
    fused = int(kv.pop("fused", 1))  # diagnostics from the render kernel itself (rm_render_attach_diagnostics); 0: reduce kernels
    frames = int(kv.pop("frames", 96))
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    dev = torch.device("cuda:0")
    ctx = R.Context(0)
    for k, v in kv.items():
        ctx.set_option(k, int(v))
    streams = [torch.cuda.Stream() for _ in range(S)]
    for name in names:
        wl = WL[name]
        W, H = wl["W"], wl["H"]
        sc = R.Scene(wl["accel"], ctx=ctx)
        if "synthetic" in wl:
            from cpu_raymarcher_amd.synthetic import synthetic_spheres
            sp = synthetic_spheres(wl["synthetic"])
            sc.loadSpheres(sp[:, :3], sp[:, 3])
        elif "mixed" in wl:
            from cpu_raymarcher_amd.synthetic import mixed_prims_as_triples, synthetic_mixed_prims
            sc.loadPrims(mixed_prims_as_triples(synthetic_mixed_prims(wl["mixed"]), R.make_transform))
        else:
            sc.loadPreset(wl["preset"])
        sc.camera.setAngles(*wl.get("ang", (0.0, 0.0)))
        u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)  # noqa: E731
        sets = [dict(d=u8(W * H), n=u8(3 * W * H), s=torch.zeros(W * H, dtype=torch.int16, device=dev),
                     i=torch.zeros(W * H, dtype=torch.int16, device=dev), r=u8(4 * W * H),
                     acc=torch.zeros(4, dtype=torch.int64, device=dev)) for _ in range(S)]
        tr = R.SphereTracer()

        def render(b, diag=None):
            tr.runRaymarcher(sc, b["d"], b["n"], b["s"], b["i"], W, H, 0.0, shadedBuffer=b["r"], shader=wl["shader"], diagnostics=diag)

        b = sets[0]
        render(b)
        torch.cuda.synchronize()
        verdict = "(no fixture)"
        if want_hash:
            verdict = hashlib.sha256(b"".join(t.cpu().numpy().tobytes() for t in (b["d"], b["n"], b["s"], b["i"], b["r"]))).hexdigest()[:16]
        if "golden" in wl:
            g = golden[wl["golden"]]["sha256"]
            bad = [k for k, t in (("depth", b["d"]), ("normal", b["n"]), ("sdf", b["s"]), ("iters", b["i"]), ("rgba", b["r"]))
                   if not (k == "rgba" and wl["shader"] == "phong") and hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest() != g[k]]
            verdict = "golden OK" if not bad else "GOLDEN MISMATCH " + ",".join(bad)
        for _ in range(max(0, ctx.get_option("specialise_v2_after"))):  # (the launch at which a configuration's own kernel is compiled is not timed)
            render(b)
        torch.cuda.synchronize()
        ev = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            render(b)
            e1.record()
            torch.cuda.synchronize()
            ev.append(e0.elapsed_time(e1))
        alone = sum(ev[2:]) / len(ev[2:])
        kern = ctx.last_kernel()
        inflight = {"blocks_per_cu": 1, "item_px": 256, "tile_w": 8}
        saved = {k: ctx.get_option(k) for k in inflight}
        for k, v in inflight.items():
            if k not in kv:
                ctx.set_option(k, v)

        def run(n):
            for f in range(n):
                k = f % S
                with torch.cuda.stream(streams[k]):
                    if fused:
                        render(sets[k], sets[k]["acc"])
                    else:
                        render(sets[k])
                        ctx.reduce_counters_enqueue(sets[k]["s"], sets[k]["i"], sets[k]["acc"])
        run(2 * S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(frames)
        torch.cuda.synchronize()
        fps = frames / (time.perf_counter() - t0)
        for k, v in saved.items():
            ctx.set_option(k, v)
        d = ctx.decode_acc(sets[0]["acc"])
        chk = torch.zeros(4, dtype=torch.int64, device=dev)  # the fused diagnostics against the reduction kernels
        ctx.reduce_counters_enqueue(sets[0]["s"], sets[0]["i"], chk)
        torch.cuda.synchronize()
        if ctx.decode_acc(chk) != d:
            verdict += " DIAG MISMATCH %s vs %s" % (d, ctx.decode_acc(chk))
        print("%-9s %-16s alone %7.3f ms | %2d in flight %7.1f frames/s (%.3f ms) | avg sdf %.3f it %.3f | %s"
              % (name, verdict, alone, S, fps, 1e3 / fps, d["total_sdf"] / (W * H), d["total_iters"] / (W * H), kern), flush=True)
        done, bad, log = ctx.rtc_status()
        if bad:
            print("   run-time compiles: %d ready, %d FAILED: %s" % (done, bad, log[:1500]), flush=True)


if __name__ == "__main__":
    main()
