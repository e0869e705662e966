"""Do consecutive frames overlap when they are enqueued on alternating HIP streams?  The persistent render
kernel ends with a tail (the last, slowest rays); a second stream lets the next frame's workgroups take the
CUs that drain.  Prints ms/frame with 1 stream and with 2 / 3 streams, whole frames and 1/8-row shards.
usage: python scripts/overlap_probe.py [frames] [N=world ...] [S=streams ...] [option=value ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cpu_raymarcher_amd as R
from cpu_raymarcher_amd import distributed as D


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
    opts = [a for a in sys.argv[1:] if "=" in a and not a.startswith("N=") and not a.startswith("S=")]
    stream_counts = [int(a[2:]) for a in sys.argv[1:] if a.startswith("S=")] or [1, 6]
    W, H = 3840, 2160
    dev = torch.device("cuda:0")
    ctx = R.Context(0)
    for kv in opts:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    print("options", opts, flush=True)
    scene = R.Scene("BVH", ctx=ctx)
    scene.loadPreset(3)
    worlds = [int(a[2:]) for a in sys.argv[1:] if a.startswith("N=")] or [1, 8]
    for world, sections in [(w, ("rgba", "sdf", "iters")) for w in worlds]:
        layout = D.FrameLayout(W, H, world, sections, "interleaved", 16)
        print("sections", sections, flush=True)
        render_all = D.gpu_render_all(ctx, scene, W, H, "iteration-heatmap", layout, 0)
        for nstreams in stream_counts:
            streams = [torch.cuda.Stream() for _ in range(nstreams)]
            packed = [torch.zeros(layout.nbytes, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
            accs = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(nstreams)]

            def frame(i):
                k = i % nstreams
                with torch.cuda.stream(streams[k]):
                    render_all(packed[k])
                    sdf = layout.section(packed[k], "sdf").view(torch.int16)
                    it = layout.section(packed[k], "iters").view(torch.int16)
                    ctx.reduce_counters_enqueue(sdf, it, accs[k])

            for i in range(6):
                frame(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(frames):
                frame(i)
            t_enq = time.perf_counter() - t0  # host time to enqueue everything (close to dt: the host is the limit)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("1/%d of the rows, %d stream(s): %.3f ms/frame (%.0f frames/s), host enqueue %.3f ms/frame  acc %s"
                  % (world, nstreams, 1e3 * dt / frames, frames / dt, 1e3 * t_enq / frames, accs[0].tolist()[:2]), flush=True)


if __name__ == "__main__":
    main()
